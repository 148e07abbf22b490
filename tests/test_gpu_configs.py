"""GPU parity at the shapes BASELINE.json's configs name and the batch-1 edge the reference's predict path hits:

 * Pix2Pix 256x256 batch 1 with BatchNorm (pix2pix.py:228 on np.expand_dims inputs, :311,:338): the 1x1x512
   bottleneck sees one value per channel, variance 0, output == beta - the dead bottleneck must be reproduced;
 * Pix2Pix 512x512 bf16 (config 4's per-GPU shape, batch 8) through the directional-derivative property, and
   oracle parity at 512x512 bf16 batch 2;
 * CycleGAN 512x512 (config 5's resolution) fp32 against the oracle at batch 1, and CycleGAN 256x256 bf16 batch 4
   (config 3) through the same property.
"""
import numpy as np
import pytest
import torch

from oracle import gan_oracle as O

pytestmark = pytest.mark.gpu


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def test_pix2pix_batch1_batchnorm_dead_bottleneck_f32():
    """generate_images' call (pix2pix.py:228): batch of 1, BatchNorm batch statistics, dropout on."""
    from gan_amd.base_gan import GeneratorModel
    from gan_amd.nets import Ctx
    from gan_amd.steps import Pix2PixStep
    ctx = Ctx('cuda:0', 'f32')
    st = Pix2PixStep(ctx, 1, 256, 1, lam=100.0, seed=123)
    Gp, Dp = O.init_generator(1, seed=11), O.init_discriminator(1, True, seed=12)
    rng = np.random.default_rng(3)
    for k in Gp:                              # non-trivial beta so that "output == beta" is visible
        if k.endswith('.beta'):
            Gp[k] = (0.2 * rng.standard_normal(Gp[k].shape)).astype(np.float32)
    st.G.params.load_numpy(Gp); st.D.params.load_numpy(Dp)
    beta7 = Gp['down7.beta'].astype(np.float32).copy()         # (the oracle's train step updates Gp in place)
    inp, tar = O.synthetic_pair(1, 256, 1, seed=123)
    masks = O.dropout_masks(1, 256, seed=5)
    st.g.set_dropmasks(masks)
    ref = O.pix2pix_train_step(Gp, Dp, O.AdamTF(), O.AdamTF(), inp, tar, 100.0, masks, True, return_grads=True)
    ti, tt = torch.from_numpy(inp).to(ctx.device), torch.from_numpy(tar).to(ctx.device)
    losses = st.train_step(ti, tt, False).cpu().numpy()          # forward + losses (pix2pix.py:291-292 val pass)
    gen = st.g.output_f32().cpu().numpy()
    err = float(np.abs(gen - ref[4]).max())
    print(f"batch-1 BN: gen max-abs err {err:.3e}; losses {losses} ref {[float(v) for v in ref[:4]]}")
    assert err < 1e-3 and np.allclose(losses, np.array(ref[:4], np.float64), rtol=1e-3)
    # rows-per-group = 1 through reduce / finalize / norm_act_fwd: LeakyReLU(beta) exactly, whatever the input
    a7 = st.g.a7.t.float().cpu().numpy().reshape(-1)
    b = beta7
    assert np.allclose(a7, np.where(b > 0, b, 0.3 * b), atol=1e-6)
    mean, rstd = st.g.stats['down7']
    # variance 0 + Keras eps (E[y^2] - mean^2 from fp32 partial sums leaves ~1e-7 * y^2 of rounding in the variance)
    assert np.allclose(rstd.cpu().numpy(), 1.0 / np.sqrt(1e-3), rtol=1e-3)
    # and the full step at batch 1 (dgrad / wgrad / BN backward with a dead bottleneck)
    losses = st.train_step(ti, tt, True).cpu().numpy()
    gG = st.G.params.to_numpy('grad')
    assert np.abs(gG["down7.kernel"]).max() < 1e-4 * np.abs(ref[5]["up0.kernel"]).max() + 1e-12     # nothing flows through
    for k in ('up0.kernel', 'up6.kernel', 'down0.kernel', 'last.kernel'):
        assert _cos(gG[k], ref[5][k]) > 0.999, k
    # the model object the reference's generate_images calls
    model = GeneratorModel(st.G)
    out = model(inp, training=True).cpu().numpy()
    assert out.shape == (1, 256, 256, 1) and np.isfinite(out).all() and np.abs(out).max() <= 1.0


def _directional(step, nets_losses, ti, tt, eps_frac, tol):
    """<grad, delta> == central difference of two forward-only steps, for each (net, loss index)."""
    step._forward_backward(ti, tt, True)
    torch.cuda.synchronize()
    base = step.losses.clone()
    gaps = []
    for net, li in nets_losses:
        P = net.params
        g, w0 = P.grad.clone(), P.master.clone()
        if step.ctx.ls is not None:          # fp16 path: gradients carry the loss scale
            g *= step.ctx.ls[1]
        g2 = float((g.double() ** 2).sum())
        eps = eps_frac * abs(float(base[li])) / g2
        vals = []
        for sgn in (+1.0, -1.0):
            P.master.copy_(w0 + sgn * eps * g)
            P.prepare()
            step._forward_backward(ti, tt, False)
            vals.append(float(step.losses[li]))
        P.master.copy_(w0)
        P.prepare()
        fd, pred = (vals[0] - vals[1]) / 2.0, eps * g2
        print(f"loss[{li}] = {float(base[li]):.5f}: finite difference {fd:.6e} vs <g,delta> {pred:.6e}")
        assert abs(fd - pred) < tol * abs(pred), (li, fd, pred)
        gaps.append(abs(fd - pred) / abs(pred))
    return gaps


def test_pix2pix_512_bf16_batch8_directional_derivative():
    """BASELINE config 4 per-GPU shape: Pix2Pix 512x512 bf16, batch 8."""
    from gan_amd.nets import Ctx, workspace_mb_for
    from gan_amd.steps import Pix2PixStep
    ctx = Ctx('cuda:0', 'bf16', workspace_mb=workspace_mb_for(8, 512))
    st = Pix2PixStep(ctx, 8, 512, 1, lam=100.0, seed=123)
    inp, tar = O.synthetic_pair(8, 512, 1, seed=5)
    st.g.set_dropmasks(O.dropout_masks(8, 512, seed=6))
    ti, tt = torch.from_numpy(inp).to(ctx.device), torch.from_numpy(tar).to(ctx.device)
    _directional(st, ((st.G, 0), (st.D, 3)), ti, tt, 3e-2, 0.12)        # (measured 4 % / 0.02 % off)


def test_pix2pix_512_bf16_batch2_vs_oracle():
    from gan_amd.nets import Ctx
    from gan_amd.steps import Pix2PixStep
    ctx = Ctx('cuda:0', 'bf16')
    B, S = 2, 512
    st = Pix2PixStep(ctx, B, S, 1, lam=100.0, seed=123)
    Gp, Dp = O.init_generator(1, seed=41), O.init_discriminator(1, True, seed=42)
    st.G.params.load_numpy(Gp); st.D.params.load_numpy(Dp)
    inp, tar = O.synthetic_pair(B, S, 1, seed=77)
    masks = O.dropout_masks(B, S, seed=8)
    st.g.set_dropmasks(masks)
    ref = O.pix2pix_train_step(Gp, Dp, O.AdamTF(), O.AdamTF(), inp, tar, 100.0, masks, True, return_grads=True)
    losses = st.train_step(torch.from_numpy(inp).to(ctx.device), torch.from_numpy(tar).to(ctx.device), True).cpu().numpy()
    gen = st.g.output_f32().cpu().numpy()
    err = float(np.abs(gen - ref[4]).max())
    print(f"[512 bf16 B=2] gen max-abs err {err:.3e}; losses {losses} ref {[float(v) for v in ref[:4]]}")
    assert err < 0.03 and np.allclose(losses, np.array(ref[:4], np.float64), rtol=5e-3)      # (measured 1.0e-2, <= 1.1e-3)
    gG, gD = st.G.params.to_numpy('grad'), st.D.params.to_numpy('grad')
    gmax = max(np.linalg.norm(v) for v in ref[5].values())
    for k, v in ref[5].items():
        if np.linalg.norm(v) > 1e-3 * gmax:
            assert _cos(gG[k], v) > 0.9, (k, _cos(gG[k], v))
    for k in ('down0.kernel', 'conv.kernel', 'last.kernel'):
        assert _cos(gD[k], ref[6][k]) > 0.95, k


def test_cyclegan_512_f32_vs_oracle():
    """CycleGAN at 512x512 (config 5's resolution; 2x2 bottleneck, InstanceNorm over 4 values), batch 1, fp32."""
    from gan_amd.nets import Ctx
    from gan_amd.steps import CycleGANStep
    ctx = Ctx('cuda:0', 'f32')
    B, S, C = 1, 512, 1
    st = CycleGANStep(ctx, B, S, C, lam=10.0, seed=7, dropout=True)
    n = 'instancenorm'
    Ps = [O.init_generator(C, n, seed=51), O.init_generator(C, n, seed=52),
          O.init_discriminator(C, False, n, seed=53), O.init_discriminator(C, False, n, seed=54)]
    for net, P in zip(st.nets(), Ps):
        net.params.load_numpy(P)
    rx, ry = O.synthetic_pair(B, S, C, seed=19)
    keys = ['fake_y', 'cycled_x', 'fake_x', 'cycled_y', 'same_x', 'same_y']
    masks = {k: O.dropout_masks(B, S, seed=60 + i) for i, k in enumerate(keys)}
    for k, call in st.gen_calls().items():
        call.set_dropmasks(masks[k])
    out = O.cyclegan_train_step(*Ps, [O.AdamTF() for _ in range(4)], rx, ry, 10.0, masks, True, return_grads=True)
    ref_losses, fakes, grads = np.array(out[:7], np.float64), out[7], out[8:]
    losses = st.train_step(torch.from_numpy(rx).to(ctx.device), torch.from_numpy(ry).to(ctx.device), True).cpu().numpy()
    fy = st.fy.output_f32().cpu().numpy()
    err = float(np.abs(fy - fakes['fake_y']).max())
    print(f"[cyclegan 512 f32] fake_y max-abs err {err:.3e}; losses {losses} ref {ref_losses}")
    assert err < 1e-3 and np.allclose(losses, ref_losses, rtol=1e-3)
    for nm, net, g in zip(('Gg', 'Gf', 'Dx', 'Dy'), st.nets(), grads):
        got = net.params.to_numpy('grad')
        gmax = max(np.linalg.norm(v) for v in g.values())
        for k, v in g.items():
            if np.linalg.norm(v) > 1e-3 * gmax:
                assert _cos(got[k], v) > 0.999, (nm, k, _cos(got[k], v))


@pytest.mark.parametrize("batch,size", [(4, 256), (2, 512)])
def test_cyclegan_bf16_directional_derivative(batch, size):
    """BASELINE config 3 (CycleGAN 256x256 bf16) at batch 4, and config 5's resolution (512x512) in the 16-bit path:
    every generator's and discriminator's gradient is the directional derivative of its own loss
    (cycle_gan.py:252-260: total_gen_g_loss -> G_g, total_gen_f_loss -> G_f, disc_x_loss -> D_x, disc_y_loss -> D_y)."""
    from gan_amd.nets import Ctx, workspace_mb_for
    from gan_amd.steps import CycleGANStep
    ctx = Ctx('cuda:0', 'bf16', workspace_mb=workspace_mb_for(batch, size))
    st = CycleGANStep(ctx, batch, size, 1, lam=10.0, seed=7, dropout=True)
    rx, ry = O.synthetic_pair(batch, size, 1, seed=29)
    for i, call in enumerate(st.gen_calls().values()):
        call.set_dropmasks(O.dropout_masks(batch, size, seed=80 + i))
    tx, ty = torch.from_numpy(rx).to(ctx.device), torch.from_numpy(ry).to(ctx.device)
    _directional(st, ((st.Gg, 3), (st.Gf, 4), (st.Dx, 5), (st.Dy, 6)), tx, ty, 3e-2, 0.25)


def test_cyclegan_directional_derivative_curvature_not_bf16():
    """The generators' 12-15 % gap between <g, delta> and the central difference in the test above is CURVATURE of the loss
    along the gradient at a 3 % step (lambda * |.| terms: kinks on both sides), not a bf16 gradient bias: the fp32 path shows the
    same gap at the same step, and in both dtypes it shrinks about linearly when the step is halved and quartered."""
    from gan_amd.nets import Ctx, workspace_mb_for
    from gan_amd.steps import CycleGANStep
    batch, size = 4, 256
    rx, ry = O.synthetic_pair(batch, size, 1, seed=29)
    gaps = {}
    for dtype in ('f32', 'bf16'):
        for frac in (3e-2, 1.5e-2, 7.5e-3):
            ctx = Ctx('cuda:0', dtype, workspace_mb=workspace_mb_for(batch, size))
            st = CycleGANStep(ctx, batch, size, 1, lam=10.0, seed=7, dropout=True)
            for i, call in enumerate(st.gen_calls().values()):
                call.set_dropmasks(O.dropout_masks(batch, size, seed=80 + i))
            tx, ty = torch.from_numpy(rx).to(ctx.device), torch.from_numpy(ry).to(ctx.device)
            gaps[dtype, frac] = _directional(st, ((st.Gg, 3), (st.Gf, 4)), tx, ty, frac, 0.25)
    print("relative gaps (G_g, G_f):", {k: [round(x, 4) for x in v] for k, v in gaps.items()})
    for i in range(2):
        assert abs(gaps['f32', 3e-2][i] - gaps['bf16', 3e-2][i]) < 0.04          # the same gap in fp32: not a bf16 effect
        for dtype in ('f32', 'bf16'):
            assert gaps[dtype, 7.5e-3][i] < 0.6 * gaps[dtype, 3e-2][i] + 0.01     # ... and it goes away with the step
        assert gaps['f32', 7.5e-3][i] < 0.06


def test_loss_scale_state_machine():
    """Dynamic loss scaling of the fp16 path on the device (include/gan_amd.h): a non-finite gradient raises the flag,
    every Adam entry point then leaves weights, moments and the step counter alone, the update halves the scale; finite
    steps un-scale the gradients and the scale doubles after `growth_interval` of them."""
    from gan_amd.nets import Ctx
    ctx = Ctx('cuda:0', 'f16')
    lib, dev = ctx.lib, ctx.device
    ls = torch.tensor([1024.0, 1.0 / 1024.0, 0.0, 0.0], device=dev)
    n = 4096
    rng = np.random.default_rng(0)
    g_true = rng.standard_normal(n).astype(np.float32)
    p0 = rng.standard_normal(n).astype(np.float32)
    p = torch.from_numpy(p0.copy()).to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step, lr_t = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
    st = ctx.stream()

    def one_step(grad):
        assert lib.gan_grads_check(grad.data_ptr(), n, ls.data_ptr(), st) == 0
        assert lib.gan_adam_begin(step.data_ptr(), lr_t.data_ptr(), 2e-4, 0.5, 0.999, ls.data_ptr(), st) == 0
        assert lib.gan_adam_tf(p.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), n, lr_t.data_ptr(), 0.5, 0.999, 1e-7, 1.0,
                               ls.data_ptr(), 0, st) == 0
        assert lib.gan_loss_scale_update(ls.data_ptr(), 2, 65536.0, st) == 0
        torch.cuda.synchronize()

    bad = torch.from_numpy(g_true * 1024.0).to(dev)
    bad[1234] = float('inf')
    one_step(bad)
    assert step.item() == 0 and np.array_equal(p.cpu().numpy(), p0) and float(m.abs().max()) == 0.0     # skipped
    assert ls.cpu().tolist() == [512.0, 1.0 / 512.0, 0.0, 0.0]
    bad[1234] = float('nan')
    one_step(bad)
    assert step.item() == 0 and ls[0].item() == 256.0
    opt, P = O.AdamTF(), {'w': p0.astype(np.float64)}
    for k in range(2):                               # two finite steps: gradients arrive scaled, Adam un-scales them
        scale = ls[0].item()
        one_step(torch.from_numpy(g_true * scale).to(dev))
        opt.apply(P, {'w': g_true.astype(np.float64)})
        assert step.item() == k + 1
        assert np.abs(p.cpu().numpy() - P['w']).max() < 1e-6
    assert ls.cpu().tolist() == [512.0, 1.0 / 512.0, 0.0, 0.0]             # grew after growth_interval = 2 finite steps


def test_pix2pix_f16_step_vs_oracle():
    """The fp16 storage path (v_mfma_f32_16x16x32_f16, fp32 master weights, loss-scaled gradients) on a Pix2Pix step."""
    from gan_amd.nets import Ctx
    from gan_amd.steps import Pix2PixStep
    ctx = Ctx('cuda:0', 'f16', loss_scale=1024.0)
    B, S = 2, 256
    st = Pix2PixStep(ctx, B, S, 1, lam=100.0, seed=123)
    Gp, Dp = O.init_generator(1, seed=11), O.init_discriminator(1, True, seed=12)
    st.G.params.load_numpy(Gp); st.D.params.load_numpy(Dp)
    inp, tar = O.synthetic_pair(B, S, 1, seed=123)
    masks = O.dropout_masks(B, S, seed=5)
    st.g.set_dropmasks(masks)
    G0 = {k: v.copy() for k, v in Gp.items()}
    ref = O.pix2pix_train_step(Gp, Dp, O.AdamTF(), O.AdamTF(), inp, tar, 100.0, masks, True, return_grads=True)
    losses = st.train_step(torch.from_numpy(inp).to(ctx.device), torch.from_numpy(tar).to(ctx.device), True).cpu().numpy()
    gen = st.g.output_f32().cpu().numpy()
    err = float(np.abs(gen - ref[4]).max())
    print(f"[f16] gen max-abs err {err:.3e}; losses {losses} ref {[float(v) for v in ref[:4]]}; scale state {ctx.ls.cpu().tolist()}")
    assert err < 6e-3 and np.allclose(losses, np.array(ref[:4], np.float64), rtol=1e-3)      # (measured 1.7e-3, <= 7e-5)
    assert ctx.ls.cpu().tolist() == [1024.0, 1.0 / 1024.0, 1.0, 0.0]        # finite step, nothing skipped
    gG = st.G.params.to_numpy('grad')
    gmax = max(np.linalg.norm(v) for v in ref[5].values())
    for k, v in ref[5].items():
        if np.linalg.norm(v) > 1e-3 * gmax:
            assert _cos(gG[k] / 1024.0, v) > 0.98, (k, _cos(gG[k], v))
            assert abs(np.linalg.norm(gG[k]) / 1024.0 / np.linalg.norm(v) - 1.0) < 0.05, k      # scaled by exactly the loss scale
    newG = st.G.params.to_numpy()
    d = np.abs(newG['down3.kernel'] - Gp['down3.kernel'])                  # Gp now holds the oracle's post-Adam weights
    assert d.max() < 4.1e-4 and (d < 4.1e-5).mean() > 0.5
    assert np.abs(newG['down3.kernel'] - G0['down3.kernel']).max() > 1e-4  # the step was applied


def test_cyclegan_512_f16_batch16_directional_derivative():
    """BASELINE config 5's per-GPU shape: CycleGAN 512x512 fp16, batch 16."""
    from gan_amd.nets import Ctx, workspace_mb_for
    from gan_amd.steps import CycleGANStep
    B, S = 16, 512
    ctx = Ctx('cuda:0', 'f16', workspace_mb=workspace_mb_for(B, S), loss_scale=1024.0)
    st = CycleGANStep(ctx, B, S, 1, lam=10.0, seed=7, dropout=True)
    rx, ry = O.synthetic_pair(B, S, 1, seed=31)
    tx, ty = torch.from_numpy(rx).to(ctx.device), torch.from_numpy(ry).to(ctx.device)
    _directional(st, ((st.Gg, 3), (st.Gf, 4), (st.Dx, 5), (st.Dy, 6)), tx, ty, 3e-2, 0.25)
    l = st.train_step(tx, ty, True).cpu().numpy()                           # and a full step: finite, applied or cleanly skipped
    assert np.isfinite(l).all() and ctx.ls[3].item() == 0.0
