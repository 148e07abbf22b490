"""Pins the CPU oracle (oracle/gan_oracle.py): analytic known-answer values, parameter-count
KATs (SURVEY.md 8c items 4,5), finite differences, and an independent PyTorch-CPU autograd
implementation of the same graph (oracle/torch_ref.py)."""
import numpy as np
import pytest

from oracle import gan_oracle as O
from oracle import torch_ref as TR

RNG = np.random.default_rng(0)


def rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_param_counts():
    # counts recomputed from base_gan.py shapes
    assert O.trainable_count(O.init_generator(3)) == 54_414_979
    assert O.trainable_count(O.init_generator(1)) == 54_408_833
    assert O.trainable_count(O.init_discriminator(3, True)) == 2_768_641
    assert O.trainable_count(O.init_discriminator(1, True)) == 2_764_545
    assert len(O.init_generator(1)) == 45 and len(O.init_discriminator(1)) == 12


def test_shape_kats():
    # base_gan.py:141-161 comments: (bs,128,128,64) ... (bs,30,30,1); base_gan.py:180-197
    x = RNG.standard_normal((1, 256, 256, 1)).astype(np.float32)
    D = O.init_discriminator(1, True)
    out, caches = O.discriminator_fwd(D, x, x)
    assert [c['a'].shape[1:] for c in caches] == [(128, 128, 64), (64, 64, 128), (32, 32, 256), (31, 31, 512), (30, 30, 1)]
    G = O.init_generator(1)
    g, gc = O.generator_fwd(G, x)
    assert g.shape == (1, 256, 256, 1)
    assert [c['a'].shape[1] for c in gc] == [128, 64, 32, 16, 8, 4, 2, 1, 2, 4, 8, 16, 32, 64, 128, 256]
    # batch-1 bottleneck BN output == beta (SURVEY section 7 "hard parts"): relu(0)=0... lrelu(beta=0)=0
    assert np.all(gc[7]['a'] == 0)


@pytest.mark.parametrize("stride", [1, 2])
def test_conv_vs_torch(stride):
    x = RNG.standard_normal((2, 8, 8, 3))
    w = RNG.standard_normal((4, 4, 3, 5))
    y = O.conv2d_fwd(x, w, stride)
    xt, wt = TR.t(x, grad=True), TR.t(w, grad=True)
    yt = TR.conv(xt, wt, stride)
    assert y.shape == tuple(yt.shape)
    assert rel(y, yt.detach().numpy()) < 1e-12
    dy = RNG.standard_normal(y.shape)
    yt.backward(TR.t(dy))
    dx, dw = O.conv2d_bwd(x, w, dy, stride)
    assert rel(dx, xt.grad.numpy()) < 1e-12 and rel(dw, wt.grad.numpy()) < 1e-12


def test_convT_vs_torch_and_index_map():
    x = RNG.standard_normal((2, 5, 5, 3))
    w = RNG.standard_normal((4, 4, 6, 3))     # (kh,kw,cout,cin)
    y = O.convT2d_fwd(x, w)
    assert y.shape == (2, 10, 10, 6)
    # literal index map of SURVEY 2.1: out[2i+kh-1] += in[i]*w[kh]
    ref = np.zeros((2, 12, 12, 6))
    for i in range(5):
        for j in range(5):
            for kh in range(4):
                for kw in range(4):
                    ref[:, 2 * i + kh, 2 * j + kw, :] += x[:, i, j, :] @ w[kh, kw].T
    assert rel(y, ref[:, 1:-1, 1:-1]) < 1e-12
    xt, wt = TR.t(x, grad=True), TR.t(w, grad=True)
    yt = TR.convT(xt, wt)
    assert rel(y, yt.detach().numpy()) < 1e-12
    dy = RNG.standard_normal(y.shape)
    yt.backward(TR.t(dy))
    dx, dw = O.convT2d_bwd(x, w, dy)
    assert rel(dx, xt.grad.numpy()) < 1e-12 and rel(dw, wt.grad.numpy()) < 1e-12


@pytest.mark.parametrize("kind", ['batchnorm', 'instancenorm'])
def test_norm_vs_torch(kind):
    y = RNG.standard_normal((3, 4, 4, 5)) * 2 + 1
    g, b = RNG.standard_normal(5), RNG.standard_normal(5)
    z, cache = O.norm_fwd(y, g, b, kind)
    yt, gt, bt = TR.t(y, grad=True), TR.t(g, grad=True), TR.t(b, grad=True)
    zt = TR.norm(yt, gt, bt, kind)
    assert rel(z, zt.detach().numpy()) < 1e-12
    dz = RNG.standard_normal(z.shape)
    zt.backward(TR.t(dz))
    dy, dg, db = O.norm_bwd(dz, cache, g, kind)
    assert rel(dy, yt.grad.numpy()) < 1e-11 and rel(dg, gt.grad.numpy()) < 1e-12 and rel(db, bt.grad.numpy()) < 1e-12


def test_instance_norm_constant_map_is_offset():
    y = np.full((2, 4, 4, 3), 7.0)
    z, _ = O.norm_fwd(y, np.array([1., 2., 3.]), np.array([.5, .25, 0.]), 'instancenorm')
    assert np.allclose(z, np.array([.5, .25, 0.]))


def test_bce_and_l1_kats():
    x = np.zeros((1, 30, 30, 1), np.float32)
    l, d = O.bce_logits(x, 1.0)
    assert abs(l - np.log(2)) < 1e-7
    assert np.allclose(d, -0.5 / 900)
    x = RNG.standard_normal((2, 30, 30, 1)) * 5
    for tgt in (0.0, 1.0):
        l, d = O.bce_logits(x, tgt)
        xt = TR.t(x, grad=True)
        lt = TR.bce(xt, tgt)
        lt.backward()
        assert abs(l - lt.item()) < 1e-12 and rel(d, xt.grad.numpy()) < 1e-12
    a, b = RNG.standard_normal((2, 4, 4, 1)), RNG.standard_normal((2, 4, 4, 1))
    l, d = O.l1_mean(a, b)
    assert abs(l - np.abs(a - b).mean()) < 1e-15 and np.allclose(d, np.sign(a - b) / a.size)


def test_adam_tf_form():
    # step 1 with g != 0 moves theta by ~lr*sign(g); eps outside bias correction
    p = {'w': np.array([1.0, -2.0, 3.0])}
    g = {'w': np.array([0.5, -1e-3, 2.0])}
    opt = O.AdamTF(2e-4, 0.5, 0.999)
    opt.apply(p, g)
    lr_t = 2e-4 * np.sqrt(1 - 0.999) / (1 - 0.5)
    m = 0.5 * g['w']
    v = 0.001 * g['w'] ** 2
    exp = np.array([1.0, -2.0, 3.0]) - lr_t * m / (np.sqrt(v) + 1e-7)
    assert np.allclose(p['w'], exp, rtol=1e-12)
    assert np.allclose(np.array([1.0, -2.0, 3.0]) - p['w'], 2e-4 * np.sign(g['w']), rtol=1e-2)


def test_bn_moving_stats():
    P = {'l.kernel': RNG.standard_normal((4, 4, 2, 3)), 'l.gamma': np.ones(3), 'l.beta': np.zeros(3)}
    x = RNG.standard_normal((2, 8, 8, 2))
    st = {}
    O.block_fwd(x, P, 'l', 'conv', 2, 'batchnorm', 'lrelu', training_state=st)
    y = O.conv2d_fwd(x, P['l.kernel'], 2)
    n = 2 * 4 * 4
    assert np.allclose(st['l.moving_mean'], 0.01 * y.mean(axis=(0, 1, 2)))
    assert np.allclose(st['l.moving_variance'], 0.99 + 0.01 * y.var(axis=(0, 1, 2)) * n / (n - 1))


def test_block_finite_difference():
    # fp64 central differences through conv->BN->lrelu and convT->IN->dropout->relu blocks
    for kind, norm, act, wshape, xshape in [('conv', 'batchnorm', 'lrelu', (4, 4, 3, 5), (2, 8, 8, 3)),
                                            ('convT', 'instancenorm', 'relu', (4, 4, 5, 3), (2, 4, 4, 3))]:
        gk, bk = ('.gamma', '.beta') if norm == 'batchnorm' else ('.scale', '.offset')
        P = {'l.kernel': RNG.standard_normal(wshape), 'l' + gk: 1 + 0.1 * RNG.standard_normal(5),
             'l' + bk: 0.1 * RNG.standard_normal(5)}
        x = RNG.standard_normal(xshape)
        mask = (RNG.random((2, 8, 8, 5)) > 0.5).astype(np.float64) if kind == 'convT' else None
        a, c = O.block_fwd(x, P, 'l', kind, 2, norm, act, dropmask=mask)
        r = RNG.standard_normal(a.shape)
        grads = {}
        dx = O.block_bwd(r, c, P, grads)
        f = lambda: (O.block_fwd(x, P, 'l', kind, 2, norm, act, dropmask=mask)[0] * r).sum()
        for name, arr, g in [('l.kernel', P['l.kernel'], grads['l.kernel']), ('l' + gk, P['l' + gk], grads['l' + gk]),
                             ('l' + bk, P['l' + bk], grads['l' + bk]), ('x', x, dx)]:
            for _ in range(4):
                idx = tuple(RNG.integers(0, s) for s in arr.shape)
                old = arr[idx]
                arr[idx] = old + 1e-6
                fp = f()
                arr[idx] = old - 1e-6
                fm = f()
                arr[idx] = old
                fd = (fp - fm) / 2e-6
                assert abs(fd - g[idx]) <= 1e-5 * max(1.0, abs(fd)), (kind, name, fd, g[idx])


@pytest.mark.slow
def test_pix2pix_step_vs_torch_autograd():
    G = O.init_generator(1, seed=11, dtype=np.float64)
    D = O.init_discriminator(1, True, seed=12, dtype=np.float64)
    inp, tar = O.synthetic_pair(2, 256, 1, seed=123, dtype=np.float64)
    masks = O.dropout_masks(2, 256, seed=5, dtype=np.float64)
    losses_t, gen_t, gG_t, gD_t = TR.pix2pix_losses_and_grads(G, D, inp, tar, 100.0, masks)
    G0 = {k: v.copy() for k, v in G.items()}
    optG, optD = O.AdamTF(), O.AdamTF()
    out = O.pix2pix_train_step(G, D, optG, optD, inp, tar, 100.0, masks, True, return_grads=True)
    losses, gen, gG, gD = out[:4], out[4], out[5], out[6]
    assert np.allclose(losses, losses_t, rtol=1e-10)
    assert np.abs(gen - gen_t).max() < 1e-10
    for k in gG_t:
        assert rel(gG[k], gG_t[k]) < 1e-8, k
    for k in gD_t:
        assert rel(gD[k], gD_t[k]) < 1e-8, k
    # epoch-1 sanity band from the reference's published loss plot: total G loss O(10..60)
    assert 10 < losses[0] < 80
    # first Adam step moved every weight with a non-zero gradient by ~lr
    k = 'down3.kernel'
    moved = np.abs(G[k] - G0[k])
    assert np.all(moved[np.abs(gG[k]) > 1e-3] > 1.9e-4) and moved.max() < 2.1e-4


@pytest.mark.slow
def test_cyclegan_step_vs_torch_autograd():
    dt = np.float64
    Gg = O.init_generator(1, 'instancenorm', seed=21, dtype=dt)
    Gf = O.init_generator(1, 'instancenorm', seed=22, dtype=dt)
    Dx = O.init_discriminator(1, False, 'instancenorm', seed=23, dtype=dt)
    Dy = O.init_discriminator(1, False, 'instancenorm', seed=24, dtype=dt)
    rx, ry = O.synthetic_pair(1, 256, 1, seed=9, dtype=dt)
    keys = ['fake_y', 'cycled_x', 'fake_x', 'cycled_y', 'same_x', 'same_y']
    masks = {k: O.dropout_masks(1, 256, seed=40 + i, dtype=dt) for i, k in enumerate(keys)}
    lt, g1, g2, g3, g4 = TR.cyclegan_losses_and_grads(Gg, Gf, Dx, Dy, rx, ry, 10.0, masks)
    opts = [O.AdamTF() for _ in range(4)]
    out = O.cyclegan_train_step(Gg, Gf, Dx, Dy, opts, rx, ry, 10.0, masks, True, return_grads=True)
    assert np.allclose(out[:7], lt, rtol=1e-10)
    for mine, ref in zip(out[8:], (g1, g2, g3, g4)):
        for k in ref:
            assert rel(mine[k], ref[k]) < 1e-7, k


def test_oracle_reproduces_committed_golden():
    """tests/golden/*.npz (made by tests/golden/make_golden.py from the reference's example_images): the fp32
    oracle must reproduce the committed fp64 expected values — pins the oracle against drift."""
    import os
    g = os.path.join(os.path.dirname(__file__), 'golden')
    pairs = np.load(os.path.join(g, 'example_pairs_256.npz'))
    gold = np.load(os.path.join(g, 'golden_pix2pix_step.npz'))
    assert pairs['input_u8'].shape == (2, 256, 256, 1) and pairs['input_u8'].dtype == np.uint8
    inp, tar = O.normalize(pairs['input_u8']), O.normalize(pairs['target_u8'])
    G, D = O.init_generator(1, seed=11), O.init_discriminator(1, True, seed=12)
    masks = O.dropout_masks(2, 256, seed=5)
    out = O.pix2pix_train_step(G, D, O.AdamTF(), O.AdamTF(), inp, tar, 100.0, masks, True, return_grads=True)
    assert np.allclose(np.array(out[:4], np.float64), gold['losses'], rtol=1e-5)
    assert np.abs(out[4] - gold['gen']).max() < 1e-4
    assert np.abs(G['down3.kernel'][0, 0, :8, :8] - gold['new_G_down3_kernel_slice']).max() < 4.1e-4
