"""GPU parity of every C-ABI op against the CPU oracle (same seeded inputs), fp32 (exact-MFMA parity path)
and bf16 (fast path).  Calls go through the C ABI (ctypes), never through a torch op."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import gan_oracle as O

pytestmark = pytest.mark.gpu

TOL = {'f32': 2e-5, 'bf16': 2.5e-2, 'f16': 4e-3}   # max-abs error relative to max|ref|


@pytest.fixture(scope="module", params=['f32', 'bf16', 'f16'])
def ctx(request):
    from gan_amd.nets import Ctx
    return Ctx('cuda:0', request.param)


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def dev(ctx, x, pitch=None, c0=0):
    """numpy NHWC -> Buf with `pitch` channels, data at [c0, c0+c); returns (buf, view)."""
    from gan_amd.nets import Buf
    n, h, w, c = x.shape
    pitch = pitch or ((c + 7) // 8 * 8)
    b = Buf(ctx, n, h, w, pitch)
    b.t[..., c0:c0 + c] = torch.from_numpy(np.ascontiguousarray(x)).to(b.t.dtype).to(ctx.device)
    return b, b.view(c0, c)


def host(buf, c0=0, c=None):
    c = buf.c - c0 if c is None else c
    return buf.t[..., c0:c0 + c].float().cpu().numpy().astype(np.float64)


def q(ctx, x):
    """round inputs to the storage dtype so the oracle sees what the kernel sees"""
    if ctx.dtype in ('bf16', 'f16'):
        return torch.from_numpy(x.astype(np.float32)).to(ctx.tdtype).float().numpy().astype(np.float64)
    return x.astype(np.float32).astype(np.float64)


def prep(ctx, w):
    """fp32 Keras-layout kernel (4,4,A,B) -> (nat [16,A,B8], tr [16,B,A8]) typed device tensors"""
    A, B = w.shape[2], w.shape[3]
    m = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(ctx.device)
    nat = torch.zeros((16, A, (B + 7) // 8 * 8), dtype=ctx.tdtype, device=ctx.device)
    tr = torch.zeros((16, B, (A + 7) // 8 * 8), dtype=ctx.tdtype, device=ctx.device)
    rc = ctx.lib.gan_weights_prepare(m.data_ptr(), A, B, ctx.dt, nat.data_ptr(), tr.data_ptr(), ctx.stream())
    assert rc == 0
    torch.cuda.synchronize()
    return nat, tr


def conv_call(ctx, op, x, y, w, w_rows, stride=2, bias=None, act=0, y_f32=0):
    from gan_amd import _lib as L
    d = L.GanConvDesc(ctx.dt, stride, x, y, w.data_ptr(), w_rows, bias.data_ptr() if bias is not None else None, act, 0.3,
                      y_f32, ctx.ws_ptr, ctx.ws_bytes)
    fn = getattr(ctx.lib, {'conv_fwd': 'gan_conv2d_fwd', 'conv_dgrad': 'gan_conv2d_dgrad', 'convT_fwd': 'gan_convT2d_fwd',
                           'convT_dgrad': 'gan_convT2d_dgrad'}[op])
    rc = fn(C.byref(d), ctx.stream())
    assert rc == 0, (op, rc)
    torch.cuda.synchronize()


CONV_CASES = [  # (N, H, Cin, Cout, stride)
    (2, 16, 1, 64, 2),      # first layer, C=1 padded to 8 (K = 128)
    (2, 16, 6, 64, 2),      # D first layer C=3 target=True
    (3, 16, 64, 128, 2),
    (2, 8, 128, 256, 2),
    (16, 2, 512, 512, 2),   # bottleneck: M = 16, split-K, BM=16
    (4, 4, 512, 512, 2),    # M = 16
    (2, 32, 256, 512, 1),   # D conv4: 32 -> 31
    (2, 31, 512, 1, 1),     # D last: 31 -> 30, Cout = 1
    (1, 64, 64, 128, 2),    # M = 1024
    (8, 128, 64, 256, 2),   # M = 32768, N = 256: 256x256 tile (8 waves)
    (8, 128, 64, 128, 2),   # M = 32768, N = 128: 256x128 tile
    (5, 100, 64, 128, 2),   # M = 12500: ragged last tile
    (3, 18, 3, 64, 2),      # thin-K streaming kernel, M = 243 (ragged 16-pixel tiles), C = 3
    (3, 18, 2, 128, 2),     # thin-K with two 64-channel groups; dgrad into 2 channels (thin-N, parity form)
    (1, 13, 512, 3, 1),     # thin-N, stride 1, three output channels, ragged
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(ctx, case):
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    N, H, ci, co, s = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = q(ctx, rng.standard_normal((N, H, H, ci)))
    w = q(ctx, 0.05 * rng.standard_normal((4, 4, ci, co)))
    bias = rng.standard_normal(co).astype(np.float32) if co == 1 else None
    Ho = (H + 2 - 4) // s + 1
    ci8 = (ci + 7) // 8 * 8
    xb, _ = dev(ctx, x, pitch=ci8 + 8)               # pitch > c: exercises channel-slice views
    xv = xb.view(0, ci8)
    nat, tr = prep(ctx, w)
    # forward (bias + fp32 output for the Cout=1 logits layer; fused LeakyReLU otherwise)
    yb = Buf(ctx, N, Ho, Ho, (co + 7) // 8 * 8 + 8, torch.float32 if co == 1 else None)
    bt = torch.from_numpy(bias).to(ctx.device) if bias is not None else None
    act = 0 if co == 1 else L.ACT_LRELU
    conv_call(ctx, 'conv_fwd', xv, yb.view(8, co), tr, co, s, bt, act, 1 if co == 1 else 0)
    ref = O.conv2d_fwd(x, w, s)
    if bias is not None:
        ref = ref + bias
    else:
        ref = O.act_fwd(ref, 'lrelu')
    assert rel(host(yb, 8, co), ref) < TOL[ctx.dtype]
    assert np.all(host(yb, 0, 8) == 0)               # neighbouring channels untouched
    # dgrad / wgrad
    dy = q(ctx, rng.standard_normal((N, Ho, Ho, co)))
    dyb, dyv = dev(ctx, dy)
    dyv = dyb.view(0, (co + 7) // 8 * 8)
    dxb = Buf(ctx, N, H, H, ci8)
    conv_call(ctx, 'conv_dgrad', dyv, dxb.view(0, ci), nat, ci, s)
    dx_ref, dw_ref = O.conv2d_bwd(x, w, dy, s)
    assert rel(host(dxb, 0, ci), dx_ref) < TOL[ctx.dtype]
    dw = torch.full((16, ci, co), 7.0, dtype=torch.float32, device=ctx.device)
    d = L.GanWgradDesc(ctx.dt, s, xv, dyv, dw.data_ptr(), ci, co, 0, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
    torch.cuda.synchronize()
    assert rel(dw.cpu().numpy().reshape(4, 4, ci, co), dw_ref) < TOL[ctx.dtype]
    d.accumulate = 1                                  # dw += (cycle_gan: a net called several times)
    assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
    torch.cuda.synchronize()
    assert rel(dw.cpu().numpy().reshape(4, 4, ci, co), 2 * dw_ref) < TOL[ctx.dtype]


TAPSHARE_CASES = [  # (op, N, H of x, Cin (x channels), Cout (y channels), stride, taps per staged A tile, oracle?)
    ('conv_fwd', 8, 128, 64, 128, 2, 2, True),       # 256x128 tiles, rows of 64 positions
    ('conv_fwd', 8, 128, 64, 256, 2, 2, False),      # 256x256 tiles
    ('conv_fwd', 16, 100, 64, 128, 2, 2, True),      # rows of 50 positions: segments cut inside the 64-row blocks, ragged last tile
    ('conv_fwd', 32, 32, 256, 512, 1, 4, False),     # D conv4 at batch 16: 256x256 tiles, rows of 31, four taps per staged tile
    ('conv_fwd', 40, 32, 64, 128, 1, 4, True),       # stride 1 on 256x128 tiles
    ('conv_dgrad', 32, 31, 512, 256, 1, 4, False),   # D conv4's dgrad (shifts to the left: dstep = -1), 256x128 tiles
    ('conv_dgrad', 8, 64, 256, 128, 2, 2, True),     # stride-2 dgrad: four parity sub-GEMMs of two taps per row
    ('convT_fwd', 4, 64, 128, 128, 2, 2, True),      # transposed forward (parity form)
    ('convT_fwd', 16, 16, 1024, 256, 2, 2, False),   # rows of 16, split K
    ('convT_dgrad', 16, 64, 128, 512, 2, 2, False),  # = stride-2 convolution over dy, 256 blocks
    ('convT_dgrad', 16, 32, 256, 1024, 2, 2, False), # rows of 16, split K
]


@pytest.mark.parametrize("case", TAPSHARE_CASES)
def test_tap_shared_pingpong_equals_pingpong(ctx, case, planner_options):
    """conv_gemm_ps_kernel (the taps of a kernel row share one staged A tile) against conv_gemm_pp_kernel on the same launch: same
    products, another fp32 summation order - fp32 outputs agree to 1e-5 of the largest value, 16-bit outputs to one rounding step
    on a small fraction of the elements; the smaller shapes also against the oracle."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    if ctx.dtype == 'f32':
        pytest.skip("fp32 stays on the one-tap-per-tile kernel")
    op, N, H, cx, cy, s, sh, with_oracle = case
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    Ho = {'conv_fwd': (H + 2 - 4) // s + 1, 'conv_dgrad': H * 2 if s == 2 else H + 1, 'convT_fwd': 2 * H, 'convT_dgrad': H // 2}[op]
    x = q(ctx, rng.standard_normal((N, H, H, cx)))
    # weights in Keras layout (4, 4, in, out) of the LAYER: conv_fwd in = cx; dgrads run over the layer's output side
    kin, kout = {'conv_fwd': (cx, cy), 'conv_dgrad': (cy, cx), 'convT_fwd': (cy, cx), 'convT_dgrad': (cx, cy)}[op]
    w = q(ctx, 0.05 * rng.standard_normal((4, 4, kin, kout)))
    nat, tr = prep(ctx, w)
    wk = {'conv_fwd': tr, 'conv_dgrad': nat, 'convT_fwd': nat, 'convT_dgrad': tr}[op]
    xb, xv = dev(ctx, x)
    outs = {}
    for f32 in (0, 1):
        for share in (0, 3, 7):                      # 3: tap-shared kernel, 7: its table-driven form on the 256x128 tiles
            planner_options('conv.tap_share', share)
            yb = Buf(ctx, N, Ho, Ho, cy, torch.float32 if f32 else None)
            d = L.GanConvDesc(ctx.dt, s, xv, yb.view(0, cy), wk.data_ptr(), cy, None, 0, 0.3, f32, ctx.ws_ptr, ctx.ws_bytes)
            assert ctx.lib.gan_conv_tap_shared(C.byref(d), ['conv_fwd', 'conv_dgrad', 'convT_fwd', 'convT_dgrad'].index(op)) == (sh if share else 0)
            conv_call(ctx, op, xv, yb.view(0, cy), wk, cy, s, None, 0, f32)
            outs[f32, share] = host(yb)
    scale = np.abs(outs[1, 0]).max()
    assert scale > 0
    assert np.abs(outs[1, 3] - outs[1, 0]).max() < 1e-5 * scale
    diff = outs[0, 3] != outs[0, 0]
    assert diff.mean() < 0.02 and np.abs(outs[0, 3] - outs[0, 0]).max() <= (2.0 ** -7 if ctx.dtype == 'bf16' else 2.0 ** -10) * scale
    assert np.array_equal(outs[1, 7], outs[1, 3]) and np.array_equal(outs[0, 7], outs[0, 3])     # same arithmetic, same order
    if with_oracle:
        ref = {'conv_fwd': lambda: O.conv2d_fwd(x, w, s), 'conv_dgrad': lambda: O.conv2d_bwd(np.zeros((N, Ho, Ho, cy)), w, x, s)[0],
               'convT_fwd': lambda: O.convT2d_fwd(x, w), 'convT_dgrad': lambda: O.convT2d_bwd(np.zeros((N, Ho, Ho, cy)), w, x)[0]}[op]()
        assert rel(outs[0, 3], ref) < TOL[ctx.dtype]


CONVT_CASES = [  # (N, h, Cin, Cout)
    (16, 1, 512, 512),      # up0 at 256: 1x1 -> 2x2
    (2, 4, 1024, 512),
    (2, 16, 512, 128),
    (2, 32, 256, 64),
    (2, 32, 128, 1),        # head: Cout = 1, bias + tanh
    (1, 16, 128, 3),        # head, 3 channels
    (4, 64, 128, 128),      # M = 16384 x 4 parities: 256x128 tile
    (3, 5, 128, 2),         # thin-N head, ragged pixel tiles
    (16, 64, 64, 64),       # 64 output channels on a big map: 256x64 tile (1024 tiles)
]


@pytest.mark.parametrize("case", [('conv_fwd', 8, 128, 64, 128, 1), ('conv_fwd', 8, 128, 64, 128, 2), ('convT_fwd', 2, 64, 128, 128, 1),
                                  ('convT_fwd', 4, 64, 128, 128, 4),
                                  # split-K layers: the slab-reduce kernel emits the partials
                                  ('conv_fwd', 4, 16, 128, 256, 2), ('convT_fwd', 4, 8, 512, 256, 2), ('conv_fwd', 16, 4, 512, 512, 1),
                                  ('conv_fwd', 4, 4, 512, 512, 4),
                                  # ... of several row groups per workgroup (round 5): G.down3 / up3 / up4 forward at batch 16 - 4,096 rows x 512
                                  # channels was 2,048 chunks of two rows and fell back to a separate statistics pass
                                  ('conv_fwd', 16, 32, 256, 512, 1), ('convT_fwd', 16, 8, 1024, 512, 1), ('convT_fwd', 16, 16, 1024, 256, 2),
                                  # ONE group whose last 256-row tile is ragged (16 x 31 x 31 = 15,376 rows: the PatchGAN 31 x 31 layer's row count)
                                  # on the table-driven 256x128 tile and (stride 1, 8 x 63 x 63 rows) on the tap-shared 256x256 tile
                                  ('conv_fwd', 16, 62, 128, 512, 1), ('conv_s1', 8, 64, 256, 512, 1)])
def test_conv_epilogue_statistics_partials(ctx, case):
    """Fused normalisation statistics (GanConvDesc.stats_partial): per-channel (sum, sum of squares) of the STORED
    output, summed over the chunks the plan reports, per statistics group (tile partials from the epilogue, or row-block
    partials from the slab-reduce kernel of a split-K layer)."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    op, N, H, ci, co, groups = case
    rng = np.random.default_rng(7)
    x = q(ctx, rng.standard_normal((N, H, H, ci)))
    stride = 1 if op == 'conv_s1' else 2             # 'conv_s1': ZeroPadding2D + Conv2D(strides=1), base_gan.py:145-147 (H -> H - 1)
    if op == 'conv_s1':
        op = 'conv_fwd'
    w = q(ctx, 0.05 * rng.standard_normal((4, 4, ci, co) if op == 'conv_fwd' else (4, 4, co, ci)))
    xb, xv = dev(ctx, x)
    nat, tr = prep(ctx, w)
    Ho = H - 1 if stride == 1 else (H // 2 if op == 'conv_fwd' else 2 * H)
    yb = Buf(ctx, N, Ho, Ho, co)
    part = torch.full((4 << 20,), 9.0, dtype=torch.float32, device=ctx.device)
    d = L.GanConvDesc(ctx.dt, stride, xv, yb.view(), (tr if op == 'conv_fwd' else nat).data_ptr(), co, None, 0, 0.3, 0,
                      ctx.ws_ptr, ctx.ws_bytes, part.data_ptr(), groups, part.numel() * 4)
    opi = 0 if op == 'conv_fwd' else 2
    info = (C.c_int32 * 5)()
    assert ctx.lib.gan_conv_plan_info(C.byref(d), opi, info) == 0
    chunks = info[4]
    assert chunks > 0, "shape chosen to be fusable"
    fn = ctx.lib.gan_conv2d_fwd if op == 'conv_fwd' else ctx.lib.gan_convT2d_fwd
    assert fn(C.byref(d), ctx.stream()) == 0
    torch.cuda.synchronize()
    got = part[:groups * chunks * co * 2].cpu().numpy().reshape(groups, chunks, co, 2).astype(np.float64).sum(1)
    y = host(yb).reshape(groups, -1, co)
    assert rel(got[..., 0], y.sum(1)) < 1e-4 and rel(got[..., 1], (y * y).sum(1)) < 1e-4
    # a partial-sums region that is too small for the plan is refused, not overrun
    d.stats_partial_bytes = groups * chunks * co * 8 - 4
    assert fn(C.byref(d), ctx.stream()) == -3                 # GAN_E_WORKSPACE


def test_thin_layers_take_streaming_kernels():
    """The <= 8-channel layers of the bf16 path must run on csrc/thin.hip (plan info: BM = 0, BN = family)."""
    from gan_amd import _lib as L
    from gan_amd.nets import Ctx, Buf
    c = Ctx('cuda:0', 'bf16')
    info = (C.c_int32 * 5)()
    x8, y64, x128, y1 = Buf(c, 2, 32, 32, 8), Buf(c, 2, 16, 16, 64), Buf(c, 2, 16, 16, 128), Buf(c, 2, 32, 32, 8)
    d = L.GanConvDesc(c.dt, 2, x8.view(), y64.view(), 16, 64, None, 0, 0.3, 0, c.ws_ptr, c.ws_bytes)
    assert c.lib.gan_conv_plan_info(C.byref(d), 0, info) == 0 and (info[0], info[1]) == (0, 2)
    d = L.GanConvDesc(c.dt, 2, x128.view(), y1.view(0, 1), 16, 1, None, 0, 0.3, 0, c.ws_ptr, c.ws_bytes)
    assert c.lib.gan_conv_plan_info(C.byref(d), 2, info) == 0 and (info[0], info[1]) == (0, 1)
    assert c.lib.gan_conv_workspace_bytes(C.byref(d), 2) == 0                # one output channel: Z stays in LDS (conv_thin_n_fused_kernel)
    prev = L.set_option('conv.thin_fused', 0)
    assert c.lib.gan_conv_workspace_bytes(C.byref(d), 2) == 2 * 16 * 16 * 16 * 4    # two-kernel form: Z[pixel][c][tap] in the workspace
    L.set_option('conv.thin_fused', prev)
    c16 = Ctx('cuda:0', 'f16')
    d = L.GanConvDesc(c16.dt, 2, x8.view(), y64.view(), 16, 64, None, 0, 0.3, 0, c.ws_ptr, c.ws_bytes)
    assert c16.lib.gan_conv_plan_info(C.byref(d), 0, info) == 0 and (info[0], info[1]) == (0, 2)      # fp16 streams too
    c32 = Ctx('cuda:0', 'f32')
    d = L.GanConvDesc(c32.dt, 2, x8.view(), y64.view(), 16, 64, None, 0, 0.3, 0, c.ws_ptr, c.ws_bytes)
    assert c32.lib.gan_conv_plan_info(C.byref(d), 0, info) == 0 and info[0] > 0      # fp32 parity path: tiled kernel


FUSE_CASES = [  # (op, N, H of dy, channels of dy, channels of the produced gradient, groups, kind, cols, with skip gradient)
    ('conv_dgrad', 8, 32, 128, 64, 0, 'act', 64, True),          # down1 -> down0: LeakyReLU on the saved activation, 256x64 tile
    ('conv_dgrad', 8, 16, 256, 128, 1, 'lrelu', 128, True),      # BatchNorm encoder layer, tile epilogue
    ('conv_dgrad', 8, 16, 256, 128, 8, 'lrelu', 128, True),      # InstanceNorm: one group per image
    ('conv_dgrad', 16, 32, 256, 128, 2, 'lrelu', 128, False),    # ping-pong 256x128 tile, D's two invocations
    ('convT_dgrad', 16, 64, 128, 512, 1, 'relu', 256, False),    # decoder: leading half of a concat, ping-pong 256x256 tile
    ('convT_dgrad', 4, 64, 128, 256, 1, 'relu+mask', 128, False),
    ('convT_dgrad', 4, 8, 512, 1024, 1, 'relu+mask', 512, False),   # split-K: the slab-reduce kernel carries it
    ('conv_dgrad', 4, 4, 512, 512, 4, 'lrelu', 512, True),          # split-K, InstanceNorm, skip gradient
    ('conv_dgrad', 4, 4, 512, 64, 0, 'act', 64, False),             # split-K, activation only
    ('convT_dgrad', 16, 32, 256, 1024, 1, 'relu', 512, False),      # G.up4's dgrad at batch 16 (4,096 rows x 1,024 channels, split K): several row groups per chunk
    ('conv_dgrad', 16, 8, 512, 512, 1, 'lrelu', 512, True),         # G.down4's dgrad: 1,024 x 4 rows, split K, skip gradient
]


@pytest.mark.parametrize("case", FUSE_CASES)
def test_dgrad_fused_backward_epilogue(ctx, case, planner_options):
    """GanBwdFuse: a dgrad launch that starts the layer-below backward in its epilogue (dz + partial sums) followed by
    gan_norm_act_bwd_fused must equal the plain dgrad followed by gan_norm_act_bwd / gan_act_bwd (themselves checked
    against the oracle above); channels >= cols must be the plain dgrad's.  option conv.bwd_fuse_tile = 1 lets every tile epilogue carry
    it (the default policy keeps it to the 64-column tiles and the slab-reduce kernels, where it pays)."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    planner_options('conv.bwd_fuse_tile', 1)          # every tile epilogue carries it (planner option, include/gan_amd.h)
    op, N, H, cdy, cg, G, kind, cols, skip = case
    rng = np.random.default_rng(11)
    Hg = 2 * H if op == 'conv_dgrad' else H // 2
    dy = q(ctx, rng.standard_normal((N, H, H, cdy)))
    # Conv2D master HWIO (in = cg, out = cdy) -> native NK copy; Conv2DTranspose master (kh, kw, out = cdy, in = cg) -> transposed copy
    w = q(ctx, 0.05 * rng.standard_normal((4, 4, cg, cdy) if op == 'conv_dgrad' else (4, 4, cdy, cg)))
    nat, tr = prep(ctx, w)
    wt = nat if op == 'conv_dgrad' else tr
    wrows = cg
    dyb, dyv = dev(ctx, dy)
    ref = q(ctx, rng.standard_normal((N, Hg, Hg, cols)) * 1.3 + 0.2)
    refb, refv = dev(ctx, ref)
    addv = None
    if skip:
        addb, addv = dev(ctx, q(ctx, rng.standard_normal((N, Hg, Hg, cols))), pitch=cols + 8)
    f32 = torch.float32
    norm = kind != 'act'
    act = {'act': 'lrelu', 'lrelu': 'lrelu', 'relu': 'relu', 'relu+mask': 'relu'}[kind]
    tm = None
    if kind == 'relu+mask':
        tm = torch.from_numpy((rng.random((N, Hg, Hg, cols)) > 0.5).astype(np.uint8)).to(ctx.device)
    if norm:
        gamma = torch.from_numpy((1 + 0.2 * rng.standard_normal(cols)).astype(np.float32)).to(ctx.device)
        beta = torch.from_numpy((0.2 * rng.standard_normal(cols)).astype(np.float32)).to(ctx.device)
        mean, rstd = torch.zeros(G * cols, dtype=f32, device=ctx.device), torch.zeros(G * cols, dtype=f32, device=ctx.device)
        nd = L.GanNormDesc(ctx.dt, refv, refv, G, 1e-3, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, None,
                           0.99, None, L.ACTS[act], 0.3, ctx.ws_ptr, ctx.ws_bytes)
        assert ctx.lib.gan_norm_stats(C.byref(nd), ctx.stream()) == 0
    fn = ctx.lib.gan_conv2d_dgrad if op == 'conv_dgrad' else ctx.lib.gan_convT2d_dgrad
    opi = 1 if op == 'conv_dgrad' else 3
    z = L.GanTensor(None, 0, 0, 0, 0, 0)
    # ---- plain: dgrad, then the stand-alone layer backward
    da_b = Buf(ctx, N, Hg, Hg, cg)
    d0 = L.GanConvDesc(ctx.dt, 2, dyv, da_b.view(), wt.data_ptr(), wrows, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes)
    assert fn(C.byref(d0), ctx.stream()) == 0
    out_a = Buf(ctx, N, Hg, Hg, cols)
    dg_a, db_a = torch.zeros(cols, dtype=f32, device=ctx.device), torch.zeros(cols, dtype=f32, device=ctx.device)
    if norm:
        bd = L.GanNormBwdDesc(ctx.dt, refv, da_b.view(0, cols), addv if skip else z, out_a.view(), G, gamma.data_ptr(), beta.data_ptr(),
                              mean.data_ptr(), rstd.data_ptr(), tm.data_ptr() if tm is not None else None, L.ACTS[act], 0.3,
                              dg_a.data_ptr(), db_a.data_ptr(), 0, ctx.ws_ptr, ctx.ws_bytes)
        assert ctx.lib.gan_norm_act_bwd(C.byref(bd), ctx.stream()) == 0
    else:
        ad = L.GanActBwdDesc(ctx.dt, refv, da_b.view(0, cols), addv if skip else z, out_a.view(), L.ACTS[act], 0.3, None, 0,
                             ctx.ws_ptr, ctx.ws_bytes)
        assert ctx.lib.gan_act_bwd(C.byref(ad), ctx.stream()) == 0
    torch.cuda.synchronize()
    # ---- fused
    dz_b = Buf(ctx, N, Hg, Hg, cg)
    part = torch.full((4 << 20,), 7.0, dtype=f32, device=ctx.device)
    bf = L.GanBwdFuse(refv, addv if skip else z, mean.data_ptr() if norm else None, rstd.data_ptr() if norm else None,
                      gamma.data_ptr() if norm else None, beta.data_ptr() if norm else None, tm.data_ptr() if tm is not None else None,
                      cols, L.ACTS[act], 0.3, cols)
    d1 = L.GanConvDesc(ctx.dt, 2, dyv, dz_b.view(), wt.data_ptr(), wrows, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes,
                       part.data_ptr() if norm else None, G if norm else 0, part.numel() * 4, C.addressof(bf))
    info = (C.c_int32 * 5)()
    assert ctx.lib.gan_conv_plan_info(C.byref(d1), opi, info) == 0
    chunks = info[4]
    assert chunks > 0, ("shape chosen to be fusable", list(info))
    assert fn(C.byref(d1), ctx.stream()) == 0
    tol = {'f32': 2e-5, 'bf16': 2e-2, 'f16': 3e-3}[ctx.dtype]
    if norm:
        out_f = Buf(ctx, N, Hg, Hg, cols)
        dg_f, db_f = torch.zeros(cols, dtype=f32, device=ctx.device), torch.zeros(cols, dtype=f32, device=ctx.device)
        fd = L.GanNormBwdDesc(ctx.dt, refv, dz_b.view(0, cols), z, out_f.view(), G, gamma.data_ptr(), beta.data_ptr(),
                              mean.data_ptr(), rstd.data_ptr(), None, 0, 0.3, dg_f.data_ptr(), db_f.data_ptr(), 0,
                              part.data_ptr(), part.numel() * 4)
        assert ctx.lib.gan_norm_act_bwd_fused(C.byref(fd), chunks, ctx.stream()) == 0
        torch.cuda.synchronize()
        assert rel(host(out_f), host(out_a)) < tol
        assert rel(dg_f.cpu().numpy(), dg_a.cpu().numpy().astype(np.float64)) < max(tol, 1e-4)
        assert rel(db_f.cpu().numpy(), db_a.cpu().numpy().astype(np.float64)) < max(tol, 1e-4)
    else:
        torch.cuda.synchronize()
        assert rel(host(dz_b, 0, cols), host(out_a)) < tol
    if cols < cg:            # the skip half of a decoder concat is the plain gradient
        assert np.array_equal(host(dz_b, cols), host(da_b, cols))
    # a launch shape that cannot carry the epilogue refuses it instead of silently dropping it
    if norm and G == 1:
        d2 = L.GanConvDesc(ctx.dt, 2, dyv, dz_b.view(), wt.data_ptr(), wrows, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes,
                           part.data_ptr(), 3, part.numel() * 4, C.addressof(bf))      # 3 groups do not divide the batch
        assert ctx.lib.gan_conv_plan_info(C.byref(d2), opi, info) == 0 and info[4] == 0
        assert fn(C.byref(d2), ctx.stream()) == -2            # GAN_E_SHAPE
        torch.cuda.synchronize()


@pytest.mark.parametrize("case", CONVT_CASES)
def test_convT2d_fwd_dgrad_wgrad(ctx, case):
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    N, h, ci, co = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = q(ctx, rng.standard_normal((N, h, h, ci)))
    w = q(ctx, 0.05 * rng.standard_normal((4, 4, co, ci)))
    head = co < 8
    bias = (0.1 * rng.standard_normal(co)).astype(np.float32) if head else None
    xb, xv = dev(ctx, x)
    nat, tr = prep(ctx, w)
    co8 = (co + 7) // 8 * 8
    yb = Buf(ctx, N, 2 * h, 2 * h, co8)
    bt = torch.from_numpy(bias).to(ctx.device) if head else None
    conv_call(ctx, 'convT_fwd', xv, yb.view(0, co), nat, co, 2, bt, L.ACT_TANH if head else 0)
    ref = O.convT2d_fwd(x, w)
    if head:
        ref = np.tanh(ref + bias)
    assert rel(host(yb, 0, co), ref) < TOL[ctx.dtype]
    dy = q(ctx, rng.standard_normal((N, 2 * h, 2 * h, co)))
    dyb, _ = dev(ctx, dy)
    dyv = dyb.view(0, co8)
    dxb = Buf(ctx, N, h, h, ci)
    conv_call(ctx, 'convT_dgrad', dyv, dxb.view(), tr, ci, 2)
    dx_ref, dw_ref = O.convT2d_bwd(x, w, dy)
    assert rel(host(dxb), dx_ref) < TOL[ctx.dtype]
    dw = torch.zeros((16, co, ci), dtype=torch.float32, device=ctx.device)
    d = L.GanWgradDesc(ctx.dt, 2, dyv, xv, dw.data_ptr(), co, ci, 0, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
    torch.cuda.synchronize()
    assert rel(dw.cpu().numpy().reshape(4, 4, co, ci), dw_ref) < TOL[ctx.dtype]


PAR_CASES = [  # (N, h of the coarse grid, Cin, Cout): stride-2 transposed conv h -> 2h and the stride-2 conv dgrad of the same shape
    (2, 16, 32, 64),       # Wg = 16: a tile = one whole image (16 rows), one K chunk
    (1, 32, 64, 64),       # Wg = 32: 8 rows per tile, 4 tiles, two chunks (patch double buffer)
    (1, 64, 128, 64),      # Wg = 64: 4 rows per tile, four chunks (ring wrap), halo rows between tiles
    (1, 128, 32, 64),      # Wg = 128: 2 rows per tile (512^2 images), 5 patch pieces per wave
    (2, 32, 64, 128),      # two 64-channel column blocks
]


@pytest.mark.parametrize("case", PAR_CASES)
def test_parity_patch_kernel(ctx, case, planner_options):
    """conv_par_kernel (all four output parities of 256 grid positions from one staged input patch): Conv2DTranspose forward
    (base_gan.py:106-110) and the input gradient of a stride-2 Conv2D (base_gan.py:77-79 under the tape) against the oracle,
    with bias + activation, fused statistics partials, and the fused backward epilogue equal to the four-sub-GEMM kernels."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    if ctx.dtype == 'f32':
        pytest.skip("the fp32 parity path keeps the tap-gather kernel")
    planner_options('conv.parity_patch_min_blocks', 1)
    planner_options('conv.parity_patch_max_n', 128)
    N, h, ci, co = case
    rng = np.random.default_rng(hash(case) % 2**31)
    info = (C.c_int32 * 5)()
    # ---- Conv2DTranspose forward: x [N,h,h,ci] -> y [N,2h,2h,co], bias + LeakyReLU, statistics partials of the stored output
    x = q(ctx, rng.standard_normal((N, h, h, ci)))
    w = q(ctx, 0.05 * rng.standard_normal((4, 4, co, ci)))
    bias = (0.1 * rng.standard_normal(co)).astype(np.float32)
    xb, _ = dev(ctx, x, pitch=ci + 8)                 # channel-slice view of a wider buffer
    xv = xb.view(0, ci)
    nat, tr = prep(ctx, w)
    yb = Buf(ctx, N, 2 * h, 2 * h, co + 8)
    bt = torch.from_numpy(bias).to(ctx.device)
    part = torch.full((1 << 20,), 9.0, dtype=torch.float32, device=ctx.device)
    groups = N
    d = L.GanConvDesc(ctx.dt, 2, xv, yb.view(8, co), nat.data_ptr(), co, bt.data_ptr(), L.ACT_LRELU, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes,
                      part.data_ptr(), groups, part.numel() * 4)
    assert ctx.lib.gan_conv_plan_info(C.byref(d), 2, info) == 0
    assert (info[0], info[1], info[2], info[3]) == (1024, 64, 1, 4), list(info)
    chunks = info[4]
    assert chunks == (h * h // 256) * 4
    assert ctx.lib.gan_convT2d_fwd(C.byref(d), ctx.stream()) == 0
    torch.cuda.synchronize()
    ref = O.act_fwd(O.convT2d_fwd(x, w) + bias, 'lrelu')
    assert rel(host(yb, 8, co), ref) < TOL[ctx.dtype]
    assert np.all(host(yb, 0, 8) == 0)
    got = part[:groups * chunks * co * 2].cpu().numpy().reshape(groups, chunks, co, 2).astype(np.float64).sum(1)
    ys = host(yb, 8, co).reshape(groups, -1, co)
    assert rel(got[..., 0], ys.sum(1)) < 1e-4 and rel(got[..., 1], (ys * ys).sum(1)) < 1e-4
    # ---- stride-2 Conv2D input gradient: dy [N,h,h,ci'] (ci' = this case's Cin) -> dx [N,2h,2h,co]
    wc = q(ctx, 0.05 * rng.standard_normal((4, 4, co, ci)))          # HWIO: in = co (the gradient's channels), out = ci
    dy = q(ctx, rng.standard_normal((N, h, h, ci)))
    natc, trc = prep(ctx, wc)
    dyb, dyv = dev(ctx, dy)
    dxb = Buf(ctx, N, 2 * h, 2 * h, co)
    d2 = L.GanConvDesc(ctx.dt, 2, dyv, dxb.view(), natc.data_ptr(), co, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_conv_plan_info(C.byref(d2), 1, info) == 0 and info[0] == 1024
    assert ctx.lib.gan_conv2d_dgrad(C.byref(d2), ctx.stream()) == 0
    torch.cuda.synchronize()
    dx_ref, _ = O.conv2d_bwd(np.zeros((N, 2 * h, 2 * h, co)), wc, dy, 2)
    assert rel(host(dxb), dx_ref) < TOL[ctx.dtype]
    # ---- fused backward epilogue on this kernel == on the four-sub-GEMM kernel (every tile epilogue carries it)
    planner_options('conv.bwd_fuse_tile', 1)
    refy = q(ctx, rng.standard_normal((N, 2 * h, 2 * h, co)) * 1.3 + 0.2)
    refb, refv = dev(ctx, refy)
    addb, addv = dev(ctx, q(ctx, rng.standard_normal((N, 2 * h, 2 * h, co))), pitch=co + 8)
    f32 = torch.float32
    gamma = torch.from_numpy((1 + 0.2 * rng.standard_normal(co)).astype(np.float32)).to(ctx.device)
    beta = torch.from_numpy((0.2 * rng.standard_normal(co)).astype(np.float32)).to(ctx.device)
    mean, rstd = torch.zeros(co, dtype=f32, device=ctx.device), torch.zeros(co, dtype=f32, device=ctx.device)
    nd = L.GanNormDesc(ctx.dt, refv, refv, 1, 1e-3, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, None,
                       0.99, None, L.ACT_LRELU, 0.3, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_norm_stats(C.byref(nd), ctx.stream()) == 0
    bf = L.GanBwdFuse(refv, addv, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 0, L.ACT_LRELU, 0.3, co)
    outs = []
    for use_par in (1, 0):
        planner_options('conv.parity_patch', use_par)
        dzb = Buf(ctx, N, 2 * h, 2 * h, co)
        pf = torch.full((1 << 20,), 7.0, dtype=f32, device=ctx.device)
        d3 = L.GanConvDesc(ctx.dt, 2, dyv, dzb.view(), natc.data_ptr(), co, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes,
                           pf.data_ptr(), 1, pf.numel() * 4, C.addressof(bf))
        assert ctx.lib.gan_conv_plan_info(C.byref(d3), 1, info) == 0 and info[4] > 0 and (info[0] == 1024) == bool(use_par)
        assert ctx.lib.gan_conv2d_dgrad(C.byref(d3), ctx.stream()) == 0
        torch.cuda.synchronize()
        sums = pf[:info[4] * co * 2].cpu().numpy().reshape(info[4], co, 2).astype(np.float64).sum(0)
        outs.append((host(dzb), sums))
    assert rel(outs[0][0], outs[1][0]) < {'bf16': 2e-2, 'f16': 3e-3}[ctx.dtype]
    assert rel(outs[0][1], outs[1][1]) < {'bf16': 2e-2, 'f16': 3e-3}[ctx.dtype]


def test_parity_patch_kernel_at_baseline_shape(planner_options):
    """Generator up6 forward at BASELINE's batch 16 (16 x 64 x 64 x 256 -> 128 x 128 x 64, bf16): the parity-patch kernel
    against the four-sub-GEMM kernel on the same data (the oracle takes minutes at this size)."""
    from gan_amd import _lib as L
    from gan_amd.nets import Ctx, Buf
    c = Ctx('cuda:0', 'bf16')
    g = torch.Generator(device='cuda').manual_seed(3)
    x = Buf(c, 16, 64, 64, 256)
    x.t.copy_(torch.randn(x.t.shape, device='cuda', generator=g).to(c.tdtype))
    w = (0.05 * torch.randn((16, 64, 256), device='cuda', generator=g)).to(c.tdtype)
    info = (C.c_int32 * 5)()
    res = []
    for use_par in (1, 0):
        planner_options('conv.parity_patch', use_par)
        y = Buf(c, 16, 128, 128, 64)
        d = L.GanConvDesc(c.dt, 2, x.view(), y.view(), w.data_ptr(), 64, None, 0, 0.3, 0, c.ws_ptr, c.ws_bytes)
        assert c.lib.gan_conv_plan_info(C.byref(d), 2, info) == 0 and (info[0] == 1024) == bool(use_par), list(info)
        assert c.lib.gan_convT2d_fwd(C.byref(d), c.stream()) == 0
        torch.cuda.synchronize()
        res.append(y.t.float())
    err = float((res[0] - res[1]).abs().max() / res[1].abs().max())
    assert err < 1e-2, err          # same products, different summation order; bf16 output rounding




NORMFUSE_CASES = [  # (op, N, H of x, Cin, Cout, groups, act, dropout)
    ('conv_fwd', 16, 8, 512, 512, 1, 'lrelu', False),     # generator down5 at batch 16: M = 256, BatchNorm over the batch
    ('conv_fwd', 16, 16, 512, 512, 1, 'lrelu', False),    # down4: M = 1024 (8 rows per thread)
    ('conv_fwd', 4, 4, 512, 512, 4, 'lrelu', False),      # InstanceNorm: one group per image, 4 rows each
    ('conv_fwd', 2, 2, 512, 512, 2, 'lrelu', False),      # 1 x 1 maps: one row per group
    ('convT_fwd', 16, 2, 512, 512, 1, 'relu', True),      # up1: four parities, dropout
    ('convT_fwd', 4, 8, 1024, 512, 2, 'relu', False),     # two BatchNorm calls batched (moving averages in call order)
]


def _normfuse_fwd(ctx, case, planner_options, variants):
    """One GanNormFuse forward layer under each option set of `variants` ({option: value}; 'conv.norm_fuse' = 0: the four-launch path):
    (y, activation, mean, rstd, moving mean, moving variance, pad channels) per variant."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    op, N, H, ci, co, G, act, drop = case
    rng = np.random.default_rng(23)
    x = q(ctx, rng.standard_normal((N, H, H, ci)))
    w = q(ctx, 0.05 * rng.standard_normal((4, 4, ci, co) if op == 'conv_fwd' else (4, 4, co, ci)))
    xb, xv = dev(ctx, x)
    nat, tr = prep(ctx, w)
    Ho = H // 2 if op == 'conv_fwd' else 2 * H
    f32 = torch.float32
    gamma = torch.from_numpy((1 + 0.2 * rng.standard_normal(co)).astype(np.float32)).to(ctx.device)
    beta = torch.from_numpy((0.2 * rng.standard_normal(co)).astype(np.float32)).to(ctx.device)
    tm = torch.from_numpy((rng.random((N, Ho, Ho, co)) > 0.5).astype(np.uint8)).to(ctx.device) if drop else None
    fn = ctx.lib.gan_conv2d_fwd if op == 'conv_fwd' else ctx.lib.gan_convT2d_fwd
    opi = 0 if op == 'conv_fwd' else 2
    info = (C.c_int32 * 5)()
    res = []
    for opts in variants:
        fuse = opts.get('conv.norm_fuse', 1)
        own = opts.get('conv.own_max_rows', 0)
        for k, v in opts.items():
            planner_options(k, v)
        yb, ab = Buf(ctx, N, Ho, Ho, co), Buf(ctx, N, Ho, Ho, co + 8)
        mean, rstd = torch.zeros(G * co, dtype=f32, device=ctx.device), torch.zeros(G * co, dtype=f32, device=ctx.device)
        mm, mv = torch.zeros(co, dtype=f32, device=ctx.device), torch.ones(co, dtype=f32, device=ctx.device)
        part = torch.zeros(1 << 20, dtype=f32, device=ctx.device)
        nf = L.GanNormFuse(ab.view(8, co), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                           1e-3, 0.99, tm.data_ptr() if drop else None, L.ACTS[act], 0.3, None, None, 0)
        d = L.GanConvDesc(ctx.dt, 2, xv, yb.view(), (tr if op == 'conv_fwd' else nat).data_ptr(), co, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes,
                          part.data_ptr(), G, part.numel() * 4, None, C.addressof(nf))
        assert ctx.lib.gan_conv_plan_info(C.byref(d), opi, info) == 0
        if own:
            assert info[0] == 0 and info[1] == 8 and info[2] == 1 and info[4] == -1, ("shape chosen for the column-owner kernel", list(info))
        else:
            assert info[2] > 1, "shape chosen to be split-K"
        assert (info[4] == -1) == bool(fuse), list(info)
        if not fuse:          # a request the plan cannot honour is refused (the caller would otherwise skip the layer's normalisation)
            assert fn(C.byref(d), ctx.stream()) == L.E_SHAPE
            d.norm_fuse = None
        assert fn(C.byref(d), ctx.stream()) == 0
        if not fuse:
            nd = L.GanNormDesc(ctx.dt, yb.view(), ab.view(8, co), G, 1e-3, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                               mm.data_ptr(), mv.data_ptr(), 0.99, tm.data_ptr() if drop else None, L.ACTS[act], 0.3, part.data_ptr(), part.numel() * 4)
            if info[4] > 0:
                assert ctx.lib.gan_norm_stats_finalize(C.byref(nd), info[4], ctx.stream()) == 0
            else:
                nd.workspace, nd.workspace_bytes = ctx.ws_ptr, ctx.ws_bytes
                assert ctx.lib.gan_norm_stats(C.byref(nd), ctx.stream()) == 0
            assert ctx.lib.gan_norm_act_fwd(C.byref(nd), ctx.stream()) == 0
        torch.cuda.synchronize()
        res.append((host(yb), host(ab, 8, co), mean.cpu().numpy(), rstd.cpu().numpy(), mm.cpu().numpy(), mv.cpu().numpy(), host(ab, 0, 8)))
    return res


@pytest.mark.parametrize("case", NORMFUSE_CASES)
def test_split_k_layer_finished_by_its_slab_reduce(ctx, case, planner_options):
    """GanNormFuse, forward: conv -> BN|IN -> [dropout] -> activation (base_gan.py:77-87, :106-120) of a small split-K layer in
    two launches (GEMM + slab reduce that also normalises) equals the four-launch path (slab reduce, statistics finalize,
    apply), which the tests above check against the oracle."""
    a, b = _normfuse_fwd(ctx, case, planner_options, [{'conv.norm_fuse': 1, 'conv.own_max_rows': 0}, {'conv.norm_fuse': 0, 'conv.own_max_rows': 0}])
    assert np.array_equal(a[0], b[0])                       # y: the same slab sums
    assert rel(a[2], b[2].astype(np.float64)) < 1e-5 and rel(a[3], b[3].astype(np.float64)) < 1e-5
    assert rel(a[4], b[4].astype(np.float64)) < 1e-5 and rel(a[5], b[5].astype(np.float64)) < 1e-5
    assert rel(a[1], b[1]) < {'f32': 1e-5, 'bf16': 1e-2, 'f16': 2e-3}[ctx.dtype]     # (an ulp of the storage type where rstd differs in its last bit)
    assert np.all(a[6] == 0)


OWN_FWD_CASES = [  # (op, N, H of x, Cin, Cout, groups, act, dropout)
    ('conv_fwd', 16, 2, 512, 512, 1, 'lrelu', False),     # generator down7 at batch 16: 2x2 -> 1x1, M = 16, 12 of the 16 taps never meet the map
    ('conv_fwd', 16, 4, 512, 512, 1, 'lrelu', False),     # down6: M = 64
    ('convT_fwd', 16, 1, 512, 512, 1, 'relu', True),      # up0: 1x1 -> 2x2, four parities of 16 rows, one live tap each, dropout
    ('convT_fwd', 16, 2, 1024, 512, 1, 'relu', True),     # up1: four parities of 64 rows
    ('conv_fwd', 4, 4, 512, 256, 4, 'lrelu', False),      # InstanceNorm: 4 groups of 4 rows
    ('conv_fwd', 3, 4, 64, 64, 1, 'lrelu', False),        # ragged M = 12; 2 K steps per tap (fewer than waves)
    ('convT_fwd', 1, 2, 256, 128, 1, 'relu', False),      # CycleGAN at batch 1: four parities of 4 rows
    ('convT_fwd', 2, 2, 512, 512, 2, 'relu', False),      # two BatchNorm calls batched (moving averages in call order)
]


@pytest.mark.parametrize("case", OWN_FWD_CASES)
def test_column_owner_kernel_equals_split_k_forward(ctx, case, planner_options):
    """conv_own_kernel (one launch: a workgroup owns 8 output channels of every row - gather GEMM straight from global memory, K split
    over its waves, statistics, moving averages, normalise, dropout, activation) against the split-K GEMM + finishing slab reduce on
    the same data: the same products in another summation order."""
    if ctx.dtype == 'f32':
        pytest.skip("column-owner kernel: 16-bit storage only")
    a, b = _normfuse_fwd(ctx, case, planner_options, [{'conv.own_max_rows': 64, 'conv.own_max_kb': 1 << 20}, {'conv.own_max_rows': 0}])
    tol = {'bf16': 1e-2, 'f16': 2e-3}[ctx.dtype]
    assert float(np.abs(b[1]).max()) > 0.1
    assert rel(a[0], b[0]) < tol and rel(a[1], b[1]) < 2 * tol
    for k in (2, 3, 4, 5):
        assert rel(a[k], b[k].astype(np.float64)) < tol, k
    assert np.all(a[6] == 0)


@pytest.mark.parametrize("N,G", [(16, 1), (2, 2), (4, 4)])
def test_layer_stack_equals_separate_launches(ctx, N, G, planner_options):
    """gan_conv_stack_*: three consecutive small split-K layers (conv s2 -> conv s2 -> transposed conv, each finished by its slab
    reduce: statistics, moving averages, normalise, [dropout,] activation) from ONE persistent launch - grid barriers between the
    phases, slabs and activations exchanged around the per-XCD L2s - against the same three layers as separate launches: y, the
    activations, mean / rstd and the moving averages bit for bit (the same arithmetic in the same order).  BatchNorm over the batch
    (G = 1), two BatchNorm invocations batched along N (G = 2), InstanceNorm (G = N)."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    planner_options('conv.own_max_rows', 0)        # (the stack is made of split-K layers)
    rng = np.random.default_rng(31)
    c = 512
    x = q(ctx, rng.standard_normal((N, 8, 8, c)))
    ws = [q(ctx, 0.05 * rng.standard_normal((4, 4, c, c))) for _ in range(3)]
    nk = [prep(ctx, w) for w in ws]
    f32 = torch.float32
    gam = [torch.from_numpy((1 + 0.2 * rng.standard_normal(c)).astype(np.float32)).to(ctx.device) for _ in range(3)]
    bet = [torch.from_numpy((0.2 * rng.standard_normal(c)).astype(np.float32)).to(ctx.device) for _ in range(3)]
    mask = torch.from_numpy((rng.random((N, 4, 4, c)) > 0.5).astype(np.uint8)).to(ctx.device)
    shapes = [(8, 4, 'conv_fwd'), (4, 2, 'conv_fwd'), (2, 4, 'convT_fwd')]
    results = []
    for stacked in (False, True):
        xb, xv = dev(ctx, x)
        ybs = [Buf(ctx, N, ho, ho, c) for _, ho, _ in shapes]
        abs_ = [Buf(ctx, N, ho, ho, c + 8) for _, ho, _ in shapes]
        stats = [[torch.zeros(G * c, dtype=f32, device=ctx.device) for _ in range(2)] + [torch.zeros(c, dtype=f32, device=ctx.device),
                                                                                      torch.ones(c, dtype=f32, device=ctx.device)] for _ in range(3)]
        part = torch.zeros(1 << 20, dtype=f32, device=ctx.device)
        descs, keep, opis = [], [], []
        src = xv
        for k, (hi, ho, op) in enumerate(shapes):
            mean, rstd, mm, mv = stats[k]
            nf = L.GanNormFuse(abs_[k].view(8, c), gam[k].data_ptr(), bet[k].data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                               mm.data_ptr() if G == 1 else None, mv.data_ptr() if G == 1 else None, 1e-3, 0.99,
                               mask.data_ptr() if k == 2 else None, L.ACTS['relu' if k == 2 else 'lrelu'], 0.3, None, None, 0)
            w = nk[k][1] if op == 'conv_fwd' else nk[k][0]
            d = L.GanConvDesc(ctx.dt, 2, src, ybs[k].view(), w.data_ptr(), c, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes,
                              part.data_ptr(), G, part.numel() * 4, None, C.addressof(nf))
            opi = 0 if op == 'conv_fwd' else 2
            info = (C.c_int32 * 5)()
            assert ctx.lib.gan_conv_plan_info(C.byref(d), opi, info) == 0 and info[4] == -1 and info[2] > 1, list(info)
            descs.append(d); keep.append(nf); opis.append(opi)
            src = abs_[k].view(8, c)
        if not stacked:
            for d, opi in zip(descs, opis):
                fn = ctx.lib.gan_conv2d_fwd if opi == 0 else ctx.lib.gan_convT2d_fwd
                assert fn(C.byref(d), ctx.stream()) == 0
        else:
            n = len(descs)
            nb = ctx.lib.gan_conv_stack_plan_bytes(n)
            hplan = C.create_string_buffer(nb)
            arr = (C.c_void_p * n)(*[C.addressof(d) for d in descs])
            assert ctx.lib.gan_conv_stack_plan(arr, (C.c_int32 * n)(*opis), n, hplan, nb) == 0
            devp = torch.frombuffer(bytearray(hplan.raw), dtype=torch.uint8).to(ctx.device)
            bar = torch.zeros(ctx.lib.gan_conv_stack_barrier_bytes(), dtype=torch.uint8, device=ctx.device)
            err = torch.zeros(1, dtype=torch.int32, device=ctx.device)
            for rep in range(2):          # twice: the barrier state carries over from launch to launch (moving averages advance twice)
                assert ctx.lib.gan_conv_stack_launch(C.addressof(hplan), devp.data_ptr(), bar.data_ptr(), err.data_ptr(), ctx.stream()) == 0
            torch.cuda.synchronize()
            assert int(err.item()) == 0
        if not stacked:                  # (the same two passes for the reference)
            for d, opi in zip(descs, opis):
                fn = ctx.lib.gan_conv2d_fwd if opi == 0 else ctx.lib.gan_convT2d_fwd
                assert fn(C.byref(d), ctx.stream()) == 0
        torch.cuda.synchronize()
        results.append([host(b) for b in ybs] + [host(b, 8, c) for b in abs_] + [t.cpu().numpy() for st_ in stats for t in st_] + [host(b, 0, 8) for b in abs_])
    a, b = results
    assert float(np.abs(a[5]).max()) > 0.1                    # the last activation is alive
    for i, (u, v) in enumerate(zip(a, b)):
        assert np.array_equal(u, v), i
    for pad in b[-3:]:
        assert np.all(pad == 0)                               # channel slices outside the views untouched


def _normfuse_bwd(ctx, case, planner_options, variants):
    """One GanNormFuse dgrad under each option set of `variants`: (dy of the layer below, dgamma, dbeta, skip half of dz) per variant."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    op, N, H, cdy, cg, G, kind, cols = case
    rng = np.random.default_rng(29)
    Hg = 2 * H if op == 'conv_dgrad' else H // 2
    dy = q(ctx, rng.standard_normal((N, H, H, cdy)))
    w = q(ctx, 0.05 * rng.standard_normal((4, 4, cg, cdy) if op == 'conv_dgrad' else (4, 4, cdy, cg)))
    nat, tr = prep(ctx, w)
    wt = nat if op == 'conv_dgrad' else tr
    dyb, dyv = dev(ctx, dy)
    ref = q(ctx, rng.standard_normal((N, Hg, Hg, cols)) * 1.3 + 0.2)
    refb, refv = dev(ctx, ref)
    addb, addv = dev(ctx, q(ctx, rng.standard_normal((N, Hg, Hg, cols))), pitch=cols + 8)
    f32 = torch.float32
    act = 'lrelu' if kind == 'lrelu' else 'relu'
    tm = torch.from_numpy((rng.random((N, Hg, Hg, cols)) > 0.5).astype(np.uint8)).to(ctx.device) if kind == 'relu+mask' else None
    gamma = torch.from_numpy((1 + 0.2 * rng.standard_normal(cols)).astype(np.float32)).to(ctx.device)
    beta = torch.from_numpy((0.2 * rng.standard_normal(cols)).astype(np.float32)).to(ctx.device)
    mean, rstd = torch.zeros(G * cols, dtype=f32, device=ctx.device), torch.zeros(G * cols, dtype=f32, device=ctx.device)
    nd = L.GanNormDesc(ctx.dt, refv, refv, G, 1e-3, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, None,
                       0.99, None, L.ACTS[act], 0.3, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_norm_stats(C.byref(nd), ctx.stream()) == 0
    fn = ctx.lib.gan_conv2d_dgrad if op == 'conv_dgrad' else ctx.lib.gan_convT2d_dgrad
    opi = 1 if op == 'conv_dgrad' else 3
    z = L.GanTensor(None, 0, 0, 0, 0, 0)
    bf = L.GanBwdFuse(refv, addv, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), tm.data_ptr() if tm is not None else None,
                      cols, L.ACTS[act], 0.3, cols)
    info = (C.c_int32 * 5)()
    res = []
    for opts in variants:
        fuse = opts.get('conv.norm_fuse', 1)
        own = opts.get('conv.own_max_rows', 0)
        for k, v in opts.items():
            planner_options(k, v)
        dzb, outb = Buf(ctx, N, Hg, Hg, cg), Buf(ctx, N, Hg, Hg, cols)
        dg, db = torch.full((cols,), 0.5, dtype=f32, device=ctx.device), torch.full((cols,), -0.25, dtype=f32, device=ctx.device)
        part = torch.zeros(4 << 20, dtype=f32, device=ctx.device)
        nf = L.GanNormFuse(outb.view(), None, None, None, None, None, None, 0.0, 0.0, None, 0, 0.3, dg.data_ptr(), db.data_ptr(), 1)
        d = L.GanConvDesc(ctx.dt, 2, dyv, dzb.view(), wt.data_ptr(), cg, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes,
                          part.data_ptr(), G, part.numel() * 4, C.addressof(bf), C.addressof(nf))
        assert ctx.lib.gan_conv_plan_info(C.byref(d), opi, info) == 0
        assert (info[2] == 1 and info[1] == 8 if own else info[2] > 1) and (info[4] == -1) == bool(fuse) and info[4] != 0, list(info)
        if not fuse:
            assert fn(C.byref(d), ctx.stream()) == L.E_SHAPE
            d.norm_fuse = None
        assert fn(C.byref(d), ctx.stream()) == 0
        if not fuse:
            fd = L.GanNormBwdDesc(ctx.dt, refv, dzb.view(0, cols), z, outb.view(), G, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                  rstd.data_ptr(), None, 0, 0.3, dg.data_ptr(), db.data_ptr(), 1, part.data_ptr(), part.numel() * 4)
            assert ctx.lib.gan_norm_act_bwd_fused(C.byref(fd), info[4], ctx.stream()) == 0
        torch.cuda.synchronize()
        res.append((host(outb), dg.cpu().numpy().astype(np.float64), db.cpu().numpy().astype(np.float64), host(dzb, cols) if cols < cg else None))
    return res, cols < cg


@pytest.mark.parametrize("case", [('convT_dgrad', 16, 16, 512, 1024, 1, 'relu+mask', 512), ('convT_dgrad', 16, 4, 512, 1024, 1, 'relu', 512),
                                  ('conv_dgrad', 16, 4, 512, 512, 1, 'lrelu', 512), ('conv_dgrad', 2, 4, 512, 512, 2, 'lrelu', 512),
                                  ('convT_dgrad', 2, 16, 512, 512, 2, 'relu', 512)])
def test_split_k_dgrad_finishes_the_layer_below(ctx, case, planner_options):
    """GanNormFuse, backward: the slab reduce of a small split-K dgrad writes dy of the layer below (and dgamma, dbeta; the skip
    half of a decoder concat unchanged) = the fused-epilogue path (dz + partial sums, finalize, apply) checked above."""
    (a, b), skip = _normfuse_bwd(ctx, case, planner_options, [{'conv.norm_fuse': 1, 'conv.own_max_rows': 0}, {'conv.norm_fuse': 0, 'conv.own_max_rows': 0}])
    tol = {'f32': 2e-5, 'bf16': 2e-2, 'f16': 3e-3}[ctx.dtype]
    assert rel(a[0], b[0]) < tol
    assert rel(a[1], b[1]) < max(tol, 1e-4) and rel(a[2], b[2]) < max(tol, 1e-4)
    if skip:
        assert np.array_equal(a[3], b[3])                   # skip half: the plain gradient


@pytest.mark.parametrize("min_rows", [257, 513])
def test_finishing_reduce_on_512_threads(ctx, min_rows, planner_options):
    """conv.skn512_min_rows: the slab reduce that finishes a split-K layer (GanNormFuse) on 512 threads - half the rows per thread for
    the 1,024-row (and, at 257, the 512-row) groups.  Same slab sums; the statistics are added in another (fixed) order."""
    tol = {'f32': 2e-5, 'bf16': 2e-2, 'f16': 3e-3}[ctx.dtype]
    for case in [('conv_fwd', 16, 16, 512, 512, 1, 'lrelu', False), ('convT_fwd', 4, 8, 1024, 512, 2, 'relu', False),
                 ('convT_fwd', 16, 4, 512, 512, 1, 'relu', True)]:        # 1,024 rows; two groups of 4 x 128; four parities of 256 rows + dropout
        a, b = _normfuse_fwd(ctx, case, planner_options, [{'conv.skn512_min_rows': min_rows, 'conv.own_max_rows': 0},
                                                           {'conv.skn512_min_rows': 0, 'conv.own_max_rows': 0}])
        assert np.array_equal(a[0], b[0])                   # y: the same slab sums
        for i in (2, 3, 4, 5):
            assert rel(a[i], b[i].astype(np.float64)) < 1e-5
        assert rel(a[1], b[1]) < {'f32': 1e-5, 'bf16': 1e-2, 'f16': 2e-3}[ctx.dtype]
    for case in [('convT_dgrad', 16, 16, 512, 1024, 1, 'relu+mask', 512), ('conv_dgrad', 16, 4, 512, 512, 1, 'lrelu', 512)]:       # 1,024 rows; four parities of 256
        (a, b), skip = _normfuse_bwd(ctx, case, planner_options, [{'conv.skn512_min_rows': min_rows, 'conv.own_max_rows': 0},
                                                                   {'conv.skn512_min_rows': 0, 'conv.own_max_rows': 0}])
        assert float(np.abs(b[0]).max()) > 1e-3
        assert rel(a[0], b[0]) < tol
        assert rel(a[1], b[1]) < max(tol, 1e-4) and rel(a[2], b[2]) < max(tol, 1e-4)
        if skip:
            assert np.array_equal(a[3], b[3])


@pytest.mark.parametrize("case", [('convT_dgrad', 16, 4, 512, 1024, 1, 'relu', 512),       # up1's dgrad -> up0's backward (skip half: plain gradient), M = 64
                                  ('convT_dgrad', 16, 2, 512, 512, 1, 'lrelu', 512),       # up0's dgrad -> down7's backward: M = 16, 4 live taps
                                  ('conv_dgrad', 16, 1, 512, 512, 1, 'lrelu', 512),        # down7's dgrad: four parities of 16 rows, one live tap each
                                  ('conv_dgrad', 16, 2, 512, 512, 1, 'relu+mask', 512),    # four parities of 64 rows, dropout mask
                                  ('conv_dgrad', 2, 2, 256, 128, 2, 'lrelu', 128),         # two statistics groups
                                  ('convT_dgrad', 3, 4, 64, 64, 1, 'relu', 64)])           # ragged M = 12
def test_column_owner_kernel_equals_split_k_backward(ctx, case, planner_options):
    """conv_own_kernel carrying the whole normalisation backward of the layer below (dz, dgamma, dbeta, dy; the skip half of a decoder
    concat as the plain gradient) against the split-K dgrad + finishing slab reduce."""
    if ctx.dtype == 'f32':
        pytest.skip("column-owner kernel: 16-bit storage only")
    (a, b), skip = _normfuse_bwd(ctx, case, planner_options, [{'conv.own_max_rows': 64, 'conv.own_max_kb': 1 << 20}, {'conv.own_max_rows': 0}])
    tol = {'bf16': 2e-2, 'f16': 3e-3}[ctx.dtype]
    assert float(np.abs(b[0]).max()) > 1e-3
    assert rel(a[0], b[0]) < tol
    assert rel(a[1], b[1]) < tol and rel(a[2], b[2]) < tol
    if skip:
        assert rel(a[3], b[3]) < tol / 2


def test_wgrad_tr_read_matches_plain(ctx, monkeypatch):
    """bf16 wgrad fragments come from ds_read_b64_tr_b16; fp32 from ds_read_b32.  Exact integer data with an
    asymmetric pattern catches any row/column mix-up in the transposing read."""
    from gan_amd import _lib as L
    N, H, ci, co = 1, 16, 64, 128
    x = np.zeros((N, H, H, ci))
    dy = np.zeros((N, H // 2, H // 2, co))
    rng = np.random.default_rng(5)
    x[:] = rng.integers(-3, 4, x.shape)
    dy[:] = rng.integers(-3, 4, dy.shape)
    xb, xv = dev(ctx, x)
    dyb, dyv = dev(ctx, dy)
    dw = torch.zeros((16, ci, co), dtype=torch.float32, device=ctx.device)
    d = L.GanWgradDesc(ctx.dt, 2, xv, dyv, dw.data_ptr(), ci, co, 0, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
    torch.cuda.synchronize()
    _, ref = O.conv2d_bwd(x, np.zeros((4, 4, ci, co)), dy, 2, need_dx=False)
    assert np.array_equal(dw.cpu().numpy().reshape(4, 4, ci, co), ref)    # small integers: exact in bf16 and fp32


def test_wgrad_pingpong_128_columns_matches_128_tile_kernel(ctx, planner_options):
    """A 64-channel x 128-channel kernel gradient (G / D down1) on the 256-column ping-pong kernel (wgrad.pingpong_128, the
    default from 30 GFLOP up: half its columns are computed and dropped) against the 128x128-tile kernel, exact integer data."""
    from gan_amd import _lib as L
    if ctx.dtype == 'f32':
        pytest.skip("ping-pong wgrad: 16-bit storage only")
    N, H, ci, co = 2, 128, 64, 128
    rng = np.random.default_rng(11)
    x = rng.integers(-2, 3, (N, H, H, ci)).astype(np.float64)
    dy = rng.integers(-2, 3, (N, H // 2, H // 2, co)).astype(np.float64)
    xb, xv = dev(ctx, x)
    dyb, dyv = dev(ctx, dy)
    planner_options('wgrad.pingpong_min_gflop', 1)
    out = []
    for pp in (1, 0):
        planner_options('wgrad.pingpong_128', pp)
        dw = torch.zeros((16, ci, co), dtype=torch.float32, device=ctx.device)
        d = L.GanWgradDesc(ctx.dt, 2, xv, dyv, dw.data_ptr(), ci, co, 0, ctx.ws_ptr, ctx.ws_bytes)
        assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
        torch.cuda.synchronize()
        out.append(dw.cpu().numpy())
    assert np.abs(out[1]).max() > 100 and np.array_equal(out[0], out[1])


@pytest.mark.parametrize("case", [(4, 128, 64, 128, 2, 4),      # 64-channel BIG tensor: four taps per 256-row tile (table of 4 columns)
                                  (8, 64, 128, 256, 2, 2),      # 128 channels: two taps per tile
                                  (8, 32, 256, 512, 1, 1),      # stride 1, 31 x 31 grid: rows of 31, the last K tile half empty (M % 64 = 8)
                                  (2, 64, 128, 256, 2, 2)])     # few K tiles per block
def test_wgrad_row_table_equals_row_decode(ctx, case, planner_options):
    """wgrad_pp_kernel<T, NTAP>: the block's reduction rows decoded once into an LDS table (wgrad.row_table, default) against the decode
    per K tile in the loop - same pieces, same order: bit-identical kernel gradients (random data, split reductions included)."""
    from gan_amd import _lib as L
    if ctx.dtype == 'f32':
        pytest.skip("ping-pong wgrad: 16-bit storage only")
    N, H, ci, co, s, ntap = case
    rng = np.random.default_rng(5)
    Ho = (H + 2 - 4) // s + 1
    x = q(ctx, rng.standard_normal((N, H, H, ci)))
    dy = q(ctx, rng.standard_normal((N, Ho, Ho, co)))
    xb, xv = dev(ctx, x)
    dyb, dyv = dev(ctx, dy)
    planner_options('wgrad.pingpong_min_gflop', 1)
    out = []
    for tab in (1, 0):
        planner_options('wgrad.row_table', tab)
        dw = torch.zeros((16, ci, co), dtype=torch.float32, device=ctx.device)
        d = L.GanWgradDesc(ctx.dt, s, xv, dyv, dw.data_ptr(), ci, co, 0, ctx.ws_ptr, ctx.ws_bytes)
        winfo = (C.c_int32 * 4)()
        assert ctx.lib.gan_wgrad_plan_info(C.byref(d), winfo) == 0 and winfo[0] == 256 and winfo[1] == 256, list(winfo)   # the ping-pong kernel
        assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
        torch.cuda.synchronize()
        out.append(dw.cpu().numpy())
    assert np.abs(out[1]).max() > 1 and np.array_equal(out[0], out[1])
    _, dw_ref = O.conv2d_bwd(x, np.zeros((4, 4, ci, co)), dy, s, need_dx=False)
    assert rel(out[0].reshape(4, 4, ci, co), dw_ref) < TOL[ctx.dtype]


@pytest.mark.parametrize("case", [(2, 8, 512, 512, 'epilogue'), (4, 64, 64, 64, 'reduce'), (4, 128, 256, 256, 'reduce-pp'), (16, 32, 512, 128, 'reduce'),
                                  (16, 2, 512, 512, 'epilogue'),            # 2x2 -> 1x1 (G.down7 / up0 at 256x256): 12 of the 16 taps never meet the map
                                  (16, 2, 512, 512, 'epilogue-moments')])   # ... with non-zero moments there: those blocks must run the update
def test_wgrad_with_fused_adam_equals_wgrad_then_adam(ctx, case, planner_options):
    """GanAdamFuse: a wgrad launch that applies TF-form Adam to its kernel and refreshes both NK copies - in its own epilogue
    (un-split 128x128 launch) or at the end of its slab reduce (split launches, ping-pong kernel included; the reduce of the small
    tensor sums interleaved split groups) - leaves exactly (bit for bit) what gan_conv_wgrad followed by gan_adam_prepare_multi
    leaves, over two steps, and does not write dw."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf, ParamSet
    if ctx.dtype != 'bf16':
        pytest.skip("bf16 only: fp32 has no 16-bit epilogue, fp16 steps keep the whole-step inf/nan check (and the un-scaling) before any update")
    N, H, ci, co, how = case
    premoments = how.endswith('-moments')
    how = how.split('-moments')[0]
    planner_options('wgrad.reduce_adam_min_params', 0)        # (the default leaves kernels under 2^20 parameters to the separate passes)
    rng = np.random.default_rng(17)
    x, dy = q(ctx, rng.standard_normal((N, H, H, ci))), q(ctx, 0.1 * rng.standard_normal((N, H // 2, H // 2, co)))
    xb, xv = dev(ctx, x)
    dyb, dyv = dev(ctx, dy)
    w0 = (0.05 * rng.standard_normal((4, 4, ci, co))).astype(np.float32)
    info = (C.c_int32 * 4)()
    sets = []
    for fused in (False, True):
        P = ParamSet(ctx, [('k.kernel', (4, 4, ci, co), True), ('k.beta', (co,), True)])
        P.load_numpy({'k.kernel': w0, 'k.beta': np.zeros(co, np.float32)})
        if premoments:       # (a checkpoint trained at another resolution: the taps that are dead here carry moments)
            P.m.copy_(torch.from_numpy((1e-3 * np.random.default_rng(3).standard_normal(P.m.numel())).astype(np.float32)))
            P.v.copy_(torch.from_numpy((1e-6 * np.random.default_rng(4).random(P.v.numel())).astype(np.float32)))
        for step in range(2):
            if fused:
                ctx.run(P.adam_begin_ops(2e-4, 0.5, 0.999))
                af = P.adam_fuse_desc('k.kernel', 0.5, 0.999)
                d = L.GanWgradDesc(ctx.dt, 2, xv, dyv, P.ptr('k.kernel', 'grad'), ci, co, 0, ctx.ws_ptr, ctx.ws_bytes, 0, C.addressof(af))
                assert ctx.lib.gan_wgrad_adam_fused(C.byref(d)) == 1
                assert ctx.lib.gan_wgrad_plan_info(C.byref(d), info) == 0
                assert (info[2] == 1) == (how == 'epilogue') and (info[0] == 256) == (how == 'reduce-pp'), list(info)
                assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
                ctx.run(P.adam_rest_ops(['k.kernel'], 0.5, 0.999))
            else:
                d = L.GanWgradDesc(ctx.dt, 2, xv, dyv, P.ptr('k.kernel', 'grad'), ci, co, 0, ctx.ws_ptr, ctx.ws_bytes)
                assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
                P.adam(2e-4, 0.5, 0.999)
        torch.cuda.synchronize()
        sets.append([t.clone() for t in (P.master, P.m, P.v, P.nat['k.kernel'], P.tr['k.kernel'], P.step)])
        if fused:
            assert float(P.grad[:16 * ci * co].abs().max()) == 0.0          # the fused launch leaves dw alone
    for a, b in zip(*sets):
        assert torch.equal(a, b)
    assert float((sets[0][0][:16 * ci * co] - torch.from_numpy(w0).to(ctx.device).flatten()).abs().max()) > 1e-4      # it did move
    # a request the plan cannot honour is refused: an accumulating launch keeps the gradient in dw (the caller's optimiser pass follows)
    P2 = ParamSet(ctx, [('s.kernel', (4, 4, ci, co), True)])
    af = P2.adam_fuse_desc('s.kernel', 0.5, 0.999)
    d = L.GanWgradDesc(ctx.dt, 2, xv, dyv, P2.ptr('s.kernel', 'grad'), ci, co, 1, ctx.ws_ptr, ctx.ws_bytes, 0, C.addressof(af))
    assert ctx.lib.gan_wgrad_adam_fused(C.byref(d)) == 0
    assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == L.E_SHAPE


@pytest.mark.parametrize("case", [(2, 8, 512, 512, 'epilogue'), (4, 64, 64, 64, 'reduce'), (4, 128, 256, 256, 'reduce-pp'), (2, 16, 1, 64, 'fold')])
def test_wgrad_writes_the_wire_format_directly(ctx, case):
    """GanWgradDesc.dw_wire (data-parallel steps): the launch's last kernel - epilogue of an un-split launch, slab reduce of a split
    one, ping-pong kernel included - writes the gradient as bfloat16 into the exchange's wire buffer = gan_grad_pack of the fp32
    gradient, bit for bit, and leaves dw alone; a tap-folded layer declines (the caller keeps its cast pass for it)."""
    from gan_amd import _lib as L
    N, H, ci, co, how = case
    rng = np.random.default_rng(23)
    x, dy = q(ctx, rng.standard_normal((N, H, H, ci))), q(ctx, 0.1 * rng.standard_normal((N, H // 2, H // 2, co)))
    xb, xv = dev(ctx, x)
    if ci < 8:
        xv = xb.view()                                       # (the 8-channel zero-padded tensor the first layers are given)
    dyb, dyv = dev(ctx, dy)
    n = 16 * ci * co
    dw = torch.zeros(n, dtype=torch.float32, device=ctx.device)
    d = L.GanWgradDesc(ctx.dt, 2, xv, dyv, dw.data_ptr(), ci, co, 0, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
    wire = torch.zeros(n, dtype=torch.bfloat16, device=ctx.device)
    dw2 = torch.full((n,), 7.0, dtype=torch.float32, device=ctx.device)
    d2 = L.GanWgradDesc(ctx.dt, 2, xv, dyv, dw2.data_ptr(), ci, co, 0, ctx.ws_ptr, ctx.ws_bytes, 0, None, wire.data_ptr())
    honoured = ctx.lib.gan_wgrad_wire_direct(C.byref(d2))
    assert honoured == (0 if how == 'fold' else 1)
    if not honoured:
        assert ctx.lib.gan_conv_wgrad(C.byref(d2), ctx.stream()) == L.E_SHAPE
        return
    assert ctx.lib.gan_conv_wgrad(C.byref(d2), ctx.stream()) == 0
    torch.cuda.synchronize()
    assert float(dw.abs().max()) > 0
    assert torch.equal(wire, dw.to(torch.bfloat16))          # the cast pass's rounding (round to nearest even)
    assert float((dw2 - 7.0).abs().max()) == 0.0             # dw untouched


@pytest.mark.parametrize("kind,groups_of", [('batchnorm', lambda n: 1), ('batchnorm', lambda n: 2), ('instancenorm', lambda n: n)])
@pytest.mark.parametrize("act,drop", [('lrelu', False), ('relu', True)])
def test_norm_act_fwd_bwd(ctx, kind, groups_of, act, drop):
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    N, H, c = 4, 8, 64
    G = groups_of(N)
    rng = np.random.default_rng(3)
    y = q(ctx, rng.standard_normal((N, H, H, c)) * 1.5 + 0.3)
    gamma = (1 + 0.2 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    mask = (rng.random((N, H, H, c)) > 0.5).astype(np.float64) if drop else None
    eps = 1e-3 if kind == 'batchnorm' else 1e-5
    yb, yv = dev(ctx, y)
    ab = Buf(ctx, N, H, H, c + 64)
    f32 = torch.float32
    tg, tb = torch.from_numpy(gamma).to(ctx.device), torch.from_numpy(beta).to(ctx.device)
    mean, rstd = torch.zeros(G * c, dtype=f32, device=ctx.device), torch.zeros(G * c, dtype=f32, device=ctx.device)
    mm, mv = torch.zeros(c, dtype=f32, device=ctx.device), torch.ones(c, dtype=f32, device=ctx.device)
    tm = torch.from_numpy(mask.astype(np.uint8)).to(ctx.device) if drop else None
    bn = kind == 'batchnorm'
    d = L.GanNormDesc(ctx.dt, yv, ab.view(64, c), G, eps, tg.data_ptr(), tb.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                      mm.data_ptr() if bn else None, mv.data_ptr() if bn else None, 0.99, tm.data_ptr() if drop else None,
                      L.ACTS[act], 0.3, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_norm_stats(C.byref(d), ctx.stream()) == 0
    assert ctx.lib.gan_norm_act_fwd(C.byref(d), ctx.stream()) == 0
    torch.cuda.synchronize()
    # oracle: groups == separate calls on batch slices
    gs = N // G
    refs, caches = [], []
    st = {}
    for g in range(G):
        sl = slice(g * gs, (g + 1) * gs)
        P = {'l.gamma': gamma.astype(np.float64), 'l.beta': beta.astype(np.float64), 'l.scale': gamma.astype(np.float64),
             'l.offset': beta.astype(np.float64)}
        if kind == 'batchnorm':
            z, cache = O.norm_fwd(y[sl], P['l.gamma'], P['l.beta'], kind)
            n = gs * H * H
            mu, var = cache[2].reshape(-1), cache[3].reshape(-1)
            st.setdefault('mm', np.zeros(c)); st.setdefault('mv', np.ones(c))
            st['mm'] += (mu - st['mm']) * 0.01
            st['mv'] += (var * n / (n - 1) - st['mv']) * 0.01
        else:
            z, cache = O.norm_fwd(y[sl], P['l.scale'], P['l.offset'], kind)
        zd = z * mask[sl] * 2 if drop else z
        refs.append(O.act_fwd(zd, act))
        caches.append((cache, zd))
    ref = np.concatenate(refs)
    tol = TOL[ctx.dtype]
    assert rel(host(ab, 64, c), ref) < tol
    if bn:
        assert rel(mm.cpu().numpy(), st['mm']) < 1e-4 and rel(mv.cpu().numpy(), st['mv']) < 1e-4
    # backward, two upstream gradients
    da = q(ctx, rng.standard_normal((N, H, H, c)))
    da2 = q(ctx, rng.standard_normal((N, H, H, c)))
    dab, dav = dev(ctx, da, pitch=c + 8)
    da2b, da2v = dev(ctx, da2)
    dyb = Buf(ctx, N, H, H, c)
    dg, db = torch.zeros(c, dtype=f32, device=ctx.device), torch.zeros(c, dtype=f32, device=ctx.device)
    bd = L.GanNormBwdDesc(ctx.dt, yv, dav, da2v, dyb.view(), G, tg.data_ptr(), tb.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                          tm.data_ptr() if drop else None, L.ACTS[act], 0.3, dg.data_ptr(), db.data_ptr(), 0, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_norm_act_bwd(C.byref(bd), ctx.stream()) == 0
    torch.cuda.synchronize()
    dys, dgs, dbs = [], 0, 0
    for g in range(G):
        sl = slice(g * gs, (g + 1) * gs)
        cache, zd = caches[g]
        dzd = O.act_bwd(da[sl] + da2[sl], zd, refs[g], act)
        dz = dzd * mask[sl] * 2 if drop else dzd
        dy_, dg_, db_ = O.norm_bwd(dz, cache, gamma.astype(np.float64), kind)
        dys.append(dy_); dgs = dgs + dg_; dbs = dbs + db_
    assert rel(host(dyb), np.concatenate(dys)) < tol
    assert rel(dg.cpu().numpy(), dgs) < max(tol, 1e-4) and rel(db.cpu().numpy(), dbs) < max(tol, 1e-4)


@pytest.mark.parametrize("kind,groups_of", [('batchnorm', lambda n: 1), ('batchnorm', lambda n: 2), ('instancenorm', lambda n: n)])
@pytest.mark.parametrize("shape", [(4, 8, 64), (4, 30, 128), (2, 64, 256), (2, 2, 512)])
def test_norm_finalize_inside_apply_equals_separate_launches(ctx, kind, groups_of, shape):
    """GanNormDesc.sync / GanNormBwdDesc.sync: the apply launch finalizes the statistics (forward) / the sums, dgamma, dbeta (backward)
    itself - its first workgroups do it, the rest wait for them.  Bit-identical to the separate finalize launch; the sync area zeroes
    itself (three launches in a row on one area); (2, 2, 512): fewer apply workgroups than finalize work units -> two launches."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    N, H, c = shape
    G = groups_of(N)
    act, drop = ('relu', True) if c == 128 else ('lrelu', False)
    rng = np.random.default_rng(11)
    y = q(ctx, rng.standard_normal((N, H, H, c)) * 1.5 + 0.3)
    f32 = torch.float32
    tg = torch.from_numpy((1 + 0.2 * rng.standard_normal(c)).astype(np.float32)).to(ctx.device)
    tb = torch.from_numpy((0.2 * rng.standard_normal(c)).astype(np.float32)).to(ctx.device)
    tm = torch.from_numpy((rng.random((N, H, H, c)) > 0.5).astype(np.uint8)).to(ctx.device) if drop else None
    yb, yv = dev(ctx, y)
    da = q(ctx, rng.standard_normal((N, H, H, c)))
    dab, dav = dev(ctx, da, pitch=c + 8)
    bn = kind == 'batchnorm'
    eps = 1e-3 if bn else 1e-5
    words = ctx.lib.gan_norm_sync_bytes() // 4
    sync = torch.zeros(2, words, dtype=torch.int32, device=ctx.device)
    res = {}
    for how in ('separate', 'inside'):
        ab, dyb = Buf(ctx, N, H, H, c), Buf(ctx, N, H, H, c)
        mean, rstd = torch.zeros(G * c, dtype=f32, device=ctx.device), torch.zeros(G * c, dtype=f32, device=ctx.device)
        mm, mv = torch.zeros(c, dtype=f32, device=ctx.device), torch.ones(c, dtype=f32, device=ctx.device)
        dg, db = torch.zeros(c, dtype=f32, device=ctx.device), torch.zeros(c, dtype=f32, device=ctx.device)
        sp = (sync[0].data_ptr(), sync[1].data_ptr()) if how == 'inside' else (None, None)
        d = L.GanNormDesc(ctx.dt, yv, ab.view(), G, eps, tg.data_ptr(), tb.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                          mm.data_ptr() if bn else None, mv.data_ptr() if bn else None, 0.99, tm.data_ptr() if drop else None,
                          L.ACTS[act], 0.3, ctx.ws_ptr, ctx.ws_bytes, sp[0])
        bd = L.GanNormBwdDesc(ctx.dt, yv, dav, L.GanTensor(None, 0, 0, 0, 0, 0), dyb.view(), G, tg.data_ptr(), tb.data_ptr(), mean.data_ptr(),
                              rstd.data_ptr(), tm.data_ptr() if drop else None, L.ACTS[act], 0.3, dg.data_ptr(), db.data_ptr(), 0,
                              ctx.ws_ptr, ctx.ws_bytes, sp[1])
        for it in range(3):                      # (the moving averages move three times either way)
            if how == 'separate':
                assert ctx.lib.gan_norm_stats(C.byref(d), ctx.stream()) == 0
                assert ctx.lib.gan_norm_act_fwd(C.byref(d), ctx.stream()) == 0
            else:
                assert ctx.lib.gan_norm_stats_partial(C.byref(d), ctx.stream()) == 0
                assert ctx.lib.gan_norm_finalize_act_fwd(C.byref(d), 0, ctx.stream()) == 0
            assert ctx.lib.gan_norm_act_bwd(C.byref(bd), ctx.stream()) == 0
        torch.cuda.synchronize()
        res[how] = [t.clone() for t in (ab.t, dyb.t, mean, rstd, mm, mv, dg, db)]
    assert int(sync.abs().sum()) == 0            # every counter back at zero, no timeout flag
    for a_, b_, what in zip(res['separate'], res['inside'], ('a', 'dy', 'mean', 'rstd', 'moving_mean', 'moving_var', 'dgamma', 'dbeta')):
        assert torch.equal(a_, b_), what
    assert float(res['inside'][1].float().abs().max()) > 0


def test_act_bwd_bias_grad_losses(ctx):
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    rng = np.random.default_rng(9)
    N, H = 2, 16
    # tanh head backward with two upstream gradients
    a = q(ctx, np.tanh(rng.standard_normal((N, H, H, 1))))
    da = q(ctx, rng.standard_normal((N, H, H, 1)))
    da2 = q(ctx, rng.standard_normal((N, H, H, 1)))
    ab, _ = dev(ctx, a); dab, _ = dev(ctx, da); da2b, _ = dev(ctx, da2)
    dyb = Buf(ctx, N, H, H, 8)
    d = L.GanActBwdDesc(ctx.dt, ab.view(), dab.view(), da2b.view(), dyb.view(), L.ACT_TANH, 0.3, None, 0, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_act_bwd(C.byref(d), ctx.stream()) == 0
    dbias = torch.zeros(8, dtype=torch.float32, device=ctx.device)
    v = dyb.view()
    assert ctx.lib.gan_bias_grad(ctx.dt, C.byref(v), dbias.data_ptr(), 0, ctx.ws_ptr, ctx.ws_bytes, ctx.stream()) == 0
    torch.cuda.synchronize()
    ref = (da + da2) * (1 - a * a)
    tol = TOL[ctx.dtype]
    assert rel(host(dyb, 0, 1), ref) < tol
    assert abs(dbias[0].item() - q(ctx, ref).sum()) < max(tol, 1e-4) * np.abs(ref).sum()
    assert np.all(dbias[1:].cpu().numpy() == 0)
    # BCE from logits (+ gradient into an 8-pitch typed buffer)
    x = (3 * rng.standard_normal((N, 30, 30, 1))).astype(np.float32)
    xt = torch.from_numpy(x).to(ctx.device)
    loss = torch.zeros(2, dtype=torch.float32, device=ctx.device)
    dxb = Buf(ctx, N, 30, 30, 8)
    for tgt in (1.0, 0.0):
        rc = ctx.lib.gan_bce_logits(xt.data_ptr(), x.size, tgt, 0.5, 0, loss.data_ptr(), 0.5, ctx.dt, dxb.t.data_ptr(), 8, ctx.ws_ptr, None, ctx.stream())
        assert rc == 0
        torch.cuda.synchronize()
        l, g = O.bce_logits(x.astype(np.float64), tgt)
        assert abs(loss[0].item() - 0.5 * l) < 1e-6 * max(1, abs(l))
        assert rel(host(dxb, 0, 1), 0.5 * g) < (1e-5 if ctx.dtype == 'f32' else 1e-2)
    # KAT: BCE(logit 0) = ln 2
    z = torch.zeros(900, dtype=torch.float32, device=ctx.device)
    ctx.lib.gan_bce_logits(z.data_ptr(), 900, 1.0, 1.0, 0, loss.data_ptr(), 1.0, ctx.dt, None, 8, ctx.ws_ptr, None, ctx.stream())
    torch.cuda.synchronize()
    assert abs(loss[0].item() - np.log(2)) < 1e-6
    # L1 mean + sign gradient, accumulate flag
    a1 = q(ctx, rng.uniform(-1, 1, (N, H, H, 3))); b1 = q(ctx, rng.uniform(-1, 1, (N, H, H, 3)))
    a1b, a1v = dev(ctx, a1); b1b, b1v = dev(ctx, b1, pitch=16, c0=3)
    gb = Buf(ctx, N, H, H, 8)
    gv = gb.view(0, 3)
    ws = torch.zeros(4096, dtype=torch.float32, device=ctx.device)
    loss.zero_(); loss[0] = 1.0
    rc = ctx.lib.gan_l1(ctx.dt, C.byref(a1v), C.byref(b1v), 2.0, 1, loss.data_ptr(), 100.0, C.byref(gv), ws.data_ptr(), None, ctx.stream())
    assert rc == 0
    torch.cuda.synchronize()
    l, g = O.l1_mean(a1, b1)
    assert abs(loss[0].item() - (1.0 + 2.0 * l)) < 1e-5
    assert rel(host(gb, 0, 3), 100.0 * g) < (1e-6 if ctx.dtype == 'f32' else 1e-2)


def test_adam_tf_and_weight_prep(ctx):
    rng = np.random.default_rng(2)
    n = 4096
    p0 = rng.standard_normal(n).astype(np.float32)
    P = {'w': p0.astype(np.float64).copy()}
    opt = O.AdamTF(2e-4, 0.5, 0.999)
    f32 = torch.float32
    p = torch.from_numpy(p0.copy()).to(ctx.device)
    m, v = torch.zeros(n, dtype=f32, device=ctx.device), torch.zeros(n, dtype=f32, device=ctx.device)
    step = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    lr_t = torch.zeros(1, dtype=f32, device=ctx.device)
    for it in range(3):
        g = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 1, n)).astype(np.float32)
        gt = torch.from_numpy(g).to(ctx.device)
        assert ctx.lib.gan_adam_begin(step.data_ptr(), lr_t.data_ptr(), 2e-4, 0.5, 0.999, None, ctx.stream()) == 0
        assert ctx.lib.gan_adam_tf(p.data_ptr(), m.data_ptr(), v.data_ptr(), gt.data_ptr(), n, lr_t.data_ptr(), 0.5, 0.999,
                                   1e-7, 1.0, None, 0, ctx.stream()) == 0
        opt.apply(P, {'w': g.astype(np.float64)})
    torch.cuda.synchronize()
    assert step.item() == 3
    assert np.abs(p.cpu().numpy() - P['w']).max() < 1e-6
    # weight prep layouts
    w = rng.standard_normal((4, 4, 3, 64)).astype(np.float32)
    nat, tr = prep(ctx, w)
    wq = q(ctx, w).reshape(16, 3, 64)
    assert np.array_equal(nat.float().cpu().numpy(), wq)
    t = tr.float().cpu().numpy()
    assert t.shape == (16, 64, 8) and np.array_equal(t[:, :, :3], wq.transpose(0, 2, 1)) and np.all(t[:, :, 3:] == 0)


def test_dropout_mask_and_pack(ctx):
    from gan_amd.nets import Buf
    step = torch.tensor([5], dtype=torch.int32, device=ctx.device)
    n = 16 * 8 * 8 * 512
    m1 = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
    m2 = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
    assert ctx.lib.gan_dropout_mask(m1.data_ptr(), n, 123, step.data_ptr(), 0, ctx.stream()) == 0
    assert ctx.lib.gan_dropout_mask(m2.data_ptr(), n, 123, step.data_ptr(), 1, ctx.stream()) == 0
    torch.cuda.synchronize()
    a, b = m1.cpu().numpy(), m2.cpu().numpy()
    assert set(np.unique(a)) == {0, 1} and abs(a.mean() - 0.5) < 5e-3 and abs((a == b).mean() - 0.5) < 5e-3
    x = np.random.default_rng(1).uniform(-1, 1, (2, 8, 8, 3)).astype(np.float32)
    xt = torch.from_numpy(x).to(ctx.device)
    b8 = Buf(ctx, 2, 8, 8, 8)
    v = b8.view(2, 3)
    assert ctx.lib.gan_pack(ctx.dt, xt.data_ptr(), C.byref(v), ctx.stream()) == 0
    out = torch.zeros_like(xt)
    assert ctx.lib.gan_unpack(ctx.dt, C.byref(v), out.data_ptr(), ctx.stream()) == 0
    b2 = Buf(ctx, 2, 8, 8, 16)
    v2 = b2.view(5, 3)
    assert ctx.lib.gan_copy_view(ctx.dt, C.byref(v), C.byref(v2), ctx.stream()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().astype(np.float64), q(ctx, x))
    assert np.array_equal(host(b2, 5, 3), q(ctx, x)) and np.all(host(b2, 0, 5) == 0)


def test_bad_arguments_return_codes(ctx):
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    xb = Buf(ctx, 1, 8, 8, 8)
    yb = Buf(ctx, 1, 5, 5, 64)        # wrong output size for stride 2
    w = torch.zeros((16, 64, 8), dtype=ctx.tdtype, device=ctx.device)
    d = L.GanConvDesc(ctx.dt, 2, xb.view(), yb.view(), w.data_ptr(), 64, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_conv2d_fwd(C.byref(d), ctx.stream()) == -2
    d2 = L.GanConvDesc(ctx.dt, 2, xb.view(), yb.view(), None, 64, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes)
    assert ctx.lib.gan_conv2d_fwd(C.byref(d2), ctx.stream()) == -1
    # empty batch, channel count that is not a multiple of 8, misaligned input pointer
    y4 = Buf(ctx, 1, 4, 4, 64)
    e = xb.view(); e.n = 0
    ye = y4.view(); ye.n = 0
    assert ctx.lib.gan_conv2d_fwd(C.byref(L.GanConvDesc(ctx.dt, 2, e, ye, w.data_ptr(), 64, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes)),
                                  ctx.stream()) == -2
    odd = xb.view(); odd.c = 6
    assert ctx.lib.gan_conv2d_fwd(C.byref(L.GanConvDesc(ctx.dt, 2, odd, y4.view(), w.data_ptr(), 64, None, 0, 0.3, 0, ctx.ws_ptr,
                                                        ctx.ws_bytes)), ctx.stream()) == -2
    mis = xb.view(); mis.ptr += 2
    assert ctx.lib.gan_conv2d_fwd(C.byref(L.GanConvDesc(ctx.dt, 2, mis, y4.view(), w.data_ptr(), 64, None, 0, 0.3, 0, ctx.ws_ptr,
                                                        ctx.ws_bytes)), ctx.stream()) == -1
    # a split-K layer and a thin-N layer without (enough) workspace
    xs, ys = Buf(ctx, 4, 4, 4, 512), Buf(ctx, 4, 2, 2, 512)
    ws_ = torch.zeros((16, 512, 512), dtype=ctx.tdtype, device=ctx.device)
    ds = L.GanConvDesc(ctx.dt, 2, xs.view(), ys.view(), ws_.data_ptr(), 512, None, 0, 0.3, 0, ctx.ws_ptr, 1024)
    assert ctx.lib.gan_conv_workspace_bytes(C.byref(ds), 0) > 1024
    assert ctx.lib.gan_conv2d_fwd(C.byref(ds), ctx.stream()) == -3
    if ctx.dtype == 'bf16':
        xt, yt = Buf(ctx, 1, 8, 8, 128), Buf(ctx, 1, 16, 16, 8)
        dt_ = L.GanConvDesc(ctx.dt, 2, xt.view(), yt.view(0, 1), ws_.data_ptr(), 1, None, 0, 0.3, 0, None, 0)
        L.set_option('conv.thin_fused', 0)          # the two-launch thin-N path keeps its per-tap products in the workspace
        try:
            assert ctx.lib.gan_convT2d_fwd(C.byref(dt_), ctx.stream()) == -3
        finally:
            L.set_option('conv.thin_fused', 1)
        assert ctx.lib.gan_convT2d_fwd(C.byref(dt_), ctx.stream()) == 0         # fused (default): they stay in LDS, no workspace
    torch.cuda.synchronize()


def test_fused_adam_prepare_equals_adam_then_prepare(ctx):
    """gan_adam_prepare_multi (Adam + NK copies in one pass over the kernel tensors) must leave exactly what
    gan_adam_tf followed by gan_weights_prepare leaves: master, both moments, native and transposed copies -
    including thin tensors (3 or 1 channels on a side) that take its scalar path."""
    from gan_amd import _lib as L
    rng = np.random.default_rng(12)
    shapes = [(64, 128), (3, 64), (128, 1), (100, 72)]                  # (A, B) per 4x4 kernel
    offs, total = [], 0
    for A, B in shapes:
        offs.append(total)
        total += (16 * A * B + 63) // 64 * 64
    f32 = torch.float32
    mk = lambda scale: torch.from_numpy((scale * rng.standard_normal(total)).astype(np.float32)).to(ctx.device)
    master, m, v, g = mk(0.05), mk(0.01), mk(0.01).abs(), mk(1.0)
    ref = [t.clone() for t in (master, m, v)]
    step = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    lr_t = torch.zeros(1, dtype=f32, device=ctx.device)
    assert ctx.lib.gan_adam_begin(step.data_ptr(), lr_t.data_ptr(), 2e-4, 0.5, 0.999, None, ctx.stream()) == 0
    pad8 = lambda c: (c + 7) // 8 * 8
    nats = [torch.zeros((16, A, pad8(B)), dtype=ctx.tdtype, device=ctx.device) for A, B in shapes]
    trs = [torch.zeros((16, B, pad8(A)), dtype=ctx.tdtype, device=ctx.device) for A, B in shapes]
    ents, tiles = [], 0
    for (A, B), o, nat, tr in zip(shapes, offs, nats, trs):
        tb = (pad8(B) + 63) // 64
        ents.append(L.GanPrepEntry(master.data_ptr() + 4 * o, nat.data_ptr(), tr.data_ptr(), A, B, tiles, tb))
        tiles += 16 * ((pad8(A) + 63) // 64) * tb
    arr = (L.GanPrepEntry * len(ents))(*ents)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(ctx.device)
    assert ctx.lib.gan_adam_prepare_multi(table.data_ptr(), len(ents), tiles, ctx.dt, master.data_ptr(), m.data_ptr(), v.data_ptr(),
                                          g.data_ptr(), lr_t.data_ptr(), 0.5, 0.999, 1e-7, 0.5, None, 0, ctx.stream()) == 0
    # reference: plain Adam over the flat buffer, then the stand-alone prep of every tensor
    assert ctx.lib.gan_adam_tf(ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr(), g.data_ptr(), total, lr_t.data_ptr(),
                               0.5, 0.999, 1e-7, 0.5, None, 0, ctx.stream()) == 0
    torch.cuda.synchronize()
    # the same two entry points reading the gradient from a bf16 wire buffer (data-parallel exchange) = fp32 Adam on the
    # bf16-rounded gradient, bit for bit
    gw = g.to(torch.bfloat16)
    gr = gw.float()
    a1 = [t.clone() for t in ref]
    a2 = [t.clone() for t in ref]
    a3 = [t.clone() for t in ref]
    assert ctx.lib.gan_adam_tf(a1[0].data_ptr(), a1[1].data_ptr(), a1[2].data_ptr(), gw.data_ptr(), total, lr_t.data_ptr(),
                               0.5, 0.999, 1e-7, 0.5, None, 1, ctx.stream()) == 0
    assert ctx.lib.gan_adam_tf(a2[0].data_ptr(), a2[1].data_ptr(), a2[2].data_ptr(), gr.data_ptr(), total, lr_t.data_ptr(),
                               0.5, 0.999, 1e-7, 0.5, None, 0, ctx.stream()) == 0
    ents3 = [L.GanPrepEntry(a3[0].data_ptr() + 4 * o, None, None, A, B, e.tile_start, e.tiles_b)      # (no NK copies here)
             for (A, B), o, e in zip(shapes, offs, ents)]
    table3 = torch.frombuffer(bytearray(bytes((L.GanPrepEntry * len(ents3))(*ents3))), dtype=torch.uint8).to(ctx.device)
    assert ctx.lib.gan_adam_prepare_multi(table3.data_ptr(), len(ents3), tiles, ctx.dt, a3[0].data_ptr(), a3[1].data_ptr(), a3[2].data_ptr(),
                                          gw.data_ptr(), lr_t.data_ptr(), 0.5, 0.999, 1e-7, 0.5, None, 1, ctx.stream()) == 0
    torch.cuda.synchronize()
    for x, y in zip(a1, a2):
        assert torch.equal(x, y)
    for (A, B), o in zip(shapes, offs):
        sl = slice(o, o + 16 * A * B)
        for x, y in zip(a3, a2):
            assert torch.equal(x[sl], y[sl]), (A, B)
    for (A, B), o, nat, tr in zip(shapes, offs, nats, trs):
        sl = slice(o, o + 16 * A * B)
        for got, want in zip((master, m, v), ref):
            assert torch.equal(got[sl], want[sl]), (A, B)
        nat2, tr2 = torch.zeros_like(nat), torch.zeros_like(tr)
        assert ctx.lib.gan_weights_prepare(ref[0].data_ptr() + 4 * o, A, B, ctx.dt, nat2.data_ptr(), tr2.data_ptr(), ctx.stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(nat, nat2) and torch.equal(tr, tr2), (A, B)


def test_patchgan_losses_equal_three_bce_calls(ctx):
    """gan_patchgan_losses (one pass over D(real), D(fake)) against three gan_bce_logits calls and the oracle."""
    from gan_amd.nets import Buf
    rng = np.random.default_rng(21)
    N = 3
    real = (2.5 * rng.standard_normal((N, 30, 30, 1))).astype(np.float32)
    fake = (2.5 * rng.standard_normal((N, 30, 30, 1)) - 0.5).astype(np.float32)
    tr_, tf_ = torch.from_numpy(real).to(ctx.device), torch.from_numpy(fake).to(ctx.device)
    cnt = real.size
    losses = torch.zeros(4, dtype=torch.float32, device=ctx.device)       # [gen_total, gan, l1, disc]
    losses[2] = 0.37
    gb, rb, fb = (Buf(ctx, N, 30, 30, 8) for _ in range(3))
    lp = losses.data_ptr()
    assert ctx.lib.gan_patchgan_losses(tr_.data_ptr(), tf_.data_ptr(), cnt, ctx.dt, gb.t.data_ptr(), rb.t.data_ptr(), fb.t.data_ptr(), 8,
                                       100.0, lp + 8, lp, lp + 4, lp + 12, ctx.ws_ptr, None, ctx.stream()) == 0
    ref = torch.zeros(2, dtype=torch.float32, device=ctx.device)
    g2, r2, f2 = (Buf(ctx, N, 30, 30, 8) for _ in range(3))
    ws2 = ctx.ws_ptr + 65536
    bce = ctx.lib.gan_bce_logits
    assert bce(tf_.data_ptr(), cnt, 1.0, 1.0, 0, ref.data_ptr(), 1.0, ctx.dt, g2.t.data_ptr(), 8, ws2, None, ctx.stream()) == 0
    assert bce(tr_.data_ptr(), cnt, 1.0, 0.5, 0, ref.data_ptr() + 4, 0.5, ctx.dt, r2.t.data_ptr(), 8, ws2, None, ctx.stream()) == 0
    assert bce(tf_.data_ptr(), cnt, 0.0, 0.5, 1, ref.data_ptr() + 4, 0.5, ctx.dt, f2.t.data_ptr(), 8, ws2, None, ctx.stream()) == 0
    torch.cuda.synchronize()
    got = losses.cpu().numpy()
    assert np.allclose(got[[1, 3]], ref.cpu().numpy(), rtol=1e-6)
    assert abs(got[0] - (got[1] + 100.0 * 0.37)) < 1e-5
    for a, b in ((gb, g2), (rb, r2), (fb, f2)):
        assert torch.equal(a.t, b.t)
    lg, _ = O.bce_logits(fake.astype(np.float64), 1.0)
    lr_, _ = O.bce_logits(real.astype(np.float64), 1.0)
    lf, _ = O.bce_logits(fake.astype(np.float64), 0.0)
    assert abs(got[1] - lg) < 1e-6 * max(1, lg) and abs(got[3] - 0.5 * (lr_ + lf)) < 1e-6


def test_multi_launch_pack_and_dropout_equal_single_calls(ctx):
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    rng = np.random.default_rng(4)
    a = torch.from_numpy(rng.standard_normal((2, 16, 16, 1)).astype(np.float32)).to(ctx.device)
    b = torch.from_numpy(rng.standard_normal((2, 16, 16, 1)).astype(np.float32)).to(ctx.device)
    big, one = Buf(ctx, 4, 16, 16, 8), Buf(ctx, 2, 16, 16, 8)
    big2, one2 = Buf(ctx, 4, 16, 16, 8), Buf(ctx, 2, 16, 16, 8)
    pairs = [(a, one.view(0, 1)), (a, big.view(0, 1, 0, 2)), (a, big.view(0, 1, 2, 2)), (b, big.view(1, 1, 0, 2))]
    srcs = (C.c_void_p * 4)(*[s.data_ptr() for s, _ in pairs])
    dsts = (L.GanTensor * 4)(*[d for _, d in pairs])
    assert ctx.lib.gan_pack_multi(ctx.dt, 4, srcs, dsts, ctx.stream()) == 0
    for s, d in [(a, one2.view(0, 1)), (a, big2.view(0, 1, 0, 2)), (a, big2.view(0, 1, 2, 2)), (b, big2.view(1, 1, 0, 2))]:
        assert ctx.lib.gan_pack(ctx.dt, s.data_ptr(), C.byref(d), ctx.stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(big.t, big2.t) and torch.equal(one.t, one2.t)
    # dropout masks: same counter hash, three masks of different sizes in one launch
    step = torch.full((1,), 5, dtype=torch.int32, device=ctx.device)
    sizes = [1000, 4096, 77]
    m1 = [torch.zeros(n, dtype=torch.uint8, device=ctx.device) for n in sizes]
    m2 = [torch.zeros(n, dtype=torch.uint8, device=ctx.device) for n in sizes]
    ptrs = (C.c_void_p * 3)(*[m.data_ptr() for m in m1])
    cnts = (C.c_int64 * 3)(*sizes)
    sids = (C.c_uint32 * 3)(8, 9, 10)
    assert ctx.lib.gan_dropout_mask_multi(3, ptrs, cnts, 1234, step.data_ptr(), sids, None, ctx.stream()) == 0
    for m, sid in zip(m2, (8, 9, 10)):
        assert ctx.lib.gan_dropout_mask(m.data_ptr(), m.numel(), 1234, step.data_ptr(), sid, ctx.stream()) == 0
    torch.cuda.synchronize()
    for x, y in zip(m1, m2):
        assert torch.equal(x, y) and 0.3 < x.float().mean().item() < 0.7
    # with a launch counter: the first launch equals the plain hash, each later launch draws new masks (the step stands still,
    # as in a validation loop), and the counter is advanced exactly once per launch by the kernel itself
    draws = torch.zeros(2, dtype=torch.int32, device=ctx.device)
    seen = []
    for it in range(3):
        assert ctx.lib.gan_dropout_mask_multi(3, ptrs, cnts, 1234, step.data_ptr(), sids, draws.data_ptr(), ctx.stream()) == 0
        torch.cuda.synchronize()
        assert draws.tolist() == [it + 1, 0]
        seen.append([m.clone() for m in m1])
    assert all(torch.equal(a_, b_) for a_, b_ in zip(seen[0], m2))
    assert not torch.equal(seen[0][1], seen[1][1]) and not torch.equal(seen[1][1], seen[2][1])
    assert 0.4 < (seen[1][1] == seen[2][1]).float().mean().item() < 0.6          # independent draws


# ---- operands embedded in NaN-filled memory: nothing behind a tensor may reach a result -------------------------------------
def _guarded(t, guard_elems=2 << 20):
    """Copy of device tensor `t` between two guard regions full of NaN -> (holder, device pointer of the copy)."""
    flat = torch.full((2 * guard_elems + t.numel(),), float('nan'), dtype=t.dtype, device=t.device)
    flat[guard_elems:guard_elems + t.numel()] = t.reshape(-1)
    return flat, flat.data_ptr() + guard_elems * t.element_size()


def _guarded_view(buf):
    from gan_amd import _lib as L
    holder, ptr = _guarded(buf.t)
    return holder, L.GanTensor(ptr, buf.n, buf.h, buf.w, buf.c, buf.c)


@pytest.mark.parametrize("case", [(8, 32, 256, 512, 1),       # D conv4's shape: 31 x 31 grid, M % 64 = 8 (the last K tile mostly past the end)
                                  (5, 100, 64, 128, 2),       # rows of 50, M = 12500: M % 64 = 20, four taps per tile
                                  (3, 64, 128, 256, 2)])      # M % 64 = 0 (control)
def test_wgrad_pingpong_never_reads_past_its_operands(ctx, case, planner_options):
    """wgrad_pp_kernel's row-table path addresses reduction rows m >= M of the SMALL operand (and K tiles past a split's range) with
    an in-range vector offset + a SCALAR offset that leaves the tensor, relying on the buffer descriptor's range check covering
    voffset + soffset (raw buffers: out of range iff offset >= num_records - soffset).  Both operands sit between NaN guards here:
    were the scalar offset not range-checked, live NaNs behind dy would meet the zeroed BIG rows in the MFMA (0 x NaN = NaN) and
    poison dW.  Result must be finite and bit-equal to the run on plain buffers, for the table path and the in-loop decode."""
    from gan_amd import _lib as L
    if ctx.dtype == 'f32':
        pytest.skip("ping-pong wgrad: 16-bit storage only")
    N, H, ci, co, s = case
    rng = np.random.default_rng(11)
    Ho = (H + 2 - 4) // s + 1
    xb, xv = dev(ctx, q(ctx, rng.standard_normal((N, H, H, ci))))
    dyb, dyv = dev(ctx, q(ctx, rng.standard_normal((N, Ho, Ho, co))))
    hx, gxv = _guarded_view(xb)
    hd, gdv = _guarded_view(dyb)
    planner_options('wgrad.pingpong_min_gflop', 1)
    out = {}
    for tab in (1, 0):
        planner_options('wgrad.row_table', tab)
        for guard, (a, b) in (('plain', (xv, dyv)), ('guarded', (gxv, gdv))):
            dw = torch.zeros((16, ci, co), dtype=torch.float32, device=ctx.device)
            d = L.GanWgradDesc(ctx.dt, s, a, b, dw.data_ptr(), ci, co, 0, ctx.ws_ptr, ctx.ws_bytes)
            winfo = (C.c_int32 * 4)()
            assert ctx.lib.gan_wgrad_plan_info(C.byref(d), winfo) == 0 and (winfo[0], winfo[1]) == (256, 256) and winfo[2] > 1, list(winfo)
            assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
            torch.cuda.synchronize()
            out[tab, guard] = dw.cpu().numpy()
    assert np.isfinite(out[1, 'guarded']).all() and np.isfinite(out[0, 'guarded']).all()
    assert np.abs(out[1, 'plain']).max() > 1
    for k in ((1, 'guarded'), (0, 'plain'), (0, 'guarded')):
        assert np.array_equal(out[k], out[1, 'plain']), k
    assert torch.isnan(hx[:16]).all() and torch.isnan(hd[-16:]).all()          # (the guards are still NaN: nothing wrote there either)


@pytest.mark.parametrize("case", [('conv_fwd', 16, 100, 64, 128, 2),       # table-driven 256x128 tiles, rows of 50, ragged last tile
                                  ('conv_fwd', 32, 32, 256, 512, 1),       # D conv4: tap-shared 256x256 tiles, rows of 31, ragged
                                  ('conv_dgrad', 32, 31, 512, 256, 1),     # its dgrad (shifts to the left), ragged
                                  ('convT_fwd', 16, 16, 1024, 256, 2),     # split K: chunks past a split's range
                                  ('convT_fwd', 1, 64, 128, 64, 2),        # parity-patch kernel: halo rows above / below the image
                                  ('conv_dgrad', 3, 32, 64, 64, 2)])       # parity-patch kernel as a stride-2 dgrad, three images
def test_conv_tiles_never_read_past_their_operands(ctx, case, planner_options):
    """The 256-row convolution kernels (tap-shared, table-driven, parity-patch) name every staged piece as register + scalar chunk
    offset and rely on the descriptor's range check (voffset + soffset) for rows past M, taps outside the map and chunks past the
    block's range.  Activation and weight operands between NaN guards: the output must be finite and bit-equal to the run on plain
    buffers."""
    from gan_amd import _lib as L
    from gan_amd.nets import Buf
    if ctx.dtype == 'f32':
        pytest.skip("16-bit kernels")
    op, N, H, cx, cy, s = case
    if cy == 64:
        planner_options('conv.parity_patch_min_blocks', 1)
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    Ho = {'conv_fwd': (H + 2 - 4) // s + 1, 'conv_dgrad': H * 2 if s == 2 else H + 1, 'convT_fwd': 2 * H, 'convT_dgrad': H // 2}[op]
    x = q(ctx, rng.standard_normal((N, H, H, cx)))
    kin, kout = {'conv_fwd': (cx, cy), 'conv_dgrad': (cy, cx), 'convT_fwd': (cy, cx), 'convT_dgrad': (cx, cy)}[op]
    nat, tr = prep(ctx, q(ctx, 0.05 * rng.standard_normal((4, 4, kin, kout))))
    wk = {'conv_fwd': tr, 'conv_dgrad': nat, 'convT_fwd': nat, 'convT_dgrad': tr}[op]
    xb, xv = dev(ctx, x)
    hx, gxv = _guarded_view(xb)
    hw, gw = _guarded(wk)
    fn = getattr(ctx.lib, {'conv_fwd': 'gan_conv2d_fwd', 'conv_dgrad': 'gan_conv2d_dgrad', 'convT_fwd': 'gan_convT2d_fwd', 'convT_dgrad': 'gan_convT2d_dgrad'}[op])
    outs = []
    for xa, wa in ((xv, wk.data_ptr()), (gxv, gw)):
        yb = Buf(ctx, N, Ho, Ho, cy)
        d = L.GanConvDesc(ctx.dt, s, xa, yb.view(0, cy), wa, cy, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes)
        info = (C.c_int32 * 5)()
        assert ctx.lib.gan_conv_plan_info(C.byref(d), ['conv_fwd', 'conv_dgrad', 'convT_fwd', 'convT_dgrad'].index(op), info) == 0
        assert info[0] in (256, 1024), list(info)                 # a 256-row tile or the parity-patch kernel
        assert fn(C.byref(d), ctx.stream()) == 0
        torch.cuda.synchronize()
        outs.append(host(yb))
    assert np.isfinite(outs[1]).all()
    assert np.abs(outs[0]).max() > 0.1 and np.array_equal(outs[0], outs[1])
