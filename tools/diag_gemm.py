#!/usr/bin/env python3
"""In-kernel time breakdown of ONE conv GEMM launch (diagnostic library with stamps, tools/diag_build.sh):
   GAN_AMD_LIB=gan_amd/libgan_amd_diag.so python tools/diag_gemm.py conv_fwd N H Cin Cout stride
Per block: entry -> setup done -> first tiles landed -> K loop done -> epilogue done (100 MHz ticks)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx, Buf

op, N, H, ci, co, s = sys.argv[1], *map(int, sys.argv[2:7])
ctx = Ctx('cuda:0', os.environ.get('DT', 'bf16'))
lib = ctx.lib
for kv in os.environ.get('OPTS', '').split(','):       # planner options: OPTS=conv.tap_share=3,...
    if kv:
        L.set_option(kv.split('=')[0], int(kv.split('=')[1]))
lib.gan_diag_set.argtypes = [C.c_void_p]
opi = {'conv_fwd': 0, 'conv_dgrad': 1, 'convT_fwd': 2, 'convT_dgrad': 3}[op]
if op == 'conv_fwd':
    Ho = (H + 2 - 4) // s + 1
    x, y = Buf(ctx, N, H, H, ci), Buf(ctx, N, Ho, Ho, co)
elif op == 'convT_fwd' or op == 'conv_dgrad':
    x, y = Buf(ctx, N, H, H, ci), Buf(ctx, N, 2 * H, 2 * H, co)
else:
    x, y = Buf(ctx, N, H, H, ci), Buf(ctx, N, H // 2, H // 2, co)
w = (torch.randn(16, co, ci, device='cuda') * 0.05).to(ctx.tdtype)
x.t.copy_(torch.randn_like(x.t.float()).to(ctx.tdtype))
stats = int(os.environ.get('STATS', '0'))
d = L.GanConvDesc(ctx.dt, s, x.view(), y.view(), w.data_ptr(), co, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes,
                  ctx.ws_lanes[1].data_ptr() if stats else None, stats, ctx.ws_lanes[1].numel())
fn = [lib.gan_conv2d_fwd, lib.gan_conv2d_dgrad, lib.gan_convT2d_fwd, lib.gan_convT2d_dgrad][opi]
info = (C.c_int32 * 5)(); lib.gan_conv_plan_info(C.byref(d), opi, info)
diag = torch.zeros(1 << 16, 16, dtype=torch.int64, device='cuda')
for _ in range(3):
    assert fn(C.byref(d), ctx.stream()) == 0
torch.cuda.synchronize()
flush = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
small = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
for cold in (0, 1, 2):       # 2: "MALL-warm" - operands read once after the flush, then the L2s (4 MB x 8) flushed by a 64-MB fill that leaves
    if cold:                 #    the 256-MB memory-side cache holding them
        flush.zero_()
    if cold == 2:
        (w.float().sum() + x.t.float().sum()).item()
        small.zero_()
    lib.gan_diag_set(diag.data_ptr()); diag.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(C.byref(d), ctx.stream()); e1.record()
    torch.cuda.synchronize()
    lib.gan_diag_set(None)
    t = diag.cpu().numpy()
    t = t[t[:, 0] != 0]
    segc = t[:, 8:14].astype(np.float64).mean(axis=0)
    nph = int(os.environ.get('NPH', '0'))
    if t[:, 5].any():
        ex = t[:, [0, 5, 6, 7, 1]].astype(np.float64) * 0.01
        print('   setup split (us): block decode %.2f  piece tables %.2f  B rows + descriptors %.2f  fragment tables %.2f' % tuple(np.diff(ex, axis=1).mean(axis=0)))
    t = t[:, :5].astype(np.float64) * 0.01    # us
    t0 = t[:, 0].min()
    seg = np.diff(t, axis=1)
    print(f"{op} N{N} H{H} {ci}->{co} s{s} tile {info[0]}x{info[1]} splits {info[2]} blocks {len(t)} {['warm', 'cold', 'mall'][cold]}: event {e0.elapsed_time(e1)*1e3:.1f} us, "
          f"kernel span {t[:, 4].max() - t0:.1f} us")
    print(f"   block start spread {t[:, 0].max() - t0:.2f} us; mean per block: setup {seg[:, 0].mean():.2f}  first-fill {seg[:, 1].mean():.2f}  "
          f"loop {seg[:, 2].mean():.2f}  epilogue {seg[:, 3].mean():.2f} (max {seg[:, 3].max():.2f})  total {(t[:, 4] - t[:, 0]).mean():.2f} us")
    tot = segc.sum()
    if tot > 0:
        print("   loop cycles of wave 0 (shader clock): " + "  ".join(f"{n} {v / tot * 100:.0f}%" for n, v in zip(
            ["frag reads", "vmcnt wait", "barrier(partner math)", "lgkm+MFMA", "barrier(partner load)", "DMA issue"], segc)) + f"  total {tot:.0f}")
