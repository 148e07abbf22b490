#!/usr/bin/env python3
"""In-kernel time breakdown of ONE column-owner launch (conv_own_kernel; diagnostic library with stamps, tools/diag_build.sh):
   GAN_AMD_LIB=gan_amd/libgan_amd_diag.so python tools/diag_own.py conv_fwd|convT_fwd N H Cin Cout
Per block: entry -> gather table + live-tap lists -> operands loaded and multiplied -> partial tiles exchanged -> layer finished."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx, Buf

op, N, H, ci, co = sys.argv[1], *map(int, sys.argv[2:6])
ctx = Ctx('cuda:0', 'bf16')
lib = ctx.lib
L.set_option('conv.own_max_rows', 64); L.set_option('conv.own_max_kb', 1 << 20)
lib.gan_diag_set.argtypes = [C.c_void_p]
Ho = H // 2 if op == 'conv_fwd' else 2 * H
x, y, a = Buf(ctx, N, H, H, ci), Buf(ctx, N, Ho, Ho, co), Buf(ctx, N, Ho, Ho, co)
x.t.copy_(torch.randn_like(x.t.float()).to(ctx.tdtype))
w = (torch.randn(16, co, ci, device='cuda') * 0.05).to(ctx.tdtype)
f32 = torch.float32
vec = lambda v: torch.full((co,), v, dtype=f32, device='cuda')
gamma, beta, mean, rstd, mm, mv = vec(1.0), vec(0.0), vec(0.0), vec(0.0), vec(0.0), vec(1.0)
part = torch.zeros(1 << 20, dtype=f32, device='cuda')
nf = L.GanNormFuse(a.view(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                   1e-3, 0.99, None, L.ACTS['lrelu'], 0.3, None, None, 0)
d = L.GanConvDesc(ctx.dt, 2, x.view(), y.view(), w.data_ptr(), co, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes,
                  part.data_ptr(), 1, part.numel() * 4, None, C.addressof(nf))
opi = 0 if op == 'conv_fwd' else 2
fn = lib.gan_conv2d_fwd if opi == 0 else lib.gan_convT2d_fwd
info = (C.c_int32 * 5)(); lib.gan_conv_plan_info(C.byref(d), opi, info)
assert info[0] == 0 and info[1] == 8, list(info)
diag = torch.zeros(1 << 12, 16, dtype=torch.int64, device='cuda')
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
    t = t[t[:, 0] != 0][:, :5].astype(np.float64) * 0.01
    seg = np.diff(t, axis=1)
    print(f"{op} N{N} H{H} {ci}->{co} blocks {len(t)} {['warm', 'cold', 'mall'][cold]}: event {e0.elapsed_time(e1) * 1e3:.1f} us, kernel span {t[:, 4].max() - t[:, 0].min():.1f} us, "
          f"start spread {t[:, 0].max() - t[:, 0].min():.2f}")
    print("   mean per block (us): row decode %.2f  loads+MFMA %.2f  exchange %.2f  finish %.2f  total %.2f (max %.2f)" %
          (*seg.mean(axis=0), (t[:, 4] - t[:, 0]).mean(), (t[:, 4] - t[:, 0]).max()))
