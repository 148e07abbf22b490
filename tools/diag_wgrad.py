#!/usr/bin/env python3
"""In-kernel time breakdown of ONE ping-pong wgrad launch (diagnostic library with stamps, tools/diag_build.sh):
   GAN_AMD_LIB=gan_amd/libgan_amd_diag.so python tools/diag_wgrad.py N H big_c small_c stride [concurrent]
Per block: entry -> setup -> first tiles landed -> K loop -> slab store (100 MHz ticks) and the loop's segment cycles of wave 0."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx, Buf

N, H, ca, cb, s = map(int, sys.argv[1:6])
conc = int(sys.argv[6]) if len(sys.argv) > 6 else 0
ctx = Ctx('cuda:0', 'bf16')
lib = ctx.lib
lib.gan_wdiag_set.argtypes = [C.c_void_p]
Hs = (H + 2 - 4) // s + 1
big, small = Buf(ctx, N, H, H, ca), Buf(ctx, N, Hs, Hs, cb)
big.t.copy_(torch.randn_like(big.t.float()).to(ctx.tdtype)); small.t.copy_(torch.randn_like(small.t.float()).to(ctx.tdtype))
dw = torch.zeros(16 * ca * cb, dtype=torch.float32, device='cuda')
d = L.GanWgradDesc(ctx.dt, s, big.view(), small.view(), dw.data_ptr(), ca, cb, 0, ctx.ws_ptr, ctx.ws_bytes, conc, None)
info = (C.c_int32 * 4)(); lib.gan_wgrad_plan_info(C.byref(d), info)
assert info[0] == 256, f"not a ping-pong plan: {list(info)}"
diag = torch.zeros(1 << 14, 16, dtype=torch.int64, device='cuda')
for _ in range(3):
    assert lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
torch.cuda.synchronize()
M = N * Hs * Hs
gf = 2.0 * M * 16 * ca * cb / 1e9
for rep in range(2):
    lib.gan_wdiag_set(diag.data_ptr()); diag.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.gan_conv_wgrad(C.byref(d), ctx.stream()); e1.record()
    torch.cuda.synchronize()
    lib.gan_wdiag_set(None)
    t = diag.cpu().numpy()
    t = t[t[:, 0] != 0]
    segc = t[:, 8:14].astype(np.float64).mean(axis=0)
    t = t[:, :5].astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    seg = np.diff(t, axis=1)
    print(f"wgrad N{N} H{H} A{ca} B{cb} s{s} M{M} {gf:.1f} GF tile {info[0]}x{info[1]} splits {info[2]} blocks {len(t)}: GEMM + reduce under HIP events {e0.elapsed_time(e1)*1e3:.1f} us, "
          f"GEMM kernel span {t[:, 4].max() - t0:.1f} us = {gf / (t[:, 4].max() - t0) * 1e3:.0f} TF/s")
    print(f"   block start spread {t[:, 0].max() - t0:.2f} us; mean per block: setup {seg[:, 0].mean():.2f}  first-fill {seg[:, 1].mean():.2f}  "
          f"loop {seg[:, 2].mean():.2f}  slab store {seg[:, 3].mean():.2f} (max {seg[:, 3].max():.2f})  total {(t[:, 4] - t[:, 0]).mean():.2f} us; "
          f"loop rate {gf / len(t) / seg[:, 2].mean() * 1e3 * min(len(t), 256):.0f} TF/s (x{min(len(t), 256)} CUs)")
    tot = segc.sum()
    if tot > 0:
        print("   loop cycles of wave 0: " + "  ".join(f"{n} {v / tot * 100:.0f}%" for n, v in zip(
            ["frag reads", "vmcnt wait", "barrier(partner math)", "lgkm+MFMA", "barrier(partner load)", "decode+DMA issue"], segc)) + f"  total {tot:.0f}")
