#!/usr/bin/env python3
"""Micro-benchmark of one wgrad op: python tools/bench_wgrad.py N H Ca Cb stride"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx, Buf
N, H, ca, cb, s = map(int, sys.argv[1:6])
ctx = Ctx('cuda:0', os.environ.get('DT', 'bf16'))
Ho = (H + 2 - 4) // s + 1
big, small = Buf(ctx, N, H, H, ca), Buf(ctx, N, Ho, Ho, cb)
big.t.normal_(); small.t.normal_()
dw = torch.zeros(16, ca, cb, device='cuda')
d = L.GanWgradDesc(ctx.dt, s, big.view(), small.view(), dw.data_ptr(), ca, cb, 0, ctx.ws_ptr, ctx.ws_bytes)
info = (C.c_int32 * 4)(); ctx.lib.gan_wgrad_plan_info(C.byref(d), info)
for _ in range(3): assert ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream()) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ctx.lib.gan_conv_wgrad(C.byref(d), ctx.stream())
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * N * Ho * Ho * 16 * ca * cb
print(f"wgrad N{N} H{H} A{ca} B{cb} s{s}: tile {info[0]}x{info[1]} splits {info[2]}: {ms*1e3:.1f} us {fl/ms/1e9:.1f} TF/s")
