#!/usr/bin/env python3
"""Micro-benchmark of ONE conv GEMM op (for rocprofv3 --pmc runs): python tools/bench_op.py conv_fwd 32 32 256 512 1"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx, Buf

op, N, H, ci, co, s = sys.argv[1], *map(int, sys.argv[2:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
ctx = Ctx('cuda:0', os.environ.get('DT', 'bf16'))
if op in ('conv_fwd',):
    Ho = (H + 2 - 4) // s + 1
    x, y = Buf(ctx, N, H, H, ci), Buf(ctx, N, Ho, Ho, co)
    w = torch.randn(16, co, ci, device='cuda').to(ctx.tdtype) * 0.05
    rows, flops = co, 2.0 * N * Ho * Ho * co * ci * 16
elif op == 'convT_fwd':
    x, y = Buf(ctx, N, H, H, ci), Buf(ctx, N, 2 * H, 2 * H, co)
    w = torch.randn(16, co, ci, device='cuda').to(ctx.tdtype) * 0.05
    rows, flops = co, 2.0 * N * 4 * H * H * co * ci * 4
x.t.copy_(torch.randn_like(x.t.float()).to(ctx.tdtype))
d = L.GanConvDesc(ctx.dt, s, x.view(), y.view(), w.data_ptr(), rows, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes)
fn = {'conv_fwd': ctx.lib.gan_conv2d_fwd, 'convT_fwd': ctx.lib.gan_convT2d_fwd}[op]
info = (C.c_int32 * 5)()
ctx.lib.gan_conv_plan_info(C.byref(d), {'conv_fwd': 0, 'convT_fwd': 2}[op], info)
for _ in range(3):
    assert fn(C.byref(d), ctx.stream()) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn(C.byref(d), ctx.stream())
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{op} N{N} H{H} {ci}->{co} s{s}: tile {info[0]}x{info[1]} splits {info[2]}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TF/s")
