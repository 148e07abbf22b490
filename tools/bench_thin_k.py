#!/usr/bin/env python3
"""thin-K streaming kernel alone (G.down0 / D.down0 forward shapes), per grid cap: python tools/bench_thin_k.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx, Buf

ctx = Ctx('cuda:0', 'bf16')
flush = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')        # > Infinity Cache: a cold-operand variant
for N, H, co in ((16, 256, 64), (32, 256, 64), (16, 256, 128)):
    x, y = Buf(ctx, N, H, H, 8), Buf(ctx, N, H // 2, H // 2, co)
    x.t.copy_(torch.randn_like(x.t.float()).to(ctx.tdtype))
    w = (torch.randn(16, co, 8, device='cuda') * 0.05).to(ctx.tdtype)
    d = L.GanConvDesc(ctx.dt, 2, x.view(), y.view(), w.data_ptr(), co, None, L.ACTS['lrelu'], 0.3, 0, ctx.ws_ptr, ctx.ws_bytes)
    info = (C.c_int32 * 5)()
    ctx.lib.gan_conv_plan_info(C.byref(d), 0, info)
    mb = (x.t.numel() + y.t.numel()) * 2 / 1e6
    for cap in (256, 512, 1024, 2048, 4096, 8192, 16384):
        L.set_option('conv.thin_k_blocks', cap)
        for _ in range(3):
            assert ctx.lib.gan_conv2d_fwd(C.byref(d), ctx.stream()) == 0
        torch.cuda.synchronize()
        res = []
        for cold in (False, True):
            ts = []
            for _ in range(10):
                if cold:
                    flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ctx.lib.gan_conv2d_fwd(C.byref(d), ctx.stream()); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res.append(sorted(ts)[len(ts) // 2])
        print(f"N{N} {H}x{H} 8->{co} plan {list(info)[:3]} {mb:.0f} MB  blocks {cap:6d}: warm {res[0]:6.1f} us ({mb / res[0]:.2f} TB/s incl. the event bracket)  cold {res[1]:6.1f} us", flush=True)
    L.set_option('conv.thin_k_blocks', 2048)
