#!/usr/bin/env python3
"""Do a GEMM launch and an HBM-bound streaming pass OVERLAP when they run on two streams?  For each GEMM class: time R launches
alone, S streaming passes alone, and both together (two streams, started together); overlap = (T_gemm + T_stream - T_both) /
min(T_gemm, T_stream): 1 = the shorter one hides completely, 0 = they time-slice.  python tools/probe_overlap.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx, Buf

ctx = Ctx('cuda:0', 'bf16')
lib = ctx.lib
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def conv_desc(op, N, H, ci, co, s):
    opi = {'conv_fwd': 0, 'conv_dgrad': 1, 'convT_fwd': 2, 'convT_dgrad': 3}[op]
    if op == 'conv_fwd':
        Ho = (H + 2 - 4) // s + 1
        x, y = Buf(ctx, N, H, H, ci), Buf(ctx, N, Ho, Ho, co)
    elif op in ('convT_fwd', 'conv_dgrad'):
        x, y = Buf(ctx, N, H, H, ci), Buf(ctx, N, 2 * H, 2 * H, co)
    else:
        x, y = Buf(ctx, N, H, H, ci), Buf(ctx, N, H // 2, H // 2, co)
    w = (torch.randn(16, co, ci, device='cuda') * 0.05).to(ctx.tdtype)
    x.t.copy_(torch.randn_like(x.t.float()).to(ctx.tdtype))
    d = L.GanConvDesc(ctx.dt, s, x.view(), y.view(), w.data_ptr(), co, None, 0, 0.3, 0, ctx.ws_ptr, ctx.ws_bytes, None, 0, 0)
    fn = [lib.gan_conv2d_fwd, lib.gan_conv2d_dgrad, lib.gan_convT2d_fwd, lib.gan_convT2d_dgrad][opi]
    info = (C.c_int32 * 5)(); lib.gan_conv_plan_info(C.byref(d), opi, info)
    return (fn, d, (x, y, w)), f"{op} N{N} H{H} {ci}->{co} s{s} tile {info[0]}x{info[1]}"


# the streaming pass: normalise + activation of a 67 MB tensor (read y, write a)
yb, ab = Buf(ctx, 16, 128, 128, 128), Buf(ctx, 16, 128, 128, 128)
yb.t.copy_(torch.randn_like(yb.t.float()).to(ctx.tdtype))
f32 = torch.float32
gam, bet = torch.ones(128, dtype=f32, device='cuda'), torch.zeros(128, dtype=f32, device='cuda')
mean, rstd = torch.zeros(128, dtype=f32, device='cuda'), torch.ones(128, dtype=f32, device='cuda')
nd = L.GanNormDesc(ctx.dt, yb.view(), ab.view(), 1, 1e-3, gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, None, 0.99, None,
                   L.ACTS['lrelu'], 0.3, ctx.ws_lanes[1].data_ptr(), ctx.ws_lanes[1].numel())


def run(gemm, R, S):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    if R:
        e[0].record(s1)
        for _ in range(R):
            assert gemm[0](C.byref(gemm[1]), s1.cuda_stream) == 0
        e[1].record(s1)
    if S:
        e[2].record(s2)
        for _ in range(S):
            assert lib.gan_norm_act_fwd(C.byref(nd), s2.cuda_stream) == 0
        e[3].record(s2)
    torch.cuda.synchronize()
    tg = e[0].elapsed_time(e[1]) * 1e3 if R else 0.0
    ts = e[2].elapsed_time(e[3]) * 1e3 if S else 0.0
    if R and S:
        both = max(e[0].elapsed_time(e[1]), e[0].elapsed_time(e[3]), e[2].elapsed_time(e[1]), e[2].elapsed_time(e[3])) * 1e3
        return tg, ts, both
    return tg, ts, max(tg, ts)


for args in (('convT_dgrad', 16, 64, 128, 512, 2), ('convT_fwd', 16, 64, 256, 64, 2), ('conv_fwd', 32, 32, 256, 512, 1), ('conv_fwd', 16, 128, 64, 128, 2)):
    gemm, name = conv_desc(*args)
    run(gemm, 5, 5)
    R = 40
    tg = min(run(gemm, R, 0)[0] for _ in range(3))
    t1 = min(run(gemm, 0, 40)[1] for _ in range(3)) / 40
    S = max(1, int(round(tg / t1)))
    ts = min(run(gemm, 0, S)[1] for _ in range(3))
    both = min(run(gemm, R, S)[2] for _ in range(3))
    ov = (tg + ts - both) / min(tg, ts)
    print(f"{name}: GEMM x{R} {tg:.0f} us ({tg / R:.1f} each); stream x{S} {ts:.0f} us ({ts / S:.1f} each); together {both:.0f} us -> overlap {ov:.2f}")
