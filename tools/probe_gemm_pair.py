#!/usr/bin/env python3
"""Do two GEMM launch streams OVERLAP?  Two different convolution layers on two HIP streams, R launches each: alone, alone, together;
overlap = (T_a + T_b - T_both) / min(T_a, T_b).  Run once with the default plans (256-row tiles: one 512-thread workgroup per CU) and
once with --opt conv.big_tiles=0 (128x128 tiles, 256-thread workgroups, two per CU).  python tools/probe_gemm_pair.py [--opt K=V ...]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx, Buf

ap = argparse.ArgumentParser()
ap.add_argument('--opt', action='append', default=[])
a = ap.parse_args()
for kv in a.opt:
    k, v = kv.split('=', 1)
    L.set_option(k, int(v))
ctx = Ctx('cuda:0', 'bf16')
lib = ctx.lib
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def conv_desc(op, N, H, ci, co, s, ws):
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
    d = L.GanConvDesc(ctx.dt, s, x.view(), y.view(), w.data_ptr(), co, None, 0, 0.3, 0, ctx.ws_lanes[ws].data_ptr(), ctx.ws_lanes[ws].numel(), None, 0, 0)
    fn = [lib.gan_conv2d_fwd, lib.gan_conv2d_dgrad, lib.gan_convT2d_fwd, lib.gan_convT2d_dgrad][opi]
    info = (C.c_int32 * 5)(); lib.gan_conv_plan_info(C.byref(d), opi, info)
    return (fn, d, (x, y, w)), f"{op} N{N} H{H} {ci}->{co} s{s} tile {info[0]}x{info[1]} s{info[2]}"


def run(ga, gb, Ra, Rb):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    if Ra:
        e[0].record(s1)
        for _ in range(Ra):
            assert ga[0](C.byref(ga[1]), s1.cuda_stream) == 0
        e[1].record(s1)
    if Rb:
        e[2].record(s2)
        for _ in range(Rb):
            assert gb[0](C.byref(gb[1]), s2.cuda_stream) == 0
        e[3].record(s2)
    torch.cuda.synchronize()
    ta = e[0].elapsed_time(e[1]) * 1e3 if Ra else 0.0
    tb = e[2].elapsed_time(e[3]) * 1e3 if Rb else 0.0
    if Ra and Rb:
        return ta, tb, max(e[0].elapsed_time(e[1]), e[0].elapsed_time(e[3]), e[2].elapsed_time(e[1]), e[2].elapsed_time(e[3])) * 1e3
    return ta, tb, max(ta, tb)


PAIRS = [(('convT_dgrad', 16, 64, 128, 512, 2), ('conv_fwd', 16, 128, 64, 128, 2)),      # G.up5 dgrad | G.down1 forward
         (('conv_dgrad', 16, 32, 256, 128, 2), ('convT_fwd', 16, 32, 512, 128, 2)),      # G.down2 dgrad | G.up5 forward
         (('conv_fwd', 32, 64, 128, 256, 2), ('conv_dgrad', 32, 32, 256, 128, 2))]       # D.down2 forward | D.down2 dgrad
for pa, pb in PAIRS:
    ga, na = conv_desc(*pa, 0)
    gb, nb = conv_desc(*pb, 2)
    run(ga, gb, 5, 5)
    R = 40
    ta = min(run(ga, gb, R, 0)[0] for _ in range(3))
    tb1 = min(run(ga, gb, 0, R)[1] for _ in range(3)) / R
    Rb = max(1, int(round(ta / tb1)))
    tb = min(run(ga, gb, 0, Rb)[1] for _ in range(3))
    both = min(run(ga, gb, R, Rb)[2] for _ in range(3))
    ov = (ta + tb - both) / min(ta, tb)
    print(f"{na} x{R}: {ta:.0f} us ({ta / R:.1f} each) | {nb} x{Rb}: {tb:.0f} us ({tb / Rb:.1f} each) | together {both:.0f} us -> overlap {ov:.2f}", flush=True)
