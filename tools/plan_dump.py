#!/usr/bin/env python3
"""Launch plans of every convolution / wgrad of a Pix2Pix or CycleGAN step at a batch size (host-side planners only: runs without
a GPU).  tools/plan_dump.py [--batch 16] [--size 256] [--dtype bf16] [--groups 1|N]"""
import argparse
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _lib as L  # noqa: E402

G_DOWN = [64, 128, 256, 512, 512, 512, 512, 512]
G_UP = [512, 512, 512, 512, 256, 128, 64]


def T(n, h, c, pitch=None):
    return L.GanTensor(16, n, h, h, c, pitch or c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--concurrent', type=int, default=1)
    ap.add_argument('--instancenorm', action='store_true')
    a = ap.parse_args()
    lib = L.load()
    dt = {'f32': 0, 'bf16': 1, 'f16': 2}[a.dtype]
    B, S = a.batch, a.size
    groups = B if a.instancenorm else 1
    info, winfo = (C.c_int32 * 5)(), (C.c_int32 * 4)()

    def conv(tag, op, x, y, w_rows, stride=2, stats=True):
        d = L.GanConvDesc(dt, stride, x, y, 16, w_rows, None, 0, 0.3, 0, 16, 1 << 40, 16 if stats else None, groups if stats else 0, 1 << 30, None, None)
        rc = lib.gan_conv_plan_info(C.byref(d), op, info)
        par = 4 if info[3] == 4 else 1
        M = x.n * (x.h * x.w if par == 4 else y.h * y.w)
        K = (4 if par == 4 else 16) * x.c
        gf = 2.0 * M * par * K * y.c / 1e9
        ws = lib.gan_conv_workspace_bytes(C.byref(d), op)
        print(f"{tag:22s} {['conv_fwd','conv_dgrad','convT_fwd','convT_dgrad'][op]:11s} M{M:6d}x{par} N{y.c:4d} K{K:5d} {gf:7.2f} GF  tile {info[0]:4d}x{info[1]:3d} "
              f"splits {info[2]:3d} chunks {info[4]:4d} slabs {ws / 1e6:7.1f} MB rc={rc}")

    def wgrad(tag, big, small, big_c, small_c, stride=2):
        d = L.GanWgradDesc(dt, stride, big, small, 16, big_c, small_c, 0, 16, 1 << 40, a.concurrent, None)
        rc = lib.gan_wgrad_plan_info(C.byref(d), winfo)
        M = small.n * small.h * small.w
        gf = 2.0 * M * 16 * big_c * small_c / 1e9
        ws = lib.gan_wgrad_workspace_bytes(C.byref(d))
        print(f"{tag:22s} wgrad       M{M:6d}   A{big_c:4d} B{small_c:4d}   {gf:7.2f} GF  tile {winfo[0]:4d}x{winfo[1]:3d} splits {winfo[2]:3d} fold {winfo[3]} "
              f"slabs {ws / 1e6:7.1f} MB params {16 * big_c * small_c / 1e6:5.2f} M rc={rc}")

    hs = [S >> (i + 1) for i in range(8)]
    print(f"== generator, batch {B}, {S}x{S}, {a.dtype}")
    cin = 8
    for i, co in enumerate(G_DOWN):
        h_in = S >> i
        conv(f"G.down{i} fwd", 0, T(B, h_in, cin), T(B, hs[i], co), co, stats=i > 0)
        if i > 0:
            conv(f"G.down{i} dgrad", 1, T(B, hs[i], co), T(B, h_in, cin), cin)
        wgrad(f"G.down{i}", T(B, h_in, cin), T(B, hs[i], co), cin if i else 1, co)
        cin = co
    cin = 512
    for j, co in enumerate(G_UP):
        h_in = hs[7 - j]
        conv(f"G.up{j} fwd", 2, T(B, h_in, cin), T(B, 2 * h_in, co), co)
        conv(f"G.up{j} dgrad", 3, T(B, 2 * h_in, co), T(B, h_in, cin), cin)
        wgrad(f"G.up{j}", T(B, 2 * h_in, co), T(B, h_in, cin), co, cin)
        cin = co + G_DOWN[6 - j]
    print(f"== discriminator, {2 * B} images")
    N = 2 * B
    cin, h = 8, S
    for name, co, stride in [('down0', 64, 2), ('down1', 128, 2), ('down2', 256, 2), ('conv', 512, 1)]:
        ho = h // 2 if stride == 2 else h - 1
        conv(f"D.{name} fwd", 0, T(N, h, cin), T(N, ho, co), co, stride, stats=name != 'down0')
        if name != 'down0':
            conv(f"D.{name} dgrad", 1, T(N, ho, co), T(N, h, cin), cin, stride)
        wgrad(f"D.{name}", T(N, h, cin), T(N, ho, co), cin if name != 'down0' else 2, co, stride)
        cin, h = co, ho


if __name__ == '__main__':
    main()
