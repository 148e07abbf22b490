#!/usr/bin/env python3
"""Per-node cost of hipGraph replay on this stack: N tiny dependent kernels (gan_sum3 on 1 element) as one chain, as two /
four independent chains on forked streams, and stream-launched without a graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.nets import Ctx
from gan_amd import _lib as L

ctx = Ctx('cuda:0', 'bf16')
lib = ctx.lib
N = 1024
buf = torch.zeros(64, dtype=torch.float32, device='cuda')
p = buf.data_ptr()


def chain(n, stream, off):
    for _ in range(n):
        lib.gan_sum3(p + off, p + off + 4, p + off + 8, p + off + 12, 1, stream)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for lanes in (1, 2, 4):
    g = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream() for _ in range(lanes - 1)]
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            for s in side:
                s.wait_stream(cap)
            chain(N // lanes, cap.cuda_stream, 0)
            for i, s in enumerate(side):
                chain(N // lanes, s.cuda_stream, 64 * (i + 1))
            for s in side:
                cap.wait_stream(s)
    t = timed(g.replay)
    print(f"graph, {lanes} chain(s): {N} kernels in {t*1e3:.3f} ms = {t/N*1e6:.2f} us per kernel")
st = torch.cuda.current_stream().cuda_stream
t = timed(lambda: chain(N, st, 0), reps=5)
print(f"stream launches from python/ctypes: {t/N*1e6:.2f} us per kernel")

# cost of a graph boundary: K graphs of N/K dependent tiny kernels replayed back to back on one stream
for parts in (1, 2, 4, 8):
    gs = []
    for _ in range(parts):
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                chain(N // parts, cap.cuda_stream, 0)
        gs.append(g)
    def run():
        for g in gs:
            g.replay()
    t = timed(run)
    print(f"{parts} graph(s) of {N // parts} kernels back to back: {t*1e3:.3f} ms")
