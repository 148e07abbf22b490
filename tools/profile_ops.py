#!/usr/bin/env python3
"""Per-op timing of one eager Pix2Pix/CycleGAN train_step (HIP events around every C-ABI call)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.nets import Ctx
from gan_amd.steps import Pix2PixStep, CycleGANStep

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16); ap.add_argument('--img-size', type=int, default=256)
ap.add_argument('--dtype', default='bf16'); ap.add_argument('--model', default='pix2pix'); ap.add_argument('--reps', type=int, default=5)
a = ap.parse_args()
ctx = Ctx('cuda:0', a.dtype, lanes=False)
step = (Pix2PixStep if a.model == 'pix2pix' else CycleGANStep)(ctx, a.batch, a.img_size, 1)
x = [torch.rand(a.batch, a.img_size, a.img_size, 1, device='cuda') * 2 - 1 for _ in range(2)]
recs = []
orig = ctx.run
def timed(ops, lane=0):
    st = ctx.stream()
    for op in ops:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = op[0](*op[1], st); e1.record()
        assert rc == 0, op[2]
        recs.append((op[2], op[3] if len(op) > 3 else None, e0, e1))
for _ in range(2):
    step._run(*x, training=True)
ctx.run = timed
for _ in range(a.reps):
    step._run(*x, training=True)
torch.cuda.synchronize()
n = len(recs) // a.reps
tot = 0.0
agg = {}
for i in range(n):
    ms = sum(recs[r * n + i][2].elapsed_time(recs[r * n + i][3]) for r in range(a.reps)) / a.reps
    lab, meta = recs[i][0], recs[i][1]
    tot += ms
    if meta:
        print(f"{i:4d} {ms*1e3:9.1f} us  {meta['flops']/ms/1e9:8.1f} TF/s  {meta['kernel']:28s} {meta['shape']}")
    else:
        print(f"{i:4d} {ms*1e3:9.1f} us  {lab}")
    key = meta['kernel'] if meta else lab.split('(')[0]
    agg[key] = agg.get(key, 0) + ms
print("---- per kernel family (ms/step)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{v:8.3f}  {k}")
print(f"total (eager, event-bracketed) {tot:.3f} ms")
