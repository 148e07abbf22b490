#!/usr/bin/env python3
"""Concurrent timeline of ONE eager multi-lane Pix2Pix step: HIP events around every C-ABI call on the stream it runs on
(rocprofv3's kernel trace serialises the queues, so it cannot show which kernels share the chip).  Event records stretch the
step a little; the picture (which lane waits for which, where the chip is shared) is what this is for.
  python tools/lane_timeline.py [--opt KEY=VALUE ...] > timeline.txt"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd import _lib as L
from gan_amd.nets import Ctx
from gan_amd.steps import Pix2PixStep

ap = argparse.ArgumentParser()
ap.add_argument('--opt', action='append', default=[])
ap.add_argument('--batch', type=int, default=16)
a = ap.parse_args()
for kv in a.opt:
    k, v = kv.split('=', 1)
    L.set_option(k, int(v))
ctx = Ctx('cuda:0', 'bf16')
st = Pix2PixStep(ctx, a.batch, 256, 1)
x = [torch.rand(a.batch, 256, 256, 1, device='cuda') * 2 - 1 for _ in range(2)]
for _ in range(3):
    st._run(*x, training=True)
torch.cuda.synchronize()
recs = []
lanes = {torch.cuda.current_stream(ctx.device).cuda_stream: 0}
for i, s in enumerate(ctx.side):
    lanes[s.cuda_stream] = i + 1
streams = {torch.cuda.current_stream(ctx.device).cuda_stream: torch.cuda.current_stream(ctx.device)}
for s in ctx.side:
    streams[s.cuda_stream] = s


def wrap(op, st_handle):
    s = streams.get(st_handle)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    rc = op[0](*op[1], st_handle)
    e1.record(s)
    assert rc == 0, op[2]
    meta = op[3] if len(op) > 3 and isinstance(op[3], dict) else None
    recs.append((lanes.get(st_handle, -1), op[2] + (' ' + meta['shape'] if meta else ''), e0, e1))


def run(ops, lane=0):
    main = ctx.lane_stream(lane)
    for op in ops:
        wrap(op, main.cuda_stream)


def run_on(ops, stream):
    for op in ops:
        wrap(op, stream.cuda_stream)


ctx.run, ctx.run_on = run, run_on
base = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
base.record()
st._run(*x, training=True)
torch.cuda.synchronize()
ev = sorted([(base.elapsed_time(e0) * 1e3, base.elapsed_time(e1) * 1e3, lane, name) for lane, name, e0, e1 in recs])
end = max(e[1] for e in ev)
print(f"step span {end:.1f} us (eager, event-bracketed), {len(ev)} ops; sum of op spans {sum(e[1] - e[0] for e in ev):.1f} us")
for lane in sorted(set(e[2] for e in ev)):
    sel = [e for e in ev if e[2] == lane]
    print(f"lane {lane}: {len(sel)} ops, busy {sum(e[1] - e[0] for e in sel):.1f} us, from {sel[0][0]:.1f} to {max(e[1] for e in sel):.1f}")
for s, e, lane, name in ev:
    others = sorted(set(o[2] for o in ev if o[2] != lane and o[0] < e and o[1] > s))
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  L{lane} {'  ' * lane}{name[:70]:70s} || {others}")
