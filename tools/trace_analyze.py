#!/usr/bin/env python3
"""Timeline of the LAST graph-replayed step in a rocprofv3 kernel trace (tools/trace_step.sh):
per queue occupancy, idle gaps, and the kernels in start order with their concurrency."""
import csv, re, sys
args = [a for a in sys.argv[1:] if not a.startswith('-')]
path = args[0] if args else 'gpurun_out/trace/tr_kernel_trace.csv'
rows = list(csv.DictReader(open(path)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id'], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Z'])) for r in rows]
ev.sort()
def short(n):
    n = re.sub(r'^void ', '', n)
    m = re.match(r'([A-Za-z0-9_:]+)<([^(]*)>?\(', n)
    base = n.split('(')[0]
    base = base.replace('__hip_bfloat16', 'bf16').replace('__bf16', 'bf16')
    return base[:70]
# steps: find the adam_begin kernels ... simpler: the pack kernel starts a step
starts = [i for i, e in enumerate(ev) if 'pack_multi' in e[2] or 'pack4' in e[2]]
if len(starts) < 2:
    starts = [i for i, e in enumerate(ev) if 'pack' in e[2]]
print("step starts:", len(starts))
a, b = starts[-2], starts[-1]
step = ev[a:b]
t0 = step[0][0]; t1 = max(e[1] for e in step)
print(f"step span {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels, sum of durations {sum(e[1] - e[0] for e in step) / 1e3:.1f} us")
# union busy
iv = sorted((e[0], e[1]) for e in step)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f"union busy {busy / 1e3:.1f} us (idle {(t1 - t0 - busy) / 1e3:.1f} us)")
# concurrency-weighted time
pts = sorted([(e[0], 1) for e in step] + [(e[1], -1) for e in step])
lvl = 0; last = pts[0][0]; hist = {}
for t, d in pts:
    hist[lvl] = hist.get(lvl, 0) + t - last; last = t; lvl += d
print("time at concurrency level:", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
if '-v' in sys.argv:
    for s, e, n, q, g in step:
        conc = sum(1 for x in step if x[0] < e and x[1] > s) - 1
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{q:>2} blocks {g:6d} conc {conc}  {short(n)}")
agg = {}
for s, e, n, q, g in step:
    k = short(n); agg[k] = agg.get(k, [0, 0]); agg[k][0] += e - s; agg[k][1] += 1
print("---- per kernel (us per step, launches)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{v[0] / 1e3:9.1f} {v[1]:4d}  {k}")
