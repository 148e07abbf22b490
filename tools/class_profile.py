#!/usr/bin/env python3
"""Per-CLASS kernel statistics of the replayed (captured) step, reproducible from profiles/: every launch of the step gets the
bench.py class label and its algorithmic FLOPs beside the duration rocprofv3 measured for it.

  run   (the program rocprofv3 traces)   rocprofv3 --kernel-trace --output-format csv -d D -o t -- \\
                                             python3 tools/class_profile.py run --map D/launch_map.json [--steps 20] [--multi-lane]
        builds the bench.py step (Pix2Pix 256x256 bf16 batch 16, captured graph; ONE stream unless --multi-lane), records through the
        library's launch log (gan_launch_log, include/gan_amd.h) the kernel symbol of every launch of the CAPTURED step in enqueue order
        together with the op that made it (class label, FLOPs: gan_amd/nets.py metadata = what bench.py's `kernels` reports), then
        replays the graph --steps times.
  join  python3 tools/class_profile.py join D/t_kernel_trace.csv D/launch_map.json OUT_PREFIX
        takes the LAST steps x N dispatches of the trace (N = launches per step), checks symbol by symbol that they are the logged
        launches, and writes OUT_PREFIX_launches.csv (one row per launch of a step: class, op, symbol, flops, avg/min/max us over the
        replays) and OUT_PREFIX_classes.csv (per class: launches, flops_per_launch, us per step, TFLOP/s, frac of 2.5 PF).
        frac(class) = sum(flops) / sum(duration) / 2.5e15 is one line of arithmetic on either file.
"""
import argparse
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MFMA_PEAK = 2.5e15


def friendly(sym):
    """Class label of a kernel symbol that no GEMM op claimed (streaming kernels): the function name without its arguments."""
    s = re.sub(r'^void ', '', sym)
    m = re.match(r'_Z\d+([A-Za-z0-9_]+?)(I.*)?$', s)
    if m and s.startswith('_Z'):
        n = re.match(r'_Z(\d+)', s)
        k = int(n.group(1))
        return s[2 + len(n.group(1)):2 + len(n.group(1)) + k]
    return s.split('(')[0].split('<')[0][:60]


def cmd_run(a):
    import torch
    from gan_amd import _lib as L
    from gan_amd.nets import Ctx, workspace_mb_for
    from gan_amd.steps import CycleGANStep, Pix2PixStep
    for kv in a.opt:
        k, v = kv.split('=', 1)
        L.set_option(k, int(v))
    B, S = a.batch, a.img_size
    ctx = Ctx('cuda:0', a.dtype, workspace_mb=workspace_mb_for(B, S), lanes=a.multi_lane)
    step = (Pix2PixStep if a.model == 'pix2pix' else CycleGANStep)(ctx, B, S, 1, lam=100.0 if a.model == 'pix2pix' else 10.0, seed=123)
    ops_seen = []            # (log index before, log index after, label, meta)
    nlog = lambda: len(L.launch_log())

    def wrap(run_ops):
        def f(ops, *rest, **kw):
            for op in ops:
                i0 = nlog()
                run_ops([op], *rest, **kw)
                meta = op[3] if len(op) > 3 and isinstance(op[3], dict) else None
                ops_seen.append((i0, nlog(), op[2], meta))
        return f
    ctx.run, ctx.run_on = wrap(ctx.run), wrap(ctx.run_on)
    cap0 = []
    orig_capture = ctx.capture_graph

    def capture_graph(fn, *r):
        cap0.append(nlog())
        return orig_capture(fn, *r)
    ctx.capture_graph = capture_graph
    L.set_option('diag.launch_log', 1)
    replay = step.capture(training=True)
    log = L.launch_log()
    L.set_option('diag.launch_log', 0)
    start = cap0[-1]
    launches = [dict(symbol=s, cls=None, flops=0.0, op=None) for s in log[start:]]
    for i0, i1, label, meta in ops_seen:
        if i1 <= start or i1 == i0:
            continue
        for j in range(i0, i1):
            e = launches[j - start]
            e['op'] = label if not meta else meta.get('shape', label)
            if meta and meta.get('kind') == 'gemm':
                e['cls'] = meta['kernel']
                e['flops'] = meta['flops'] if j == i0 else 0.0       # an op's FLOPs sit on its first kernel (a split-K op's reduce rides in the class)
    for e in launches:
        if e['cls'] is None:
            e['cls'] = friendly(e['symbol'])
    g = torch.Generator(device='cpu').manual_seed(123)
    for dst in replay.inputs:
        dst.copy_((torch.randint(0, 256, tuple(dst.shape), generator=g).float() / 127.5 - 1.0).to(dst.device))
    torch.cuda.synchronize()
    for _ in range(a.steps):
        replay(*replay.inputs)
    torch.cuda.synchronize()
    losses = step.losses.cpu().tolist()
    json.dump(dict(model=a.model, batch=B, img_size=S, dtype=a.dtype, multi_lane=a.multi_lane, steps=a.steps, options=a.opt,
                   launches_per_step=len(launches), launches=launches, losses=losses[:4]), open(a.map, 'w'))
    print(f"class_profile: {len(launches)} launches per captured step, {a.steps} replays, losses {losses[:4]}")


def cmd_join(a):
    m = json.load(open(a.map))
    L_, steps = m['launches'], m['steps']
    N = len(L_)
    rows = [r for r in csv.DictReader(open(a.trace)) if r['Kind'] == 'KERNEL_DISPATCH']
    if m['multi_lane']:                  # lanes: dispatch order differs from enqueue order
        raise SystemExit("join: use the single-stream run for the class table")
    # one stream: dispatch order = enqueue order.  (Not the start timestamps: in a kernel trace that is not serialised the start of a
    # dependent kernel can read up to ~1 us before the start of its predecessor - 7 such pairs in 3,740 dispatches of one run.)
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    base = lambda s: friendly(s) if s.startswith('_Z') else re.match(r'(?:void\s+)?([A-Za-z0-9_:]+)', s).group(1)

    def same(got, sym):                  # rocprofv3 prints some names mangled and some demangled (with its own template spelling)
        return got == sym or base(got) == base(sym)
    # the replays: the last `steps` runs of N dispatches that start with the step's first kernel, N apart
    firsts = [i for i, r in enumerate(rows) if same(r['Kernel_Name'], L_[0]['symbol'])]
    starts = [i for i in firsts if i + N <= len(rows)][-steps:]
    assert len(starts) == steps and all(b - a == N for a, b in zip(starts, starts[1:])), (len(rows), N, starts[-4:])
    per = [[] for _ in range(N)]
    for s0 in starts:
        for j in range(N):
            r = rows[s0 + j]
            assert same(r['Kernel_Name'], L_[j]['symbol']), f"launch {j} of a step: trace has {r['Kernel_Name']!r}, the launch log {L_[j]['symbol']!r}"
            per[j].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    with open(a.out + '_launches.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['launch', 'class', 'op', 'symbol', 'flops', 'avg_us', 'min_us', 'max_us', 'replays'])
        for j, (e, d) in enumerate(zip(L_, per)):
            w.writerow([j, e['cls'], e['op'] or '', e['symbol'], f"{e['flops']:.0f}", f"{sum(d) / len(d) / 1e3:.3f}", f"{min(d) / 1e3:.3f}",
                        f"{max(d) / 1e3:.3f}", len(d)])
    agg = {}
    for e, d in zip(L_, per):
        c = agg.setdefault(e['cls'], dict(kernels=0, ops=0, flops=0.0, ns=0.0, symbols=set()))
        c['kernels'] += 1
        c['ops'] += 1 if e['flops'] > 0 else 0
        c['flops'] += e['flops']
        c['ns'] += sum(d) / len(d)
        c['symbols'].add(friendly(e['symbol']))
    tot = sum(c['ns'] for c in agg.values())
    with open(a.out + '_classes.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['class', 'launches_per_step', 'kernels_per_step', 'flops_per_launch', 'flops_per_step', 'us_per_step', 'avg_launch_us', 'tflops',
                    'frac_of_2.5PF', 'share_of_kernel_time', 'kernel_functions'])
        for k, c in sorted(agg.items(), key=lambda kv: -kv[1]['ns']):
            n = max(c['ops'], 1) if c['flops'] else c['kernels']
            tf = c['flops'] / (c['ns'] * 1e-9) / 1e12 if c['flops'] else 0.0
            w.writerow([k, n, c['kernels'], f"{c['flops'] / n:.0f}", f"{c['flops']:.0f}", f"{c['ns'] / 1e3:.2f}", f"{c['ns'] / 1e3 / n:.2f}", f"{tf:.1f}",
                        f"{tf * 1e12 / MFMA_PEAK:.4f}", f"{c['ns'] / tot:.4f}", ' '.join(sorted(c['symbols']))])
    print(f"{N} launches per step, sum of kernel durations {tot / 1e3:.1f} us per step ({steps} replays); wrote {a.out}_launches.csv, {a.out}_classes.csv")
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1]['ns'])[:14]:
        tf = c['flops'] / (c['ns'] * 1e-9) / 1e12 if c['flops'] else 0.0
        print(f"  {c['ns'] / 1e3:8.1f} us  {c['kernels']:3d} kernels  {tf:7.1f} TF/s  {k}")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest='cmd', required=True)
    r = sub.add_parser('run')
    r.add_argument('--map', required=True)
    r.add_argument('--steps', type=int, default=20)
    r.add_argument('--batch', type=int, default=16)
    r.add_argument('--img-size', type=int, default=256)
    r.add_argument('--dtype', default='bf16')
    r.add_argument('--model', default='pix2pix', choices=['pix2pix', 'cyclegan'])
    r.add_argument('--multi-lane', action='store_true')
    r.add_argument('--opt', action='append', default=[])
    j = sub.add_parser('join')
    j.add_argument('trace')
    j.add_argument('map')
    j.add_argument('out')
    a = ap.parse_args()
    (cmd_run if a.cmd == 'run' else cmd_join)(a)


if __name__ == '__main__':
    main()
