#!/usr/bin/env python3
"""Marginal cost of op classes in the captured multi-lane Pix2Pix step (timing experiment, wrong results): replay time of the full
step vs the step with every op whose label matches a pattern left out.  Kernel time is not step time when lanes overlap: this
says what removing a class could return at most.
  python tools/marginal.py [--batch 16] [NAME=REGEX ...]"""
import os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.nets import Ctx
from gan_amd.steps import Pix2PixStep
from gan_amd import _lib as L
L.set_option('wgrad.dead_taps', 0)      # (data-dependent early exits would turn a knocked-out normalisation into a free optimiser)

INNER = r'(conv_fwd|convT_fwd|conv_dgrad|convT_dgrad) .*M(16|64|256|1024)(x4)? '
DEFAULT = [
    ("full step", None),
    ("no forward finalize launches", r'^norm_stats_finalize'),
    ("no forward apply passes", r'^norm_act_fwd'),
    ("no backward normalisation ops (partial+finalize+apply)", r'^norm_act_bwd'),
    ("no inner-layer convolutions (M <= 1024)", INNER),
    ("no thin layers", r'(conv_fwd|convT_fwd|conv_dgrad|convT_dgrad) .*(N1 |K128 )'),
    ("no optimiser-carrying inner wgrads (A512)", r'wgrad.* A512 '),
    ("no ping-pong wgrads (>= 30 GF)", r'wgrad.* (A64 B256|A128 B512|A256 B1024|A256 B512|A128 B256|A64 B128) '),
    ("no act_bwd / bias_grad", r'^(act_bwd|bias_grad)'),
    ("full step again", None),
    ("full step again", None),
    ("full step again", None),
    ("full step again", None),
]


def label(op):
    meta = op[3] if len(op) > 3 and isinstance(op[3], dict) else None
    return op[2] + (' ' + meta['shape'] + ' ' if meta else '')


def bench(name, pat, batch):
    ctx = Ctx('cuda:0', 'bf16')
    st = Pix2PixStep(ctx, batch, 256, 1)
    x = [torch.rand(batch, 256, 256, 1, device='cuda') * 2 - 1 for _ in range(2)]
    rx = re.compile(pat) if pat else None
    dropped = []
    orun, orun_on = ctx.run, ctx.run_on

    live = [False]

    def keep(ops):
        if rx is None or not live[0]:
            return ops
        out = []
        for o in ops:
            if rx.search(label(o)):
                dropped.append(label(o))
            else:
                out.append(o)
        return out
    ctx.run = lambda ops, lane=0: orun(keep(ops), lane)
    ctx.run_on = lambda ops, stream: orun_on(keep(ops), stream)
    # every buffer (statistics, activations, gradients) first gets the data of complete steps: a knocked-out op then leaves realistic
    # stale values behind, not zeros (all-zero operands raise the clocks and trip data-dependent early exits)
    for _ in range(2):
        st._run(*x, training=True)
    torch.cuda.synchronize()
    live[0] = True
    rp = st.capture(True)
    nd = len(dropped)
    for _ in range(10):
        rp(*x)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            rp(*x)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 50 * 1e3)
    print(f"{name:60s} {min(ts):.3f} ms/step  (dropped {nd} ops)", flush=True)
    if os.environ.get('MARGINAL_VERBOSE') and rx is not None:
        for d in sorted(set(dropped)):
            print("      -", d)


if __name__ == '__main__':
    batch = 16
    args = sys.argv[1:]
    if args and args[0] == '--batch':
        batch = int(args[1]); args = args[2:]
    sets = [tuple(a.split('=', 1)) for a in args] if args else DEFAULT
    for n, p in sets:
        bench(n, p, batch)
