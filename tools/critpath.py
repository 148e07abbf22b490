#!/usr/bin/env python3
"""Marginal cost of the step's side lanes in the captured multi-lane graph (timing experiment, wrong gradients):
replay time of the full step vs the step with G's wgrad GEMMs / D's parameter pass / the Adam kernels left out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.nets import Ctx, GenCall, DiscCall, ParamSet
from gan_amd.steps import Pix2PixStep

def bench(label, patch):
    ctx = Ctx('cuda:0', 'bf16')
    st = Pix2PixStep(ctx, 16, 256, 1)
    undo = patch(ctx, st)
    x = [torch.rand(16, 256, 256, 1, device='cuda') * 2 - 1 for _ in range(2)]
    rp = st.capture(True)
    for _ in range(10): rp(*x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): rp(*x)
    torch.cuda.synchronize()
    print(f"{label:40s} {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step", flush=True)
    undo()

def none(ctx, st): return lambda: None
def no_gwgrad(ctx, st):
    orig = ctx.run_on
    def run_on(ops, stream):
        orig([o for o in ops if not (len(o) > 4 and o[4] and o[2] == 'conv_wgrad' and getattr(run_on, 'g', False))], stream)
    # G's staged wgrads are the only ops run_on() receives with the side flag from GenCall.backward
    import gan_amd.nets as N
    ob = N.GenCall.backward
    def backward(self, *a, **k):
        run_on.g = True
        try: return ob(self, *a, **k)
        finally: run_on.g = False
    N.GenCall.backward = backward; ctx.run_on = run_on
    def undo(): N.GenCall.backward = ob
    return undo
def no_dparams(ctx, st):
    st.d.params_ops = lambda accumulate=False: []
    return lambda: None
def no_adam(ctx, st):
    for P in (st.G.params, st.D.params):
        P.adam = lambda *a, **k: None
        P.adam_segment_ops = lambda *a, **k: []
        P.adam_begin_ops = lambda *a, **k: []
    return lambda: None
def both(ctx, st):
    u = no_gwgrad(ctx, st); no_dparams(ctx, st); return u
def all3(ctx, st):
    u = no_gwgrad(ctx, st); no_dparams(ctx, st); no_adam(ctx, st); return u

for lab, p in (("full step", none), ("without G wgrad GEMMs (lane 3)", no_gwgrad), ("without D parameter pass (lane 2)", no_dparams),
               ("without Adam", no_adam), ("without G wgrads and D pass", both), ("main chain only (no wgrads, D pass, Adam)", all3)):
    bench(lab, p)
