#!/usr/bin/env python3
"""Two independent runs of N graph-replayed steps (all lanes on, auto dropout masks) must end in bit-identical losses and
weights: the multi-lane schedule is race-free and every reduction has a fixed order.  python tools/soak_determinism.py [N] [model] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.nets import Ctx
from gan_amd.steps import Pix2PixStep, CycleGANStep

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
model = sys.argv[2] if len(sys.argv) > 2 else 'pix2pix'
batch = int(sys.argv[3]) if len(sys.argv) > 3 else (16 if model == 'pix2pix' else 4)
res = []
for run in range(2):
    ctx = Ctx('cuda:0', 'bf16')
    st = Pix2PixStep(ctx, batch, 256, 1, lam=100.0, seed=123) if model == 'pix2pix' else CycleGANStep(ctx, batch, 256, 1, lam=10.0, seed=123)
    g = torch.Generator(device='cpu').manual_seed(7)
    x = [(torch.randint(0, 256, (st.B, 256, 256, 1), generator=g).float() / 127.5 - 1.0).to(ctx.device) for _ in range(2)]
    w0 = [n.params.master.clone() for n in st.nets()]
    replay = st.capture(training=True)
    for n_, w_ in zip(st.nets(), w0):
        n_.params.master.copy_(w_); n_.params.prepare()
        n_.params.m.zero_(); n_.params.v.zero_(); n_.params.step.zero_()
        for t in n_.params.state.values():
            t.zero_()
    for call in vars(st).values():
        if hasattr(call, 'mask_draws'):
            call.mask_draws.zero_()
    for i in range(N):
        replay(*x)
    torch.cuda.synchronize()
    ctx.assert_no_stack_timeout()          # (a layer-stack kernel whose grid barrier timed out would have raised its flag)
    res.append((st.losses.clone().cpu(), [n.params.master.clone().cpu() for n in st.nets()]))
    print(f"run {run}: losses {res[-1][0][:7].tolist()}", flush=True)
same_l = torch.equal(res[0][0], res[1][0])
same_w = all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
print("bit-identical losses:", same_l, " weights:", same_w)
sys.exit(0 if (same_l and same_w) else 1)
