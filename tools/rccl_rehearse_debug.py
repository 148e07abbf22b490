#!/usr/bin/env python3
"""Debug helper: one RCCL rank, Pix2Pix bucketed schedule with bf16 wire vs single graph: per-bucket gradient comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29655', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
from gan_amd.ddp import GradSync
from gan_amd.nets import Ctx
from gan_amd.steps import Pix2PixStep
res = []
for ddp in (False, True):
    ctx = Ctx('cuda:0', 'bf16')
    st = Pix2PixStep(ctx, 2, 256, 1, lam=100.0, seed=123)
    if ddp:
        st.sync = GradSync([n.params.grad for n in st.nets()], compress_bf16=True, lib=ctx.lib, rehearse=True)
    g = torch.Generator(device='cpu').manual_seed(5)
    x = [(torch.rand(2, 256, 256, 1, generator=g) * 2 - 1).to(ctx.device) for _ in range(2)]
    w0 = [n.params.master.clone() for n in st.nets()]
    replay = st.capture(training=True)
    for n_, w_ in zip(st.nets(), w0):
        n_.params.master.copy_(w_); n_.params.prepare()
        n_.params.m.zero_(); n_.params.v.zero_(); n_.params.step.zero_()
    st.g.mask_draws.zero_()
    out = []
    for it in range(3):
        replay(*x); torch.cuda.synchronize()
        out.append(([n.params.grad.clone() for n in st.nets()], [n.params.master.clone() for n in st.nets()], st.losses.clone()))
    res.append((out, getattr(st, 'buckets', None)))
(one, _), (dd, buckets) = res
for it in range(3):
    print("step", it, "losses", one[it][2][:4].tolist(), dd[it][2][:4].tolist())
    for b, (i, lo, hi) in enumerate(buckets):
        a, c = one[it][0][i][lo:hi], dd[it][0][i][lo:hi]
        rel = float((a - c).norm() / (a.norm() + 1e-30))
        wa, wc = one[it][1][i][lo:hi], dd[it][1][i][lo:hi]
        print(f"  bucket {b} net {i} [{lo},{hi}): grad rel diff {rel:.3e}; weight max diff {float((wa-wc).abs().max()):.3e} mean {float((wa-wc).abs().mean()):.3e}")
dist.destroy_process_group()
