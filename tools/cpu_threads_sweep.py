#!/usr/bin/env python3
"""Host-side sweep for bench.py's cpu_baseline leg: eager PyTorch-CPU train_step (oracle/torch_ref.py) at several thread counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import gan_oracle as O
from oracle import torch_ref as TR
Gp, Dp = O.init_generator(1, seed=11), O.init_discriminator(1, True, seed=12)
inp, tar = O.synthetic_pair(1, 256, 1, seed=123)
masks = O.dropout_masks(1, 256, seed=5)
for nt in (8, 16, 32, 64, 128):
    if nt > (os.cpu_count() or 1):
        break
    torch.set_num_threads(nt)
    Gt, Dt = TR.params(Gp, torch.float32), TR.params(Dp, torch.float32)
    ti, tt = TR.t(inp, torch.float32), TR.t(tar, torch.float32)
    mt = [TR.t(m, torch.float32) for m in masks]
    state = {}
    TR.pix2pix_train_step_eager(Gt, Dt, state, ti, tt, 100.0, mt)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 6:
        TR.pix2pix_train_step_eager(Gt, Dt, state, ti, tt, 100.0, mt); n += 1
    print(f"threads {nt}: {n / (time.perf_counter() - t0):.3f} img/s", flush=True)
