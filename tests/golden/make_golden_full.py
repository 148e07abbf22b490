#!/usr/bin/env python3
"""Full-size golden vectors: the fp64 CPU oracle's train_step at the shapes BASELINE.json names, run ONCE in the authoring
container (minutes of CPU each) and committed as small fixtures, so that the `-m gpu` tests hold oracle VALUES - not only
size-independent properties - at the shapes where the planner picks the table-driven / tap-shared / parity-patch / ping-pong
kernels, the fused carriers and the multi-lane captured graph.

  python tests/golden/make_golden_full.py p16        Pix2Pix 256x256 batch 16   (BASELINE config 2; the object bench.py times)
  python tests/golden/make_golden_full.py p512       Pix2Pix 512x512 batch 8    (config 4's per-GPU shape)
  python tests/golden/make_golden_full.py c4         CycleGAN 256x256 batch 4   (config 3)

Inputs are NOT stored: they are regenerated at test time from the seeds below (numpy default_rng: oracle.synthetic_pair,
oracle.init_*, oracle.dropout_masks).  Stored per case: the losses, a strided sample of the generator output(s), for every
gradient tensor [sum, sum|.|, l2] plus a strided sample of <= 4096 elements (index k*stride), slices of post-Adam weights and the
BatchNorm moving statistics (Pix2Pix).

NOTE the expected values come from this repo's oracle, not from TensorFlow (not installable here): they extend
the oracle's reach to full size; they do not pin parity with TF."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import gan_oracle as O     # noqa: E402

SAMPLE = 4096
GEN_STRIDE = 7

# (B, S, seeds): shared with tests/test_gpu_golden_full.py
CASES = {
    'p16': dict(model='pix2pix', B=16, S=256, g_seed=11, d_seed=12, in_seed=123, mask_seed=5, lam=100.0),
    'p512': dict(model='pix2pix', B=8, S=512, g_seed=11, d_seed=12, in_seed=321, mask_seed=6, lam=100.0),
    'c4': dict(model='cyclegan', B=4, S=256, seeds=(21, 22, 23, 24), in_seed=9, mask_seed=40, lam=10.0),
}


def sample(v):
    f = np.asarray(v, np.float64).ravel()
    stride = max(1, f.size // SAMPLE) | 1        # odd: a stride that divides the row length would sample one channel only
    return f[::stride][:SAMPLE].astype(np.float32)


def grad_record(prefix, grads, out):
    for k in sorted(grads):
        v = np.asarray(grads[k], np.float64)
        out[f'gsum/{prefix}.{k}'] = np.array([v.sum(), np.abs(v).sum(), np.sqrt((v * v).sum())])
        out[f'gsample/{prefix}.{k}'] = sample(v)


def f64(P):
    return {k: v.astype(np.float64) for k, v in P.items()}


def pix2pix(c):
    B, S = c['B'], c['S']
    G, D = f64(O.init_generator(1, seed=c['g_seed'])), f64(O.init_discriminator(1, True, seed=c['d_seed']))
    inp, tar = O.synthetic_pair(B, S, 1, seed=c['in_seed'])
    masks = [m.astype(np.float64) for m in O.dropout_masks(B, S, seed=c['mask_seed'])]
    stG, stD = {}, {}
    out = O.pix2pix_train_step(G, D, O.AdamTF(), O.AdamTF(), inp.astype(np.float64), tar.astype(np.float64), c['lam'], masks, True,
                               stateG=stG, stateD=stD, return_grads=True)
    rec = {'losses': np.array(out[:4], np.float64), 'gen_sample': out[4][:, ::GEN_STRIDE, ::GEN_STRIDE, :].astype(np.float32),
           'gen_abs_mean': np.array(np.abs(out[4]).mean())}
    grad_record('G', out[5], rec)
    grad_record('D', out[6], rec)
    for net, P in (('G', G), ('D', D)):                   # post-Adam weights: a strided sample of every kernel
        for k in sorted(P):
            if k.endswith('.kernel'):
                rec[f'new/{net}.{k}'] = sample(P[k])
    for net, st in (('G', stG), ('D', stD)):
        for k, v in st.items():
            rec[f'moving/{net}.{k}'] = v.astype(np.float32)
    return rec


def cyclegan(c):
    B, S = c['B'], c['S']
    n = 'instancenorm'
    s = c['seeds']
    Ps = [f64(O.init_generator(1, n, seed=s[0])), f64(O.init_generator(1, n, seed=s[1])),
          f64(O.init_discriminator(1, False, n, seed=s[2])), f64(O.init_discriminator(1, False, n, seed=s[3]))]
    rx, ry = O.synthetic_pair(B, S, 1, seed=c['in_seed'])
    keys = ['fake_y', 'cycled_x', 'fake_x', 'cycled_y', 'same_x', 'same_y']
    masks = {k: [m.astype(np.float64) for m in O.dropout_masks(B, S, seed=c['mask_seed'] + i)] for i, k in enumerate(keys)}
    out = O.cyclegan_train_step(*Ps, [O.AdamTF() for _ in range(4)], rx.astype(np.float64), ry.astype(np.float64), c['lam'], masks, True,
                                return_grads=True)
    rec = {'losses': np.array(out[:7], np.float64)}
    for k, v in out[7].items():
        rec[f'{k}_sample'] = v[:, ::GEN_STRIDE, ::GEN_STRIDE, :].astype(np.float32)
    for nm, g in zip(('Gg', 'Gf', 'Dx', 'Dy'), out[8:]):
        grad_record(nm, g, rec)
    for nm, P in zip(('Gg', 'Gf', 'Dx', 'Dy'), Ps):
        for k in sorted(P):
            if k.endswith('.kernel'):
                rec[f'new/{nm}.{k}'] = sample(P[k])
    return rec


def main():
    for name in sys.argv[1:] or list(CASES):
        c = CASES[name]
        t0 = time.time()
        rec = pix2pix(c) if c['model'] == 'pix2pix' else cyclegan(c)
        path = os.path.join(HERE, f'golden_full_{name}.npz')
        np.savez_compressed(path, **rec)
        print(f"{name}: losses {rec['losses']}  {time.time() - t0:.0f} s  -> {path} ({os.path.getsize(path) / 1e3:.0f} kB)", flush=True)


if __name__ == '__main__':
    main()
