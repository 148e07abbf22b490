#!/usr/bin/env python3
"""Does confining the side lanes of the step to a SUBSET of the CUs pay?  (hipExtStreamCreateWithCUMask; eager steps - a hipGraph kernel
node carries no CU mask, so this only answers the question of principle.)  Pix2Pix 256x256 bf16 batch 16, the multi-lane schedule run
eagerly; lanes 3 + 4 (the generator's wgrad GEMMs, most of them carrying the HBM-bound optimiser step) and optionally lane 2 (the
discriminator's parameter pass) on streams masked to every k-th CU.
   python tools/probe_cu_mask.py [--steps 30]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.nets import Ctx, LaneStream, workspace_mb_for
from gan_amd.steps import Pix2PixStep

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--batch', type=int, default=16)
a = ap.parse_args()
hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


class MaskedLane(torch.cuda.ExternalStream):
    open_in_capture = False

    def wait_stream(self, stream, _join=False):
        return super().wait_stream(stream)


def masked_stream(pred):
    words = (C.c_uint32 * 8)()
    n = 0
    for cu in range(256):
        if pred(cu):
            words[cu // 32] |= 1 << (cu % 32); n += 1
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
    assert rc == 0, rc
    return MaskedLane(h.value), n


B, S = a.batch, 256
ctx = Ctx('cuda:0', 'bf16', workspace_mb=workspace_mb_for(B, S))
step = Pix2PixStep(ctx, B, S, 1, lam=100.0, seed=123)
g = torch.Generator(device='cpu').manual_seed(123)
mk = lambda: (torch.randint(0, 256, (B, S, S, 1), generator=g).float() / 127.5 - 1.0).to('cuda:0')
inputs = (mk(), mk())
orig = list(ctx.side)


def measure(tag):
    for _ in range(5):
        step._run(*inputs, training=True)
    res = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            step._run(*inputs, training=True)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / a.steps * 1e3)
    print(f"{tag:60s} {min(res):.3f} ms/step  ({B / min(res) * 1e3:.0f} img/s)  all {['%.3f' % r for r in res]}", flush=True)


measure("eager, plain streams")
for name, pred, lanes in [("lanes 3+4 on every 2nd CU (128)", lambda c: c % 2 == 0, (2, 3)),
                          ("lanes 3+4 on every 4th CU (64)", lambda c: c % 4 == 0, (2, 3)),
                          ("lanes 3+4 on CUs 0..127", lambda c: c < 128, (2, 3)),
                          ("lanes 2+3+4 on every 2nd CU (128)", lambda c: c % 2 == 0, (1, 2, 3)),
                          ("lanes 3+4 on every 8th CU (32)", lambda c: c % 8 == 0, (2, 3))]:
    ctx.side = list(orig)
    n = 0
    for i in lanes:
        ctx.side[i], n = masked_stream(pred)
    measure(f"{name} [{n} CUs]")
ctx.side = list(orig)
measure("eager, plain streams (again)")
