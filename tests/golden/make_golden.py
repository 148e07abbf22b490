#!/usr/bin/env python3
"""Generates the committed golden vectors (run in the authoring container, where /root/reference is
mounted; nothing at test time reads the reference).

Inputs: the reference's two example pairs `example_images/example{1,2}.png` (1280x512, mode L, thermal left |
visible right — the format `Pix2Pix.split_img` expects, pix2pix.py:43-52), split and nearest-neighbour resized
to 256x256 exactly as `process_images_pred` (pix2pix.py:101-112), stored as uint8.
Expected outputs: the fp64 CPU oracle's Pix2Pix train_step on that batch of 2 with seeded N(0,0.02) weights
(numpy default_rng seeds 11 / 12, see oracle.init_*), dropout masks seed 5, lambda 100: generator output, the
four losses, per-tensor gradient checksums and a slice of post-Adam weights.

NOTE these expected values come from this repo's oracle, not from TensorFlow (not installable here): they pin
the oracle against drift and give the GPU path a fixed target; they do not pin parity with TF."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import gan_oracle as O     # noqa: E402
from tests.golden import independent_io as IO     # noqa: E402  (NOT gan_amd.data: the fixture must be able to catch a bug there)

REF = '/root/reference/example_images'


def main():
    pairs = []
    for name in ('example1.png', 'example2.png'):
        a, b = IO.split_left_right(IO.decode_gray(os.path.join(REF, name)))
        pairs.append((IO.resize_nn(a, 256, 256)[..., None], IO.resize_nn(b, 256, 256)[..., None]))
    inp_u8 = np.stack([p[0] for p in pairs]).astype(np.uint8)
    tar_u8 = np.stack([p[1] for p in pairs]).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, 'example_pairs_256.npz'), input_u8=inp_u8, target_u8=tar_u8)
    inp, tar = O.normalize(inp_u8).astype(np.float64), O.normalize(tar_u8).astype(np.float64)
    G = O.init_generator(1, seed=11, dtype=np.float64)
    Dp = O.init_discriminator(1, True, seed=12, dtype=np.float64)
    masks = O.dropout_masks(2, 256, seed=5, dtype=np.float64)
    out = O.pix2pix_train_step(G, Dp, O.AdamTF(), O.AdamTF(), inp, tar, 100.0, masks, True, return_grads=True)
    losses, gen, gG, gD = np.array(out[:4], np.float64), out[4], out[5], out[6]
    gsum = {('G.' + k): np.array([v.sum(), np.abs(v).sum()]) for k, v in gG.items()}
    gsum.update({('D.' + k): np.array([v.sum(), np.abs(v).sum()]) for k, v in gD.items()})
    np.savez_compressed(os.path.join(HERE, 'golden_pix2pix_step.npz'), losses=losses, gen=gen.astype(np.float32),
                        grad_names=np.array(sorted(gsum)), grad_sums=np.stack([gsum[k] for k in sorted(gsum)]),
                        new_G_down3_kernel_slice=G['down3.kernel'][0, 0, :8, :8], new_D_conv_kernel_slice=Dp['conv.kernel'][1, 2, :8, :8])
    print("losses", losses, "gen range", gen.min(), gen.max())


if __name__ == '__main__':
    main()
