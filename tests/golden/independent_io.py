"""Input-side restatement used ONLY to build and check fixtures, written independently of gan_amd/data.py (the product's input
pipeline) so that a fixture can catch a split / resize bug there: decode with PIL, split at w // 2 (pix2pix.py:43-52), nearest-
neighbour resize with TensorFlow-2 semantics (tf.image.resize(method=NEAREST_NEIGHBOR), base_gan.py:46-54: half-pixel centres,
src = floor((dst + 0.5) * in / out)) - element by element in plain Python integer arithmetic, no vectorised index tricks shared
with the product code."""
from fractions import Fraction

import numpy as np
from PIL import Image


def decode_gray(path):
    with Image.open(path) as im:
        return np.array(im.convert('L'), dtype=np.uint8)


def split_left_right(img):
    half = img.shape[1] // 2
    return img[:, :half].copy(), img[:, half:].copy()       # (an odd width leaves the extra column on the right, as image[:, w:, :] does)


def nearest_index(dst, n_in, n_out):
    """floor((dst + 1/2) * n_in / n_out) in exact rational arithmetic, clamped to the last source element."""
    return min(int((Fraction(2 * dst + 1, 2) * n_in) // n_out), n_in - 1)


def resize_nn(img, out_h, out_w):
    h, w = img.shape[:2]
    out = np.empty((out_h, out_w) + img.shape[2:], dtype=img.dtype)
    rows = [nearest_index(y, h, out_h) for y in range(out_h)]
    cols = [nearest_index(x, w, out_w) for x in range(out_w)]
    for y, sy in enumerate(rows):
        for x, sx in enumerate(cols):
            out[y, x] = img[sy, sx]
    return out
