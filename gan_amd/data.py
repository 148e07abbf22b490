"""Host-side input pipeline: the reference's tf.data / tf.image steps restated with
PIL + numpy — decode, left/right split, nearest-neighbour resize, random jitter, normalise, batch
(base_gan.py:26-61, pix2pix.py:34-165, cycle_gan.py:40-152).  CPU I/O, not part of the accelerated path; a
background thread decodes ahead (CPU only), the consuming thread pins and uploads."""
from __future__ import annotations

import os
import queue
import random
import threading

import numpy as np


def load(image_file: str, channels: int) -> np.ndarray:
    """base_gan.py:26-44: decode PNG/JPEG to `channels` channels, float32 HWC in [0,255]."""
    from PIL import Image
    with Image.open(image_file) as im:
        im = im.convert('L' if channels == 1 else 'RGB')
        a = np.asarray(im, dtype=np.float32)
    return a[..., None] if a.ndim == 2 else a


def resize_nearest(image: np.ndarray, height: int, width: int) -> np.ndarray:
    """tf.image.resize(method=NEAREST_NEIGHBOR) (base_gan.py:46-54): half-pixel centres,
    src = floor((dst + 0.5) * in / out)."""
    h, w = image.shape[:2]
    ys = np.minimum(np.floor((np.arange(height) + 0.5) * (h / height)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor((np.arange(width) + 0.5) * (w / width)).astype(np.int64), w - 1)
    return image[ys][:, xs]


def normalize(image: np.ndarray) -> np.ndarray:
    """base_gan.py:56-61"""
    return (image / np.float32(127.5)) - np.float32(1)


def split_img(image: np.ndarray, input_img_orient: str = 'left'):
    """pix2pix.py:34-54: split a side-by-side pair at w // 2."""
    w = image.shape[1] // 2
    if input_img_orient == 'left':
        return image[:, :w, :], image[:, w:, :]
    return image[:, w:, :], image[:, :w, :]


def random_jitter_pair(a, b, size, rng):
    """pix2pix.py:70-87: resize to size+30, joint random crop, joint random mirror."""
    a = resize_nearest(a, size + 30, size + 30)
    b = resize_nearest(b, size + 30, size + 30)
    y, x = rng.integers(0, 31), rng.integers(0, 31)
    a, b = a[y:y + size, x:x + size], b[y:y + size, x:x + size]
    if rng.random() > 0.5:
        a, b = a[:, ::-1], b[:, ::-1]
    return a, b


def random_jitter_single(a, size, rng):
    """cycle_gan.py:58-72: same for one unpaired image."""
    a = resize_nearest(a, size + 30, size + 30)
    y, x = rng.integers(0, 31), rng.integers(0, 31)
    a = a[y:y + size, x:x + size]
    if rng.random() > 0.5:
        a = a[:, ::-1]
    return a


def list_images(path):
    return sorted(i for i in os.listdir(path) if 'png' in i or 'jpg' in i)


def pix2pix_split(contents, seed, test_img, validation_size):
    """Seeded train/val/test split exactly as pix2pix.py:136-147."""
    random.seed(seed)
    test = random.sample(contents, test_img)
    val_obs = int(np.ceil((len(contents) - test_img) * validation_size))
    val = random.sample([i for i in contents if i not in test], val_obs)
    train = [i for i in contents if i not in test and i not in val]
    train = random.sample(train, len(train))
    return train, val, test


class Batches:
    """Re-iterable batched dataset.  `make_example(path) -> tuple of HWC float32 arrays`; batches are tuples of
    NHWC float32 torch tensors on `device`, the last partial batch kept (no drop_remainder, pix2pix.py:163)."""

    def __init__(self, files, make_example, batch_size, device=None, shuffle_seed=None, prefetch=2):
        self.files, self.make_example, self.bs = list(files), make_example, batch_size
        self.device, self.shuffle_seed, self.prefetch = device, shuffle_seed, prefetch
        self.epoch = 0

    def __len__(self):
        return (len(self.files) + self.bs - 1) // self.bs

    def _produce(self, files, q, stop):
        """Worker thread: decode + augment + stack on the CPU only.  It never touches the GPU (a pin_memory() or an upload
        from here could land inside the main thread's hipGraph capture and invalidate it); errors travel to the consumer."""
        import torch
        try:
            for i in range(0, len(files), self.bs):
                ex = [self.make_example(f) for f in files[i:i + self.bs]]
                item = tuple(torch.from_numpy(np.ascontiguousarray(np.stack([e[k] for e in ex]))) for k in range(len(ex[0])))
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if stop.is_set():
                    return
            item = None
        except BaseException as e:          # corrupt image, bad path ...: re-raised by __iter__
            item = e
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return
            except queue.Full:
                continue

    def __iter__(self):
        files = self.files
        if self.shuffle_seed is not None:
            r = random.Random(self.shuffle_seed + self.epoch)
            files = r.sample(files, len(files))
        self.epoch += 1
        q, stop = queue.Queue(maxsize=self.prefetch), threading.Event()
        th = threading.Thread(target=self._produce, args=(files, q, stop), daemon=True)
        th.start()
        cuda = self.device is not None and str(self.device).startswith('cuda')
        try:
            while True:
                b = q.get()
                if b is None:
                    return
                if isinstance(b, BaseException):
                    raise b
                if self.device is not None:      # upload in the consumer (the thread that owns the stream / any capture)
                    b = tuple((t.pin_memory() if cuda else t).to(self.device, non_blocking=cuda) for t in b)
                yield b
        finally:                                  # abandoned iterator (zip() of unequal sets, next(iter(..))): release the worker
            stop.set()

    def unbatch(self):
        for f in self.files:
            yield self.make_example(f)
