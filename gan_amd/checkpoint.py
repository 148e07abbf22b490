"""Checkpointing with the reference's directory / object naming (pix2pix.py:400-403,419-420,
cycle_gan.py:437-444,460-461): `<dir>/checkpoint` (text, `model_checkpoint_path: "ckpt-N"`),
`ckpt-N.index`, `ckpt-N.data-00000-of-00001`, keep-last-K.  Variable keys follow TF's object-graph naming
(`generator/layer_with_weights-K/.../kernel/.ATTRIBUTES/VARIABLE_VALUE`); optimizer hyper-parameters as
`generator_optimizer/{iter,learning_rate,beta_1,beta_2,decay}/...`, Adam slots as
`generator_optimizer/slot/{m,v}/<layer>.<variable>` (flat names of our own, not TF's `.OPTIMIZER_SLOT` paths).  Kernels are stored in the Keras layouts (HWIO / HWOI), i.e.
byte-for-byte the master buffers.  The *container* is a self-describing native format (JSON index + raw
little-endian data), NOT TensorFlow's TensorBundle SSTable: no TF-written checkpoint ships with the reference
to pin that format against."""
from __future__ import annotations

import json
import os
import re

import numpy as np


def tf_variable_key(obj_name: str, layer_names: list, param: str) -> str:
    """'down3.gamma' of object 'generator' -> TF-style object-graph key."""
    layer, attr = param.rsplit('.', 1)
    k = layer_names.index(layer)
    inner = {'kernel': 0, 'bias': 0, 'gamma': 1, 'beta': 1, 'scale': 1, 'offset': 1, 'moving_mean': 1,
             'moving_variance': 1}[attr]
    return f"{obj_name}/layer_with_weights-{k}/layer_with_weights-{inner}/{attr}/.ATTRIBUTES/VARIABLE_VALUE"


GEN_LAYERS = [f'down{i}' for i in range(8)] + [f'up{i}' for i in range(7)] + ['last']
DISC_LAYERS = ['down0', 'down1', 'down2', 'conv', 'last']


class Checkpoint:
    """Named collection of networks ({obj_name: (arrays_getter, arrays_setter, layer_names)}) like
    tf.train.Checkpoint(generator=..., discriminator=..., generator_optimizer=...)."""

    def __init__(self, **objects):
        self.objects = objects       # name -> object exposing state_dict() / load_state_dict(dict)
        self.save_counter = 0

    def _gather(self):
        out = {}
        for name, obj in self.objects.items():
            for k, v in obj.state_dict().items():
                out[f"{name}/{k}"] = np.ascontiguousarray(v)
        out['save_counter/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(self.save_counter, np.int64)
        return out

    def write(self, prefix: str):
        arrays = self._gather()
        index, off = {}, 0
        with open(prefix + '.data-00000-of-00001', 'wb') as f:
            for k in sorted(arrays):
                a = arrays[k]
                b = a.tobytes()
                index[k] = {'dtype': str(a.dtype), 'shape': list(a.shape), 'offset': off, 'size': len(b)}
                f.write(b)
                off += len(b)
        with open(prefix + '.index', 'w') as f:
            json.dump({'format': 'gan_amd-bundle-v1', 'tensors': index}, f)
        return prefix

    def restore(self, prefix: str):
        """Like `.restore(...).expect_partial()` (pix2pix.py:411): keys missing on either side are ignored."""
        if prefix is None:
            raise ValueError("no checkpoint found")
        with open(prefix + '.index') as f:
            index = json.load(f)['tensors']
        data = np.memmap(prefix + '.data-00000-of-00001', dtype=np.uint8, mode='r')
        per_obj = {name: {} for name in self.objects}
        for k, meta in index.items():
            name, _, rest = k.partition('/')
            if name in per_obj:
                a = np.frombuffer(data[meta['offset']:meta['offset'] + meta['size']].tobytes(), dtype=meta['dtype'])
                per_obj[name][rest] = a.reshape(meta['shape'])
        for name, obj in self.objects.items():
            obj.load_state_dict(per_obj[name])
        if 'save_counter/.ATTRIBUTES/VARIABLE_VALUE' in index:
            m = index['save_counter/.ATTRIBUTES/VARIABLE_VALUE']
            self.save_counter = int(np.frombuffer(data[m['offset']:m['offset'] + m['size']].tobytes(), dtype=m['dtype'])[0])
        return self


class CheckpointManager:
    """tf.train.CheckpointManager(checkpoint, directory, max_to_keep) (pix2pix.py:420, cycle_gan.py:461)."""

    def __init__(self, checkpoint: Checkpoint, directory: str, max_to_keep: int = 1):
        self.ckpt, self.dir, self.keep = checkpoint, directory, max_to_keep
        # a restarted run keeps honouring max_to_keep: pick up the checkpoints the state file already lists
        self.paths = []
        state = os.path.join(directory, 'checkpoint')
        if os.path.exists(state):
            for name in re.findall(r'all_model_checkpoint_paths:\s*"([^"]+)"', open(state).read()):
                if os.path.exists(os.path.join(directory, name + '.index')):
                    self.paths.append(os.path.join(directory, name))
            for pth in self.paths:
                m = re.search(r'ckpt-(\d+)$', pth)
                if m:
                    self.ckpt.save_counter = max(self.ckpt.save_counter, int(m.group(1)))

    def save(self):
        os.makedirs(self.dir, exist_ok=True)
        self.ckpt.save_counter += 1
        prefix = os.path.join(self.dir, f"ckpt-{self.ckpt.save_counter}")
        self.ckpt.write(prefix)
        self.paths.append(prefix)
        while len(self.paths) > self.keep:
            old = self.paths.pop(0)
            for ext in ('.index', '.data-00000-of-00001'):
                if os.path.exists(old + ext):
                    os.remove(old + ext)
        with open(os.path.join(self.dir, 'checkpoint'), 'w') as f:
            f.write(f'model_checkpoint_path: "{os.path.basename(prefix)}"\n')
            for p in self.paths:
                f.write(f'all_model_checkpoint_paths: "{os.path.basename(p)}"\n')
        return prefix


def latest_checkpoint(directory: str):
    """tf.train.latest_checkpoint: read `<dir>/checkpoint`."""
    path = os.path.join(directory, 'checkpoint')
    if not os.path.exists(path):
        return None
    m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(path).read())
    return os.path.join(directory, m.group(1)) if m else None
