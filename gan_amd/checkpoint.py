"""Checkpointing with the reference's directory / object naming AND container (pix2pix.py:400-403,419-420,
cycle_gan.py:437-444,460-461): `<dir>/checkpoint` (text proto, `model_checkpoint_path: "ckpt-N"`), `ckpt-N.index`,
`ckpt-N.data-00000-of-00001` in TensorFlow's TensorBundle format (gan_amd/tfbundle.py: sorted string table of
BundleEntryProto + raw little-endian tensor data + the `_CHECKPOINTABLE_OBJECT_GRAPH` string tensor), keep-last-K.
Variable keys follow TF's object-graph naming: `generator/layer_with_weights-K[/layer_with_weights-J]/kernel/.ATTRIBUTES/
VARIABLE_VALUE`; optimizer hyper-parameters `generator_optimizer/{iter,beta_1,beta_2,decay,learning_rate}/...`; Adam slots
`<variable path>/.OPTIMIZER_SLOT/generator_optimizer/{m,v}/.ATTRIBUTES/VARIABLE_VALUE`; `save_counter/...`.  Kernels are
stored in the Keras layouts (HWIO / HWOI), byte-for-byte the master buffers.  Format parity with TensorFlow is unpinned
(tfbundle.py header); checkpoints written by round 1's JSON container are still readable."""
from __future__ import annotations

import json
import os
import re

import numpy as np

from . import tfbundle

# `layer_with_weights-K` numbering of the reference's Keras models (layers that own variables, in model order):
# Generator (base_gan.py:168-225): 8 downsample Sequentials (0..7), 7 upsample Sequentials (8..14), the Conv2DTranspose
# head (15).  Inside a Sequential: the convolution is `layer_with_weights-0`, its normalisation layer `-1`.
# Discriminator (base_gan.py:124-166): 3 downsample Sequentials (0..2), then PLAIN layers: Conv2D (3), its
# BatchNormalization / InstanceNormalization (4), the logits Conv2D (5).
_NORM_ATTRS = ('gamma', 'beta', 'scale', 'offset', 'moving_mean', 'moving_variance')


def tf_variable_key(obj_name: str, layer_names: list, param: str) -> str:
    """'down3.gamma' of object 'generator' -> TF object-graph checkpoint key."""
    layer, attr = param.rsplit('.', 1)
    sequential = layer.startswith(('down', 'up'))
    if sequential:
        k = layer_names.index(layer)
        path = f"layer_with_weights-{k}/layer_with_weights-{1 if attr in _NORM_ATTRS else 0}"
    elif layer == 'conv':                       # discriminator: Conv2D and its norm layer are two model-level layers
        path = f"layer_with_weights-{4 if attr in _NORM_ATTRS else 3}"
    else:                                       # 'last': generator head (15) / discriminator logits layer (5)
        path = f"layer_with_weights-{15 if len(layer_names) > 8 else 5}"
    return f"{obj_name}/{path}/{attr}/.ATTRIBUTES/VARIABLE_VALUE"


GEN_LAYERS = [f'down{i}' for i in range(8)] + [f'up{i}' for i in range(7)] + ['last']
DISC_LAYERS = ['down0', 'down1', 'down2', 'conv', 'last']


class Checkpoint:
    """tf.train.Checkpoint(generator=..., discriminator=..., generator_optimizer=..., ...): named objects exposing
    state_dict() / load_state_dict().  An object named `<model>_optimizer` is the optimizer of `<model>`: its Adam slots
    are stored under the model's variable paths, as TensorFlow does."""
    SAVE_COUNTER = 'save_counter/.ATTRIBUTES/VARIABLE_VALUE'

    def __init__(self, **objects):
        self.objects = objects
        self.save_counter = 0

    def _slot_key(self, opt_name, slot, param):
        model = opt_name[:-len('_optimizer')]
        var = tf_variable_key(model, self.objects[model].layers, param)
        return var.replace('/.ATTRIBUTES/', f'/.OPTIMIZER_SLOT/{opt_name}/{slot}/.ATTRIBUTES/')

    def _gather(self):
        out, names, slots = {}, {}, []
        for name, obj in self.objects.items():
            for k, v in obj.state_dict().items():
                if k.startswith('slot/'):
                    _, slot, param = k.split('/', 2)
                    key = self._slot_key(name, slot, param)
                    model = name[:-len('_optimizer')]
                    slots.append((name, slot, tf_variable_key(model, self.objects[model].layers, param), key))
                else:
                    key = f"{name}/{k}"
                out[key] = np.asarray(v)
                names[key] = key.split('/.ATTRIBUTES/')[0]
        out[self.SAVE_COUNTER] = np.array(self.save_counter, np.int64)
        names[self.SAVE_COUNTER] = 'save_counter'
        return out, names, slots

    def write(self, prefix: str):
        arrays, names, slots = self._gather()
        return tfbundle.write_bundle(prefix, arrays, tfbundle.object_graph(names, slots))

    def _read(self, prefix):
        with open(prefix + '.index', 'rb') as f:
            legacy = f.read(1) == b'{'
        if not legacy:
            return tfbundle.read_bundle(prefix)[0]
        with open(prefix + '.index') as f:              # round-1 container: JSON index + raw data
            index = json.load(f)['tensors']
        data = np.memmap(prefix + '.data-00000-of-00001', dtype=np.uint8, mode='r')
        out = {}
        for k, meta in index.items():
            a = np.frombuffer(data[meta['offset']:meta['offset'] + meta['size']].tobytes(), dtype=meta['dtype'])
            out[k] = a.reshape(meta['shape'])
        return out

    def restore(self, prefix: str):
        """Like `.restore(...).expect_partial()` (pix2pix.py:411): keys missing on either side are ignored."""
        if prefix is None:
            raise ValueError("no checkpoint found")
        arrays = self._read(prefix)
        per_obj = {name: {} for name in self.objects}
        for k, a in arrays.items():
            m = re.match(r'(.+)/\.OPTIMIZER_SLOT/([^/]+)/([^/]+)/\.ATTRIBUTES/VARIABLE_VALUE$', k)
            if m and m.group(2) in per_obj:              # slot of <optimizer>: hand it over under the optimizer's own naming
                model = m.group(2)[:-len('_optimizer')]
                if model in self.objects:
                    for param in self.objects[model].param_names():
                        if tf_variable_key(model, self.objects[model].layers, param).startswith(m.group(1) + '/'):
                            per_obj[m.group(2)][f'slot/{m.group(3)}/{param}'] = a
                            break
                continue
            name, _, rest = k.partition('/')
            if name in per_obj:
                per_obj[name][rest] = a
        for name, obj in self.objects.items():
            obj.load_state_dict(per_obj[name])
        if self.SAVE_COUNTER in arrays:
            self.save_counter = int(np.asarray(arrays[self.SAVE_COUNTER]).reshape(-1)[0])
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
