"""`GAN` base class with the reference's surface (base_gan.py:21-292) on top of the MI355X kernels.

`Generator(...)` / `Discriminator(...)` return callable model objects (`model(x, training=True)`), like the
Keras models of the reference; their weights live in `gan_amd.nets.ParamSet` buffers shared with the fused
train-step objects.  As in the reference, `training=True` is what every call site uses (batch statistics,
dropout on — pix2pix.py:200-203,228); the flag is accepted and ignored the same way Keras ignores it for
layers without inference-time state in this graph."""
from __future__ import annotations

import ctypes as C
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import _lib as L
from . import data as D
from .checkpoint import DISC_LAYERS, GEN_LAYERS, tf_variable_key
from .nets import Ctx, DiscriminatorNet, GeneratorNet, workspace_mb_for


def _to_dev(x, ctx):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return x.to(device=ctx.device, dtype=torch.float32).contiguous()


class _Model:
    """Common: state_dict()/load_state_dict() with TF object-graph style keys (checkpoint.py)."""
    layers = GEN_LAYERS
    obj_name = 'model'

    def state_dict(self):
        P = self.net.params.to_numpy()
        return {tf_variable_key(self.obj_name, self.layers, k).split('/', 1)[1]: v for k, v in P.items()}

    def load_state_dict(self, sd):
        P = self.net.params.to_numpy()
        for k in list(P):
            key = tf_variable_key(self.obj_name, self.layers, k).split('/', 1)[1]
            if key in sd:
                P[k] = sd[key].astype(np.float32).reshape(P[k].shape)
        self.net.params.load_numpy(P)

    def param_names(self):
        return list(self.net.params.entries) + list(self.net.params.state)

    @property
    def trainable_variables(self):
        return [self.net.params.tensor(n) for n in self.net.params.names]

    def count_params(self):
        return self.net.params.trainable_count()


class GeneratorModel(_Model):
    layers = GEN_LAYERS

    def __init__(self, net: GeneratorNet, obj_name='generator'):
        self.net, self.obj_name, self._calls = net, obj_name, {}

    def __call__(self, x, training=True):
        ctx = self.net.ctx
        x = _to_dev(x, ctx)
        B, S = x.shape[0], x.shape[1]
        key = (B, S)
        if key not in self._calls:
            self._calls[key] = self.net.new_call(B, S, dropout=True, stream_id=40)     # (ids 0..5 / 16..21: the train / validation steps' generator calls)
        call = self._calls[key]
        call.set_input(x)
        call.forward()
        return call.output_f32()


class DiscriminatorModel(_Model):
    layers = DISC_LAYERS

    def __init__(self, net: DiscriminatorNet, obj_name='discriminator'):
        self.net, self.obj_name, self._calls = net, obj_name, {}

    def __call__(self, inputs, training=True):
        ctx = self.net.ctx
        xs = inputs if isinstance(inputs, (list, tuple)) else [inputs]
        xs = [_to_dev(x, ctx) for x in xs]
        B, S, Cc = xs[0].shape[0], xs[0].shape[1], self.net.channels
        key = (B, S)
        if key not in self._calls:
            self._calls[key] = self.net.new_call(B, S, calls=1)
        call = self._calls[key]
        for i, x in enumerate(xs):      # concatenate([inp, tar]) realised as channel slices (base_gan.py:139)
            v = call.xin.view(i * Cc, Cc)
            L.check(ctx.lib.gan_pack(ctx.dt, x.data_ptr(), C.byref(v), ctx.stream()), "pack")
        call.forward()
        return call.logits.t.clone()


class AdamConfig:
    """`GAN.optimizer(...)` result: hyper-parameters; the state (m, v, step) lives beside the weights."""

    def __init__(self, learning_rate, beta_1, beta_2):
        self.learning_rate, self.beta_1, self.beta_2 = learning_rate, beta_1, beta_2
        self.params = None

    def bind(self, paramset):
        self.params = paramset
        return self

    def state_dict(self):
        if self.params is None:
            return {}
        ps = self.params
        out = {'iter/.ATTRIBUTES/VARIABLE_VALUE': ps.step.cpu().numpy().astype(np.int64).reshape(()),       # scalars, as Keras stores them
               'learning_rate/.ATTRIBUTES/VARIABLE_VALUE': np.float32(self.learning_rate),
               'beta_1/.ATTRIBUTES/VARIABLE_VALUE': np.float32(self.beta_1),
               'beta_2/.ATTRIBUTES/VARIABLE_VALUE': np.float32(self.beta_2),
               'decay/.ATTRIBUTES/VARIABLE_VALUE': np.float32(0.0)}
        for slot in ('m', 'v'):
            for k, a in ps.to_numpy(slot).items():
                out[f'slot/{slot}/{k}'] = a
        return out

    def load_state_dict(self, sd):
        if self.params is None:
            return
        ps = self.params
        if 'iter/.ATTRIBUTES/VARIABLE_VALUE' in sd:
            ps.step.copy_(torch.from_numpy(np.asarray(sd['iter/.ATTRIBUTES/VARIABLE_VALUE'], np.int32).reshape(1)))
        for slot in ('m', 'v'):
            for name in ps.entries:
                k = f'slot/{slot}/{name}'
                if k in sd:
                    ps.tensor(name, slot).copy_(torch.from_numpy(sd[k].astype(np.float32)).view(ps.entries[name][1]))


class GAN(ABC):
    def __init__(self, config):
        self.config = config
        self.ctx = Ctx(config.get('device', 'cuda:0'), config.get('dtype', 'bf16'),
                       workspace_mb=workspace_mb_for(int(config.get('batch_size', 1)), int(config.get('img_size', 256)),
                                                     int(config.get('channels', 1))))
        self.loss_obj = self.loss_object()

    # ---- image helpers (base_gan.py:26-61) -------------------------------------------------------
    def load(self, image_file: str, resize: bool = False):
        image = D.load(image_file, int(self.config['channels']))
        if resize:
            image = self.resize(image, self.config['img_size'], self.config['img_size'])
        return image

    def resize(self, image, height: int, width: int):
        return D.resize_nearest(image, height, width)

    def normalize(self, image):
        return D.normalize(image)

    # ---- model builders (base_gan.py:124-225) ----------------------------------------------------
    def Discriminator(self, norm_type: str = 'batchnorm', target: bool = True, seed: int = 1, name='discriminator'):
        return DiscriminatorModel(DiscriminatorNet(self.ctx, int(self.config['channels']), target, norm_type.lower(), seed), name)

    def Generator(self, norm_type='batchnorm', shape: tuple = (None, None, None), seed: int = 0, name='generator'):
        channels = shape[2] if shape[2] is not None else int(self.config['channels'])
        return GeneratorModel(GeneratorNet(self.ctx, channels, norm_type.lower(), seed), name)

    # ---- losses / optimiser (base_gan.py:227-252) ------------------------------------------------
    def loss_object(self):
        """BinaryCrossentropy(from_logits=True): callable(target_like, logits) -> scalar tensor."""
        ctx = self.ctx

        def bce(y_true, logits):
            t = float(y_true) if np.isscalar(y_true) else float(torch.as_tensor(y_true).flatten()[0])
            x = logits.to(device=ctx.device, dtype=torch.float32).contiguous()
            out = torch.zeros(1, dtype=torch.float32, device=ctx.device)
            ws = torch.empty(1024, dtype=torch.float32, device=ctx.device)
            L.check(ctx.lib.gan_bce_logits(x.data_ptr(), x.numel(), t, 1.0, 0, out.data_ptr(), 0.0, ctx.dt, None, 8,
                                           ws.data_ptr(), None, ctx.stream()), "bce_logits")
            return out[0]
        return bce

    def discriminator_loss(self, real, generated, factor: float = 1.0):
        real_loss = self.loss_obj(1.0, real)
        generated_loss = self.loss_obj(0.0, generated)
        return (real_loss + generated_loss) * factor

    def optimizer(self, learning_rate: float = 2e-4, beta_1: float = 0.5, beta_2: float = 0.999):
        return AdamConfig(learning_rate, beta_1, beta_2)

    # ---- abstract surface, as in the reference (base_gan.py:254-292) ----------------------------
    @abstractmethod
    def image_pipeline(self, *args, **kwargs): ...

    @abstractmethod
    def generator_loss(self, *args, **kwargs): ...

    @abstractmethod
    def generate_images(self, *args, **kwargs): ...

    @abstractmethod
    def train_step(self, *args, **kwargs): ...

    @abstractmethod
    def fit(self, *args, **kwargs): ...

    @abstractmethod
    def predict(self, *args, **kwargs): ...
