"""Device-side U-Net generator and PatchGAN discriminator built from the C-ABI ops (include/gan_amd.h).

Follows the topology of the reference's builders (base_gan.py:63-225): `GAN.downsample`, `GAN.upsample`,
`GAN.Generator`, `GAN.Discriminator`.  PyTorch is used only for device memory and streams; every
arithmetic op is a hand-written HIP kernel reached through ctypes.  Each "call" object owns the
activation / gradient buffers of ONE forward invocation of a network and a pre-built list of C calls, so
running it is a plain loop of enqueue calls (capturable in a hipGraph).

Skip-concat (base_gan.py:219-221) is zero-copy: the up-layer's activation and the down-layer's
activation are written by their producers straight into channel slices of one NHWC "cat" buffer.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

G_DOWN = [64, 128, 256, 512, 512, 512, 512, 512]     # base_gan.py:179-188
G_UP = [512, 512, 512, 512, 256, 128, 64]            # base_gan.py:190-198
LEAKY_ALPHA = 0.3                                    # Keras LeakyReLU() default (base_gan.py:87,155)
BN_EPS, IN_EPS, BN_MOMENTUM = 1e-3, 1e-5, 0.99       # Keras BatchNormalization defaults; utils.py:9


FUSE_BWD = True              # dgrad epilogues start the backward of the layer below (GanBwdFuse)
NORM_SYNC_SLOTS = 2048       # sync areas (2,176 B each) for normalisation launches that carry their own finalize: Ctx.new_sync()
STATS_RESERVE = 8 << 20      # room kept behind a conv's split-K slabs for its fused statistics partials


def pad8(c):
    return (c + 7) // 8 * 8


def workspace_mb_for(batch, size, channels=1, calls=2):
    """Lane workspace (MiB) that covers the largest scratch user of a step at this shape: the thin-N kernels'
    Z[pixel][c][tap] fp32 buffer of the discriminator's input gradient (2*batch... images, 2*channels outputs)."""
    need = calls * batch * (size // 2) ** 2 * max(1, 2 * channels) * 64 + (16 << 20)
    return max(256, (need >> 20) + 1)


class LaneStream(torch.cuda.Stream):
    """A side stream of the step schedule that remembers whether it has been forked into a hipGraph capture and not
    joined back yet: ending a capture with an unjoined lane took the process down inside capture_end (round 1,
    gpurun_out/crash.log) instead of raising.  `lane.wait_stream(x)` = fork, `Ctx.join(waiter, lane)` = join;
    `Ctx.assert_lanes_joined()` runs before every capture ends."""
    open_in_capture = False

    def wait_stream(self, stream, _join=False):
        """Inside a capture two patterns end in a host segfault inside hipStreamEndCapture on ROCm 7.2 (tools/capture_fork_probe.py,
        profiles/r04_capture_fork_probe.txt) and are refused here instead: FORKING a lane from a lane that is itself forked and still
        open (every side lane must be forked from the origin stream), and edges in BOTH directions between two open lanes.  One-way
        cross-waits between lanes that are already forked (the CycleGAN chains' event waits) are fine."""
        src_open = isinstance(stream, LaneStream) and stream.open_in_capture
        if src_open and not _join:        # (Ctx.join: a lane that collects another lane's work is the schedule's own business)
            if not self.open_in_capture:
                raise L.GanAmdError("step schedule bug: a lane forked from a forked lane (every side lane must be forked from the origin stream: "
                                    "hipStreamEndCapture crashes on nested forks, DESIGN.md section 5)")
            if self in stream.__dict__.get('_waits_on', ()):
                raise L.GanAmdError("step schedule bug: two forked lanes wait on each other in both directions (hipStreamEndCapture crashes, "
                                    "DESIGN.md section 5)")
            self.__dict__.setdefault('_waits_on', set()).add(stream)
        if torch.cuda.is_current_stream_capturing() or getattr(stream, 'open_in_capture', False):
            self.open_in_capture = True
        return super().wait_stream(stream)


class Ctx:
    """Device, dtype, library handle and the shared split-K / reduction workspace."""

    def __init__(self, device='cuda:0', dtype='bf16', workspace_mb=256, lanes=True, loss_scale=2.0 ** 15):
        """lanes: a captured step forks independent chains onto side streams (the schedule of gan_amd/steps.py); False puts
        every launch on the current stream (profiling, per-kernel timing).  loss_scale: initial dynamic loss scale (fp16)."""
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.GanAmdError("gan_amd needs an MI355X (no CPU fallback)")
        self.device = torch.device(device)
        if dtype not in ('f32', 'bf16', 'f16'):
            raise ValueError(f"dtype {dtype!r}: expected 'f32', 'bf16' or 'f16'")
        self.dtype = dtype
        self.dt = {'f32': L.F32, 'bf16': L.BF16, 'f16': L.F16}[dtype]
        self.tdtype = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}[dtype]
        # fp16 path: dynamic loss scale on the device {scale, 1/scale, finite steps in a row, non-finite flag}
        # (include/gan_amd.h, gan_grads_check / gan_loss_scale_update); None = no scaling
        self.ls = None
        if dtype == 'f16':
            s0 = float(loss_scale)
            self.ls = torch.tensor([s0, 1.0 / s0, 0.0, 0.0], dtype=torch.float32, device=self.device)
        self.ls_ptr = self.ls.data_ptr() if self.ls is not None else None
        self.ls_growth_interval, self.ls_max = 2000, 2.0 ** 24        # Keras LossScaleOptimizer defaults
        # "Lanes": independent launch chains that may overlap on the GPU (separate HIP streams, also inside a
        # captured graph).  Lane 0 = the current stream; lane 1 = its wgrad side stream (wgrad kernels only
        # feed Adam, so they run beside the dgrad/norm chain); lane 2 / 3 = a second chain (discriminator
        # parameter-gradient pass beside the generator backward) and its wgrad side stream.  Each lane has its
        # own workspace (split-K slabs / reduction partials).
        # (workspaces 4..7: parameter-gradient passes of the discriminators in the two-chain CycleGAN schedule)
        self.ws_lanes = [torch.empty(workspace_mb << 20, dtype=torch.uint8, device=self.device) for _ in range(8)]
        self.ws = self.ws_lanes[0]
        self.ws_ptr, self.ws_bytes = self.ws.data_ptr(), self.ws.numel()
        self.side = [LaneStream(device=self.device) for _ in range(4)]     # lanes 1..4 (4: early optimiser step)
        self.lanes = bool(lanes)
        # layer stacks (gan_conv_stack_*): one timeout flag for all of them - a grid barrier that could not complete sets it
        self.stack_err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.use_stacks = bool(L.get_option('conv.stack'))
        # sync areas of the normalisation launches that carry their own finalize (GanNormDesc.sync, include/gan_amd.h): one per
        # layer invocation and direction, handed out when the op lists are built, zero now and self-cleaning afterwards
        self.fin_in_apply = bool(L.get_option('norm.fin_in_apply'))
        self._sync_words = self.lib.gan_norm_sync_bytes() // 4
        self._sync_err = self.lib.gan_norm_sync_error_offset() // 4
        self.norm_sync = torch.zeros(NORM_SYNC_SLOTS, self._sync_words, dtype=torch.int32, device=self.device)
        self._sync_next = 0

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def new_sync(self):
        """Device pointer of an unused sync area, or None (option off / pool used up: the layer keeps its finalize launch)."""
        if not self.fin_in_apply or self._sync_next >= NORM_SYNC_SLOTS:
            return None
        self._sync_next += 1
        return self.norm_sync.data_ptr() + (self._sync_next - 1) * self._sync_words * 4

    def join(self, waiter, lane):
        """`waiter` waits for everything queued on `lane` (the join of a fork made with lane.wait_stream)."""
        if isinstance(waiter, LaneStream):
            waiter.wait_stream(lane, _join=True)
        else:
            waiter.wait_stream(lane)
        if isinstance(lane, LaneStream) and not getattr(waiter, 'open_in_capture', False):
            lane.open_in_capture = False          # joined into the capturing (or an already joined) stream
            lane.__dict__.pop('_waits_on', None)

    def assert_lanes_joined(self):
        bad = [i + 1 for i, s in enumerate(self.side) if s.open_in_capture]
        if bad:      # (the flags stay set: capture_graph() joins exactly these lanes before the capture ends)
            raise L.GanAmdError(f"step schedule bug: lane(s) {bad} were forked during graph capture and never joined")

    def join_open_lanes(self):
        """Join every lane that is still forked into the running capture back into the capturing stream (best effort: a capture
        that a HIP error has already invalidated refuses further calls)."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.side:
            if s.open_in_capture:
                try:
                    cur.wait_stream(s)
                except Exception:
                    pass
                s.open_in_capture = False

    def capture_graph(self, fn, error_mode="thread_local"):
        """fn() captured into a hipGraph.  ROCm 7.2 ends a capture that still has a forked, unjoined lane with a crash inside
        hipStreamEndCapture (or hipErrorStreamCaptureUnjoined) instead of a clean error, so NO path may leave the capture block
        with open lanes: an exception raised inside fn() - a kernel entry point's error code, the lane guard itself - is caught
        here, the open lanes are joined, the capture is ended and the exception comes back as GanAmdError."""
        gr = torch.cuda.CUDAGraph()
        err = None
        try:
            with torch.cuda.graph(gr, capture_error_mode=error_mode):
                try:
                    fn()
                    self.assert_lanes_joined()
                except BaseException as e:      # noqa: BLE001 - re-raised below, after the capture has been closed safely
                    err = e
                    self.join_open_lanes()
        except Exception as e2:                 # capture_end itself failed (the capture had been invalidated)
            if err is None:
                raise
            raise L.GanAmdError(f"graph capture failed: {err!r}; ending the capture then reported: {e2!r}") from err
        if err is not None:
            if isinstance(err, L.GanAmdError):
                raise err
            if not isinstance(err, Exception):
                raise err                       # KeyboardInterrupt / SystemExit pass through unchanged
            raise L.GanAmdError(f"graph capture failed: {err!r}") from err
        return gr

    def lane_stream(self, lane):
        return torch.cuda.current_stream(self.device) if lane == 0 else self.side[lane - 1]

    def assert_no_stack_timeout(self):
        """Synchronising check: a persistent layer-stack kernel whose grid never became resident gives up after ~2 s and raises this
        flag instead of hanging the GPU (its results are then garbage)."""
        if int(self.stack_err.item()):
            raise L.GanAmdError("a layer-stack kernel timed out at a grid barrier (more than two of them running at once?)")
        if self._sync_next and bool(self.norm_sync[:self._sync_next, self._sync_err].any().item()):
            raise L.GanAmdError("a normalisation launch timed out waiting for its own finalize workgroups")

    def run_on(self, ops, stream):
        """All ops in program order on one explicit stream."""
        st = stream.cuda_stream
        for op in ops:
            rc = op[0](*op[1], st)
            if rc:
                L.check(rc, op[2])

    def run(self, ops, lane=0):
        """All ops in program order on lane `lane` (0 = the current stream)."""
        st = self.lane_stream(lane).cuda_stream
        for op in ops:
            rc = op[0](*op[1], st)
            if rc:
                L.check(rc, op[2])


class Buf:
    """Zero-initialised NHWC device buffer; view() makes a GanTensor channel/batch slice."""

    parent = None      # the wider buffer this one is a batch slice of (Buf.slice_of)

    def __init__(self, ctx, n, h, w, c, tdtype=None):
        self.n, self.h, self.w, self.c = n, h, w, c
        self.t = torch.zeros((n, h, w, c), dtype=tdtype or ctx.tdtype, device=ctx.device)

    @classmethod
    def slice_of(cls, big, n0, n):
        """Samples [n0, n0 + n) of `big` as a buffer of its own (same storage; .parent = big)."""
        b = cls.__new__(cls)
        b.n, b.h, b.w, b.c = n, big.h, big.w, big.c
        b.t = big.t[n0:n0 + n]
        b.parent = big
        return b

    def wide(self):
        return self.parent if self.parent is not None else self

    def view(self, c0=0, c=None, n0=0, n=None):
        c = self.c - c0 if c is None else c
        n = self.n - n0 if n is None else n
        off = (n0 * self.h * self.w * self.c + c0) * self.t.element_size()
        return L.GanTensor(self.t.data_ptr() + off, n, self.h, self.w, c, self.c)


# --------------------------------------------------------------------------------------------------
class ParamSet:
    """Flat fp32 master / grad / Adam-m / Adam-v buffers of one network (Keras layouts: HWIO kernels for
    Conv2D, (kh,kw,cout,cin) for Conv2DTranspose) plus the typed NK copies the GEMM kernels consume."""

    ALIGN = 64

    def __init__(self, ctx, spec):
        """spec: list of (name, shape, is_trainable)"""
        self.ctx = ctx
        self.entries = {}
        off = 0
        # flat layout: every conv kernel first, then the vectors (norm scales/offsets, biases), so that the fused
        # Adam + NK-prep launch covers [0, vec_start) and one plain Adam launch the rest
        train = [(n_, s_) for n_, s_, t_ in spec if t_]
        self.names = [n_ for n_, _ in train]              # the reference's variable order (trainable_variables)
        for name, shape in [e for e in train if e[0].endswith('.kernel')] + [e for e in train if not e[0].endswith('.kernel')]:
            if not name.endswith('.kernel') and 'vec_start' not in self.__dict__:
                self.vec_start = off
            n = int(np.prod(shape))
            self.entries[name] = (off, tuple(shape))
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.__dict__.setdefault('vec_start', off)
        dev = ctx.device
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.state = {name: torch.zeros(shape, dtype=torch.float32, device=dev) if 'mean' in name
                      else torch.ones(shape, dtype=torch.float32, device=dev)
                      for name, shape, trainable in spec if not trainable}
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=dev)
        self.nat, self.tr = {}, {}
        for name, (o, shape) in self.entries.items():
            if name.endswith('.kernel'):
                A, B = shape[2], shape[3]
                self.nat[name] = torch.zeros((16, A, pad8(B)), dtype=ctx.tdtype, device=dev)
                self.tr[name] = torch.zeros((16, B, pad8(A)), dtype=ctx.tdtype, device=dev)
        # one launch refreshes every NK copy of the network (device table of GanPrepEntry)
        self._prep_table, self._prep_args = self._kernel_table(list(self.nat))
        self._prep_ops = [(ctx.lib.gan_weights_prepare_multi, self._prep_args, "weights_prepare_multi")]
        self._segments = None

    def _kernel_table(self, names):
        """Device table of GanPrepEntry for the kernels `names` -> (table tensor, (ptr, n, tiles, dtype))."""
        ents, tiles = [], 0
        for name in names:
            o, shape = self.entries[name]
            A, B = shape[2], shape[3]
            tb = (pad8(B) + 63) // 64
            ents.append(L.GanPrepEntry(self.master.data_ptr() + 4 * o, self.nat[name].data_ptr(), self.tr[name].data_ptr(), A, B, tiles, tb))
            tiles += 16 * ((pad8(A) + 63) // 64) * tb
        arr = (L.GanPrepEntry * len(ents))(*ents)
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.ctx.device)
        return table, (table.data_ptr(), len(ents), tiles, self.ctx.dt)

    def split_kernels_at(self, *cut_names):
        """Update segments of the kernel tensors, cut before each of `cut_names` (own tables), so that a segment can be
        updated as soon as ITS gradients are complete."""
        cache = self.__dict__.setdefault('_segment_tables', {})   # kept for good: captured graphs of several step objects (batch
        if cut_names not in cache:                                # sizes) hold the device pointers of these tables
            names = list(self.nat)
            ks = [0] + [names.index(n) for n in cut_names] + [len(names)]
            cache[cut_names] = [self._kernel_table(names[a:b]) for a, b in zip(ks[:-1], ks[1:])]
        self._segments = cache[cut_names]

    def adam_begin_ops(self, lr, b1, b2):
        return [(self.ctx.lib.gan_adam_begin, (self.step.data_ptr(), self.lr_t.data_ptr(), lr, b1, b2, self.ctx.ls_ptr), "adam_begin")]

    def adam_segment_ops(self, seg, b1, b2, eps=1e-7, grad_scale=1.0, vectors=False, kernels=True, wire_ptr=None):
        """Fused Adam + NK refresh of kernel segment `seg` (split_kernels_at), optionally followed by the vectors'
        plain Adam.  gan_adam_begin must already have run this step.  wire_ptr: read the gradient from the bf16 wire
        buffer of the data-parallel exchange (same element offsets as the flat fp32 gradient buffer) instead."""
        lib = self.ctx.lib
        gw = 1 if wire_ptr else 0
        ptrs = (self.master.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), wire_ptr or self.grad.data_ptr())
        ops = []
        if kernels:
            ops.append((lib.gan_adam_prepare_multi, self._segments[seg][1] + ptrs + (self.lr_t.data_ptr(), b1, b2, eps, grad_scale, self.ctx.ls_ptr, gw),
                        "adam_prepare_multi"))
        nvec = self.total - self.vec_start
        if vectors and nvec > 0:
            vptrs = tuple(p_ + 4 * self.vec_start for p_ in ptrs[:3]) + (ptrs[3] + (2 if gw else 4) * self.vec_start,)
            ops.append((lib.gan_adam_tf, vptrs + (nvec, self.lr_t.data_ptr(), b1, b2, eps, grad_scale, self.ctx.ls_ptr, gw), "adam_tf"))
        return ops

    def adam_fuse_desc(self, name, b1, b2, eps=1e-7):
        """GanAdamFuse for kernel `name`: its wgrad launch applies this step's Adam update and refreshes its NK copies."""
        o = 4 * self.entries[name][0]
        return L.GanAdamFuse(self.master.data_ptr() + o, self.m.data_ptr() + o, self.v.data_ptr() + o, self.nat[name].data_ptr(),
                             self.tr[name].data_ptr(), self.lr_t.data_ptr(), b1, b2, eps)

    def adam_rest_ops(self, fused_names, b1, b2, eps=1e-7):
        """Adam of everything the wgrad launches did NOT update themselves: the other kernels (own table) and the vectors.
        gan_adam_begin must already have run this step (before the first fused wgrad)."""
        key = frozenset(fused_names)
        cache = self.__dict__.setdefault('_rest_tables', {})      # one device table per fused set, kept for good: captured graphs
        if key not in cache:                                      # of several step objects (batch sizes) hold their pointers
            names = [n_ for n_ in self.nat if n_ not in key]
            cache[key] = self._kernel_table(names) if names else None
        rest = cache[key]
        lib = self.ctx.lib
        ptrs = (self.master.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.grad.data_ptr())
        ops = []
        if rest is not None:
            ops.append((lib.gan_adam_prepare_multi, rest[1] + ptrs + (self.lr_t.data_ptr(), b1, b2, eps, 1.0, self.ctx.ls_ptr, 0), "adam_prepare_multi"))
        nvec = self.total - self.vec_start
        if nvec > 0:
            vptrs = tuple(p_ + 4 * self.vec_start for p_ in ptrs)
            ops.append((lib.gan_adam_tf, vptrs + (nvec, self.lr_t.data_ptr(), b1, b2, eps, 1.0, self.ctx.ls_ptr, 0), "adam_tf"))
        return ops

    def ptr(self, name, which='master'):
        return getattr(self, which).data_ptr() + 4 * self.entries[name][0]

    def tensor(self, name, which='master'):
        o, shape = self.entries[name]
        return getattr(self, which)[o:o + int(np.prod(shape))].view(shape)

    def grads_check_ops(self):
        """fp16 path: raise the loss-scale state's non-finite flag if this network's gradients hold an inf/nan."""
        return [(self.ctx.lib.gan_grads_check, (self.grad.data_ptr(), self.total, self.ctx.ls_ptr), "grads_check")]

    def trainable_count(self):
        return int(sum(np.prod(s) for _, s in self.entries.values()))

    def load_numpy(self, P):
        for name, (o, shape) in self.entries.items():
            self.tensor(name).copy_(torch.from_numpy(np.ascontiguousarray(P[name], dtype=np.float32)).view(shape))
        for name, t in self.state.items():
            if name in P:
                t.copy_(torch.from_numpy(np.asarray(P[name], dtype=np.float32)))
        self.prepare()

    def to_numpy(self, which='master'):
        out = {name: self.tensor(name, which).detach().cpu().numpy().copy() for name in self.entries}
        if which == 'master':
            out.update({k: v.cpu().numpy().copy() for k, v in self.state.items()})
        return out

    def prepare(self):
        """(Re)build the typed NK weight copies from the fp32 master."""
        self.ctx.run(self._prep_ops)

    def adam(self, lr, b1, b2, eps=1e-7, grad_scale=1.0, stream=None, wire_ptr=None):
        """Keras Adam (base_gan.py:247-252): the kernels in one launch fused with the refresh of their NK copies, the
        vectors (norm parameters, biases) in a second, small one.  `stream`: that lane instead of the current stream.
        wire_ptr: as adam_segment_ops."""
        lib = self.ctx.lib
        table_ptr, n_ents, tiles, dt = self._prep_args
        gw = 1 if wire_ptr else 0
        ptrs = (self.master.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), wire_ptr or self.grad.data_ptr())
        ops = [(lib.gan_adam_begin, (self.step.data_ptr(), self.lr_t.data_ptr(), lr, b1, b2, self.ctx.ls_ptr), "adam_begin"),
               (lib.gan_adam_prepare_multi, (table_ptr, n_ents, tiles, dt) + ptrs + (self.lr_t.data_ptr(), b1, b2, eps, grad_scale, self.ctx.ls_ptr, gw),
                "adam_prepare_multi")]
        nvec = self.total - self.vec_start
        if nvec > 0:
            vptrs = tuple(p_ + 4 * self.vec_start for p_ in ptrs[:3]) + (ptrs[3] + (2 if gw else 4) * self.vec_start,)
            ops.append((lib.gan_adam_tf, vptrs + (nvec, self.lr_t.data_ptr(), b1, b2, eps, grad_scale, self.ctx.ls_ptr, gw), "adam_tf"))
        if stream is not None:
            self.ctx.run_on(ops, stream)
        else:
            self.ctx.run(ops)


def _norm_spec(spec, name, c, norm):
    if norm == 'batchnorm':
        spec += [(name + '.gamma', (c,), True), (name + '.beta', (c,), True),
                 (name + '.moving_mean', (c,), False), (name + '.moving_variance', (c,), False)]
    elif norm == 'instancenorm':
        spec += [(name + '.scale', (c,), True), (name + '.offset', (c,), True)]


def generator_spec(channels, norm):
    spec = []
    cin = channels
    for i, co in enumerate(G_DOWN):
        spec.append((f'down{i}.kernel', (4, 4, cin, co), True))
        if i > 0:
            _norm_spec(spec, f'down{i}', co, norm)
        cin = co
    for i, co in enumerate(G_UP):
        spec.append((f'up{i}.kernel', (4, 4, co, cin), True))
        _norm_spec(spec, f'up{i}', co, norm)
        cin = co + G_DOWN[6 - i]
    spec += [('last.kernel', (4, 4, channels, cin), True), ('last.bias', (channels,), True)]
    return spec


def discriminator_spec(channels, target, norm):
    spec = []
    cin = channels * (2 if target else 1)
    for i, co in enumerate([64, 128, 256]):
        spec.append((f'down{i}.kernel', (4, 4, cin, co), True))
        if i > 0:
            _norm_spec(spec, f'down{i}', co, norm)
        cin = co
    spec.append(('conv.kernel', (4, 4, 256, 512), True))
    _norm_spec(spec, 'conv', 512, norm)
    spec += [('last.kernel', (4, 4, 512, 1), True), ('last.bias', (1,), True)]
    return spec


def init_params_numpy(spec, seed):
    """N(0, 0.02) kernels (base_gan.py:74,103,132,200), gamma=1/beta=0, IN scale N(1,0.02) (utils.py:14-24)."""
    rng = np.random.default_rng(seed)
    P = {}
    for name, shape, trainable in spec:
        if name.endswith('.kernel'):
            P[name] = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith('.gamma') or name.endswith('.moving_variance'):
            P[name] = np.ones(shape, np.float32)
        elif name.endswith('.scale'):
            P[name] = (1.0 + 0.02 * rng.standard_normal(shape)).astype(np.float32)
        else:
            P[name] = np.zeros(shape, np.float32)
    return P


# --------------------------------------------------------------------------------------------------
class _Builder:
    """Helpers that turn layer descriptions into (fn, args, label) C calls."""

    def __init__(self, ctx, params, norm, lane=0):
        self.ctx, self.P, self.norm = ctx, params, norm
        self.ws_ptr, self.ws_bytes = ctx.ws_lanes[lane].data_ptr(), ctx.ws_lanes[lane].numel()
        self.ws_side_ptr = ctx.ws_lanes[lane + 1].data_ptr()
        self.lib = ctx.lib
        self.wgrad_concurrent = False     # planner hint of the wgrad descriptors built here: the launch shares the chip with other lanes
        self.keep = []       # ctypes structs must outlive the op list

    def _desc(self, d):
        self.keep.append(d)
        return C.byref(d)

    def fuse_spec(self, ref, add=None, name=None, groups=0, mean_ptr=None, rstd_ptr=None, act='lrelu', mask_ptr=None, mask_pitch=0,
                  cols=None):
        """Description of the layer-below backward a dgrad launch should start in its epilogue (GanBwdFuse): `name` = its
        normalisation layer (None: activation only, `ref` = saved activation)."""
        z = L.GanTensor(None, 0, 0, 0, 0, 0)
        gk, bk = self.norm_names()
        f = L.GanBwdFuse(ref, add if add is not None else z, mean_ptr, rstd_ptr,
                         self.P.ptr(name + gk) if name else None, self.P.ptr(name + bk) if name else None,
                         mask_ptr, mask_pitch, L.ACTS[act], LEAKY_ALPHA, cols if cols is not None else ref.c)
        return (f, groups if name else 0)

    def fwd_norm_fuse(self, name, a, groups, mean, rstd, act, mask_ptr, update_moving=True):
        """GanNormFuse for the convolution that produces layer `name`'s pre-normalisation output: when the launch is a small
        split-K one, its slab-reduce kernel also takes the statistics and writes the activation `a` (self.last_full)."""
        gk, bk = self.norm_names()
        eps = BN_EPS if self.norm == 'batchnorm' else IN_EPS
        mm = mv = None
        if self.norm == 'batchnorm' and update_moving:
            mm = self.P.state[name + '.moving_mean'].data_ptr()
            mv = self.P.state[name + '.moving_variance'].data_ptr()
        return L.GanNormFuse(a, self.P.ptr(name + gk), self.P.ptr(name + bk), mean.data_ptr(), rstd.data_ptr(), mm, mv, eps, BN_MOMENTUM,
                             mask_ptr, L.ACTS[act], LEAKY_ALPHA, None, None, 0)

    def bwd_norm_fuse(self, name, dy, want_param_grads, accumulate):
        """GanNormFuse for the dgrad launch whose bwd_fuse names layer `name`: dy of that layer straight from the slab reduce."""
        gk, bk = self.norm_names()
        return L.GanNormFuse(dy, None, None, None, None, None, None, 0.0, 0.0, None, 0, LEAKY_ALPHA,
                             self.P.ptr(name + gk, 'grad') if want_param_grads else None,
                             self.P.ptr(name + bk, 'grad') if want_param_grads else None, int(accumulate))

    def conv(self, op, x, y, w, w_rows, stride=2, bias=None, act=None, y_f32=0, k_real=None, stats_groups=0, bwd_fuse=None, norm_fuse=None):
        """stats_groups > 0: ask the GEMM epilogue to also emit normalisation-statistics partials; the number of
        chunks it will write (0 = not fusable for this shape) is left in self.last_stats_chunks.
        bwd_fuse = fuse_spec(...): ask a dgrad launch to start the backward of the layer below in its epilogue;
        self.last_bwd_fused > 0 (the chunk count for gan_norm_act_bwd_fused) if this launch shape carries it."""
        if bwd_fuse is not None and not FUSE_BWD:
            bwd_fuse = None
        if bwd_fuse is not None:
            self.keep.append(bwd_fuse[0])
            stats_groups = bwd_fuse[1]
        if norm_fuse is not None:
            self.keep.append(norm_fuse)
        d = L.GanConvDesc(self.ctx.dt, stride, x, y, w, w_rows, bias, L.ACTS[act], LEAKY_ALPHA, y_f32,
                          self.ws_ptr, self.ws_bytes, self.ws_ptr if stats_groups else None, stats_groups, 0,
                          C.addressof(bwd_fuse[0]) if bwd_fuse is not None else None,
                          C.addressof(norm_fuse) if norm_fuse is not None else None)
        opi = {'conv_fwd': 0, 'conv_dgrad': 1, 'convT_fwd': 2, 'convT_dgrad': 3}[op]
        fn = [self.lib.gan_conv2d_fwd, self.lib.gan_conv2d_dgrad, self.lib.gan_convT2d_fwd, self.lib.gan_convT2d_dgrad][opi]
        need = self.lib.gan_conv_workspace_bytes(C.byref(d), opi)
        if need + STATS_RESERVE > self.ws_bytes:
            raise L.GanAmdError(f"workspace too small for {op}: need {need}")
        # fused statistics partials live behind the launch's own scratch (split-K slabs) in the lane workspace
        self.last_stats_ptr = self.ws_ptr + (need + 255) // 256 * 256
        if stats_groups:
            d.stats_partial = self.last_stats_ptr
            d.stats_partial_bytes = self.ws_bytes - (self.last_stats_ptr - self.ws_ptr)
        info = (C.c_int32 * 5)()
        self.lib.gan_conv_plan_info(C.byref(d), opi, info)
        self.last_bwd_fused = 0
        self.last_full = info[4] == -1       # GanNormFuse honoured: the launch finishes the layer (no norm ops follow)
        if not self.last_full:               # a request is either honoured or dropped HERE: the library refuses one it cannot honour
            d.norm_fuse = None
        if bwd_fuse is not None:
            self.last_stats_chunks = 0
            self.last_bwd_fused = max(info[4], 0)
            if not info[4]:            # this launch shape (thin kernel, odd group size ...) cannot carry it: plain dgrad
                d.bwd_fuse = None
                d.stats_partial, d.stats_groups = None, 0
        else:
            self.last_stats_chunks = max(info[4], 0) if stats_groups else 0
            if stats_groups and not info[4]:
                d.stats_partial, d.stats_groups = None, 0
        T = 4 if info[3] == 4 else 16
        # algorithmic FLOPs, SURVEY.md 8(d) convention: real channel counts, border taps not discounted
        creal = k_real or x.c
        M = x.n * (x.h * x.w if info[3] == 4 else y.h * y.w)
        flops = 2.0 * M * info[3] * T * y.c * creal
        kname = (f"conv_gemm<{self.ctx.dtype},{info[0]},{info[1]}>" if info[0] else
                 f"conv_own<{self.ctx.dtype}>" if info[1] == 8 else                     # csrc/conv_own.hip: the whole layer in one launch
                 f"conv_thin_{'n' if info[1] == 1 else 'k'}<{self.ctx.dtype}>")        # csrc/thin.hip streaming kernels
        if info[0] and info[2] > 1:
            kname += "+splitK"          # two kernels per call (GEMM into fp32 slabs + the slab-reduce kernel): timed as their own class
        meta = dict(kind='gemm', kernel=kname, flops=flops, splits=info[2],
                    shape=f"{op} M{M}{'x4' if info[3] == 4 else ''} N{y.c} K{T * x.c} s{info[2]}")
        # a launch that finishes its layer (GanNormFuse) on a small tile may join a layer stack (merge_stacks below)
        if self.last_full and self.lib.gan_conv_stack_eligible(C.byref(d), opi) == 1:
            meta['stack'] = (d, opi)
        return (fn, (self._desc(d),), op, meta)

    def wgrad(self, big, small, dw_ptr, big_c, small_c, stride, accumulate, adam_fuse=None, wire_ptr=None, ws_ptr=None, alt=False):
        """adam_fuse: a GanAdamFuse - ask the launch to apply the optimiser step itself; self.last_adam_fused tells whether it will.
        wire_ptr: this kernel's place in the bf16 wire buffer of the data-parallel exchange - ask the launch to write the gradient
        there (GanWgradDesc.dw_wire); self.last_wire_direct tells whether it will (then dw stays untouched)."""
        if adam_fuse is not None:
            self.keep.append(adam_fuse)
        d = L.GanWgradDesc(self.ctx.dt, stride, big, small, dw_ptr, big_c, small_c, int(accumulate),
                           ws_ptr or self.ws_side_ptr, self.ws_bytes, int(self.wgrad_concurrent),
                           C.addressof(adam_fuse) if adam_fuse is not None else None, wire_ptr if (wire_ptr and not accumulate and adam_fuse is None) else None)
        self.last_wire_direct = False
        if d.dw_wire:
            rc = self.lib.gan_wgrad_wire_direct(C.byref(d))
            if rc < 0:
                L.check(rc, "wgrad_wire_direct")
            self.last_wire_direct = rc == 1
            if not self.last_wire_direct:
                d.dw_wire = None
        self.last_adam_fused = False
        if adam_fuse is not None:
            rc = self.lib.gan_wgrad_adam_fused(C.byref(d))
            if rc < 0:
                L.check(rc, "wgrad_adam_fused")
            self.last_adam_fused = rc == 1
            if not self.last_adam_fused:
                d.adam_fuse = None
        need = self.lib.gan_wgrad_workspace_bytes(C.byref(d))
        if need > self.ws_bytes:
            raise L.GanAmdError(f"workspace too small for wgrad: need {need}")
        info = (C.c_int32 * 4)()
        self.lib.gan_wgrad_plan_info(C.byref(d), info)
        flops = 2.0 * small.n * small.h * small.w * 16 * big_c * small_c
        meta = dict(kind='gemm', kernel=f"wgrad<{self.ctx.dtype},{info[0]},{info[1]}>", flops=flops, splits=info[2], alt=bool(alt),
                    shape=f"wgrad A{big_c} B{small_c} M{small.n * small.h * small.w} s{info[2]}")
        return (self.lib.gan_conv_wgrad, (self._desc(d),), "conv_wgrad", meta, True)      # True: side-stream op

    def norm_names(self):
        return ('.gamma', '.beta') if self.norm == 'batchnorm' else ('.scale', '.offset')

    def norm_fwd(self, name, y, a, groups, mean, rstd, act, mask_ptr, stat_groups_update=True, fused_chunks=0, fused_ptr=None):
        gk, bk = self.norm_names()
        eps = BN_EPS if self.norm == 'batchnorm' else IN_EPS
        mm = mv = None
        if self.norm == 'batchnorm' and stat_groups_update:
            mm = self.P.state[name + '.moving_mean'].data_ptr()
            mv = self.P.state[name + '.moving_variance'].data_ptr()
        d = L.GanNormDesc(self.ctx.dt, y, a, groups, eps, self.P.ptr(name + gk), self.P.ptr(name + bk),
                          mean.data_ptr(), rstd.data_ptr(), mm, mv, BN_MOMENTUM, mask_ptr, L.ACTS[act], LEAKY_ALPHA,
                          self.ws_ptr, self.ws_bytes, self.ctx.new_sync())
        r = self._desc(d)
        one = bool(d.sync)     # finalize + apply as one call (one launch where the apply grid can carry the finalize)
        if fused_chunks:       # the producing convolution (epilogue or split-K reduce) already wrote the partials
            df = L.GanNormDesc.from_buffer_copy(d)
            df.workspace = fused_ptr or self.ws_ptr
            if one:
                df.workspace_bytes = self.ws_bytes - (df.workspace - self.ws_ptr)
                return [(self.lib.gan_norm_finalize_act_fwd, (self._desc(df), fused_chunks), f"norm_fin_act_fwd({name})")]
            return [(self.lib.gan_norm_stats_finalize, (self._desc(df), fused_chunks), f"norm_stats_finalize({name})"),
                    (self.lib.gan_norm_act_fwd, (r,), f"norm_act_fwd({name})")]
        if one:
            return [(self.lib.gan_norm_stats_partial, (r,), f"norm_stats_partial({name})"),
                    (self.lib.gan_norm_finalize_act_fwd, (r, 0), f"norm_fin_act_fwd({name})")]
        return [(self.lib.gan_norm_stats, (r,), f"norm_stats({name})"),
                (self.lib.gan_norm_act_fwd, (r,), f"norm_act_fwd({name})")]

    def norm_bwd(self, name, y, da, da2, dy, groups, mean_ptr, rstd_ptr, act, mask_ptr, want_param_grads, accumulate):
        gk, bk = self.norm_names()
        z = L.GanTensor(None, 0, 0, 0, 0, 0)
        d = L.GanNormBwdDesc(self.ctx.dt, y, da, da2 if da2 is not None else z, dy, groups,
                             self.P.ptr(name + gk), self.P.ptr(name + bk), mean_ptr, rstd_ptr, mask_ptr,
                             L.ACTS[act], LEAKY_ALPHA,
                             self.P.ptr(name + gk, 'grad') if want_param_grads else None,
                             self.P.ptr(name + bk, 'grad') if want_param_grads else None,
                             int(accumulate), self.ws_ptr, self.ws_bytes, self.ctx.new_sync())
        return (self.lib.gan_norm_act_bwd, (self._desc(d),), f"norm_act_bwd({name})")

    def norm_bwd_fused(self, name, y, dz, dy, groups, mean_ptr, rstd_ptr, chunks, partial_ptr, want_param_grads, accumulate):
        """Second half (finalize + apply) of a layer backward whose first half ran in the producing dgrad's epilogue."""
        gk, bk = self.norm_names()
        z = L.GanTensor(None, 0, 0, 0, 0, 0)
        d = L.GanNormBwdDesc(self.ctx.dt, y, dz, z, dy, groups, self.P.ptr(name + gk), self.P.ptr(name + bk), mean_ptr, rstd_ptr, None,
                             L.ACTS[None], LEAKY_ALPHA,
                             self.P.ptr(name + gk, 'grad') if want_param_grads else None,
                             self.P.ptr(name + bk, 'grad') if want_param_grads else None,
                             int(accumulate), partial_ptr, self.ws_bytes - (partial_ptr - self.ws_ptr), self.ctx.new_sync())
        return (self.lib.gan_norm_act_bwd_fused, (self._desc(d), chunks), f"norm_act_bwd_fused({name})")

    def act_bwd(self, a, da, da2, dy, act):
        z = L.GanTensor(None, 0, 0, 0, 0, 0)
        d = L.GanActBwdDesc(self.ctx.dt, a, da, da2 if da2 is not None else z, dy, L.ACTS[act], LEAKY_ALPHA, None, 0,
                            self.ws_ptr, self.ws_bytes)
        return (self.lib.gan_act_bwd, (self._desc(d),), "act_bwd")

    def bias_grad(self, dy, dbias_ptr, accumulate, side=False):
        """side: a side-stream op like the wgrad GEMMs (it only feeds Adam): scratch in the wgrad lane's workspace."""
        self.keep.append(dy)
        op = (self.lib.gan_bias_grad, (self.ctx.dt, C.byref(dy), dbias_ptr, int(accumulate), self.ws_side_ptr if side else self.ws_ptr,
                                       self.ws_bytes), "bias_grad")
        return op + (dict(kind='side', kernel='bias_grad', flops=0.0, shape='bias'), True) if side else op


class _Stack:
    """Host plan, its device copy and the barrier state of one layer stack (kept alive by the op list that launches it)."""

    def __init__(self, ctx, items):
        if torch.cuda.is_current_stream_capturing():
            raise L.GanAmdError("a layer stack's plan is uploaded when its op list is built: build the op lists before the capture "
                                "(steps.py: _prebuild_fused_adam / the warm-up step)")
        lib, n = ctx.lib, len(items)
        self.descs = [d for d, _ in items]
        arr = (C.c_void_p * n)(*[C.addressof(d) for d in self.descs])
        ops = (C.c_int32 * n)(*[o for _, o in items])
        nbytes = lib.gan_conv_stack_plan_bytes(n)
        self.host = C.create_string_buffer(nbytes)
        self.rc = lib.gan_conv_stack_plan(arr, ops, n, self.host, nbytes)
        if self.rc == 0:
            self.dev = torch.frombuffer(bytearray(self.host.raw), dtype=torch.uint8).to(ctx.device)
            self.bar = torch.zeros(lib.gan_conv_stack_barrier_bytes(), dtype=torch.uint8, device=ctx.device)
            self.args = (C.addressof(self.host), self.dev.data_ptr(), self.bar.data_ptr(), ctx.stack_err.data_ptr())


def merge_stacks(ctx, ops):
    """Replace every run of >= 2 consecutive stack-eligible convolution launches (small split-K layers finished by their slab
    reduce, each consuming what the one before it produced) by ONE persistent launch (gan_conv_stack_launch).  Adjacent ops only:
    anything between two such launches ends the run."""
    if not ctx.use_stacks:
        return ops
    out, run = [], []

    def flush():
        if len(run) >= 2:
            st = _Stack(ctx, [o[3]['stack'] for o in run])
            if st.rc:
                L.check(st.rc, "conv_stack_plan")        # (every layer answered gan_conv_stack_eligible() == 1: a failure here is an error)
            if st.rc == 0:
                meta = dict(kind='gemm', kernel=f"conv_stack<{ctx.dtype},{len(run)} layers>", flops=sum(o[3]['flops'] for o in run), splits=0,
                            shape="stack: " + " | ".join(o[3]['shape'] for o in run), keep=(st, list(run)))
                out.append((ctx.lib.gan_conv_stack_launch, st.args, "conv_stack", meta))
                run.clear()
                return
        out.extend(run)
        run.clear()
    for o in ops:
        if len(o) > 3 and isinstance(o[3], dict) and 'stack' in o[3] and not (len(o) > 4 and o[4]):
            run.append(o)
        else:
            flush()
            out.append(o)
    flush()
    return out


class GeneratorNet:
    """`GAN.Generator(norm_type, shape)` (base_gan.py:168-225): parameters + factory of call objects."""

    def __init__(self, ctx, channels, norm='batchnorm', seed=0):
        self.ctx, self.channels, self.norm = ctx, channels, norm
        self.spec = generator_spec(channels, norm)
        self.params = ParamSet(ctx, self.spec)
        self.params.load_numpy(init_params_numpy(self.spec, seed))

    def new_call(self, batch, size, dropout=True, seed=1234, stream_id=0, wgrads_on_side_lane=False, lane=0, guest_batch=0, host=None):
        """wgrads_on_side_lane: this call's kernel-gradient GEMMs run on a side lane of a captured step beside the dgrad chain
        (GanWgradDesc.concurrent: the planner prefers half-chip grids with longer reductions).  lane: whose workspace the call's
        launches use (0 | 2): calls that may run at the same time must not share one.
        guest_batch / host: two invocations of one generator whose kernel gradients are taken by ONE wgrad GEMM per layer over
        both (CycleGAN: no write + accumulate pair): the host call allocates every saved tensor `guest_batch` samples wider and
        the guest call (host=that call) lives in those extra samples; host.backward(wgrads='wide') then covers both."""
        return GenCall(self, batch, size, dropout, seed, stream_id, wgrads_on_side_lane, lane, guest_batch, host)


class GenCall:
    """Buffers + op lists for one invocation `generator(x, training=True)` and its backward."""

    def __init__(self, net, B, S, dropout, seed, stream_id, wgrads_on_side_lane=False, lane=0, guest_batch=0, host=None):
        ctx, P = net.ctx, net.params
        self.net, self.ctx, self.B, self.S, self.C = net, ctx, B, S, net.channels
        self.guest_batch, self.host, self._pool, self._pool_i = int(guest_batch), host, [], 0
        if host is not None and (host.guest_batch != B or host.net is not net or host.S != S):
            raise ValueError("guest call: batch / network / size must match what the host call reserved")

        def Buf(ctx_, n_, h_, w_, c_, tdtype=None, _B=globals()['Buf']):
            """Saved tensors: plain buffers, or batch slices of buffers shared by a host call and its guest (same creation order)."""
            if host is not None:
                big = host._pool[self._pool_i]
                self._pool_i += 1
                assert (big.h, big.w, big.c) == (h_, w_, c_) and big.n == host.B + n_
                return _B.slice_of(big, host.B, n_)
            if self.guest_batch:
                big = _B(ctx_, n_ + self.guest_batch, h_, w_, c_, tdtype)
                self._pool.append(big)
                return _B.slice_of(big, 0, n_)
            return _B(ctx_, n_, h_, w_, c_, tdtype)
        bd = _Builder(ctx, P, net.norm, lane=lane)
        bd.wgrad_concurrent = int(wgrads_on_side_lane)      # (2: beside a mirror chain, GanWgradDesc.concurrent)
        self._bd = bd
        C_ = net.channels
        groups = 1 if net.norm == 'batchnorm' else B
        self.groups = groups
        hs = [S >> (i + 1) for i in range(8)]                       # spatial size of down i output
        self.xin = Buf(ctx, B, S, S, 8)
        self.out = Buf(ctx, B, S, S, 8)                             # tanh output, channels [0, C)
        self.cat = [Buf(ctx, B, hs[6 - j], hs[6 - j], G_UP[j] + G_DOWN[6 - j]) for j in range(7)]
        self.a7 = Buf(ctx, B, hs[7], hs[7], 512)
        self.y_down = [None] + [Buf(ctx, B, hs[i], hs[i], G_DOWN[i]) for i in range(1, 8)]
        self.y_up = [Buf(ctx, B, hs[6 - j], hs[6 - j], G_UP[j]) for j in range(7)]
        dev = ctx.device
        f32 = torch.float32
        self.stats = {}

        def stat(name, c):
            self.stats[name] = (torch.zeros(groups * c, dtype=f32, device=dev), torch.zeros(groups * c, dtype=f32, device=dev))
            return self.stats[name]

        try:        # data-parallel replicas draw different masks (the hash mixes the rank in through the seed)
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                seed = (seed + 0x9E3779B1 * dist.get_rank()) & 0x7FFFFFFFFFFFFFFF
        except Exception:
            pass
        self.masks = [torch.ones((B, hs[6 - j], hs[6 - j], 512), dtype=torch.uint8, device=dev) for j in range(3)] if dropout else None
        self.mask_ops = []
        if dropout:          # the three Dropout(0.5) masks of this call in one launch
            self._mask_args = ((C.c_void_p * 3)(*[m.data_ptr() for m in self.masks]), (C.c_int64 * 3)(*[m.numel() for m in self.masks]),
                               (C.c_uint32 * 3)(*[stream_id * 8 + j for j in range(3)]))
            # per-call launch counter (advanced by the kernel): validation passes draw new masks although the Adam step stands still
            self.mask_draws = torch.zeros(2, dtype=torch.int32, device=dev)
            self.mask_ops.append((ctx.lib.gan_dropout_mask_multi, (3, self._mask_args[0], self._mask_args[1], seed, P.step.data_ptr(),
                                                                   self._mask_args[2], self.mask_draws.data_ptr()), "dropout_mask_multi"))
        self.auto_masks = dropout

        def a_down(i):      # activation view of down i
            if i == 7:
                return self.a7.view()
            j = 6 - i
            return self.cat[j].view(G_UP[j], G_DOWN[i])

        # ---------------- forward ----------------
        fwd = []
        x = self.xin.view()
        for i in range(8):
            name = f'down{i}'
            if i == 3:
                self.fwd_inner_start = len(fwd)      # from here on (M <= 4096 rows) the layers leave most of the chip idle
            w = P.tr[name + '.kernel']
            if i == 0:      # conv -> LeakyReLU fused in the GEMM epilogue (apply_norm=False, base_gan.py:180)
                fwd.append(bd.conv('conv_fwd', x, a_down(0), w.data_ptr(), G_DOWN[0], 2, None, 'lrelu', k_real=C_))
            else:
                mean, rstd = stat(name, G_DOWN[i])
                fwd.append(bd.conv('conv_fwd', x, self.y_down[i].view(), w.data_ptr(), G_DOWN[i], 2, stats_groups=groups,
                                   norm_fuse=bd.fwd_norm_fuse(name, a_down(i), groups, mean, rstd, 'lrelu', None)))
                if not bd.last_full:
                    fwd += bd.norm_fwd(name, self.y_down[i].view(), a_down(i), groups, mean, rstd, 'lrelu', None,
                                       fused_chunks=bd.last_stats_chunks, fused_ptr=bd.last_stats_ptr)
            x = a_down(i)
        for j in range(7):
            name = f'up{j}'
            xin = self.a7.view() if j == 0 else self.cat[j - 1].view()
            w = P.nat[name + '.kernel']
            mean, rstd = stat(name, G_UP[j])
            mptr = self.masks[j].data_ptr() if (dropout and j < 3) else None
            fwd.append(bd.conv('convT_fwd', xin, self.y_up[j].view(), w.data_ptr(), G_UP[j], 2, stats_groups=groups,
                               norm_fuse=bd.fwd_norm_fuse(name, self.cat[j].view(0, G_UP[j]), groups, mean, rstd, 'relu', mptr)))
            if not bd.last_full:
                fwd += bd.norm_fwd(name, self.y_up[j].view(), self.cat[j].view(0, G_UP[j]), groups, mean, rstd, 'relu', mptr,
                                   fused_chunks=bd.last_stats_chunks, fused_ptr=bd.last_stats_ptr)
        fwd.append(bd.conv('convT_fwd', self.cat[6].view(), self.out.view(0, C_), P.nat['last.kernel'].data_ptr(), C_, 2,
                           P.ptr('last.bias'), 'tanh'))
        head = merge_stacks(ctx, fwd[:self.fwd_inner_start])
        self.fwd_ops = head + merge_stacks(ctx, fwd[self.fwd_inner_start:])
        self.fwd_inner_start = len(head)

        # ---------------- backward ----------------
        self.dcat = [Buf(ctx, B, hs[6 - j], hs[6 - j], G_UP[j] + G_DOWN[6 - j]) for j in range(7)]
        self.da7 = Buf(ctx, B, hs[7], hs[7], 512)
        self.dA = [Buf(ctx, B, hs[i], hs[i], G_DOWN[i]) for i in range(7)]     # grad wrt a_i from down i+1
        self.dy_down = [Buf(ctx, B, hs[i], hs[i], G_DOWN[i]) for i in range(8)]
        self.dy_up = [Buf(ctx, B, hs[6 - j], hs[6 - j], G_UP[j]) for j in range(7)]
        self.dpre = Buf(ctx, B, S, S, 8)
        self.dgen = Buf(ctx, B, S, S, 8)      # upstream gradient slot 1 (e.g. L1 term), channels [0, C)
        self.dgen2 = Buf(ctx, B, S, S, 8)     # upstream gradient slot 2 (e.g. from the discriminator)
        self.dxin = Buf(ctx, B, S, S, 8)
        self._bwd_cache = {}
        self.adam_fused = {}
        self.wire_direct = {}
        # kernels whose wgrad launch runs on a SECOND wgrad lane (own slab workspace) in the staged schedule: set by the step object
        # before the first backward op list is built (Pix2PixStep.wgrad_alt)
        self.alt_wgrad = frozenset()
        self.wgrad_stream2 = None

    def _build_bwd(self, use_dgen2, need_dx, accumulate, wgrads='own', adam=None, wire=None):
        """wgrads: 'own' - this call's kernel gradients (accumulate as the other gradients do); 'none' - a guest call whose host
        takes them; 'wide' - a host call: one wgrad GEMM per layer over its own AND its guest's samples, plain write (the guest's
        saved activations and output gradients must be in place: its forward and its backward(wgrads='none') have run).
        adam = (beta_1, beta_2): every wgrad launch that can applies this step's Adam update to its kernel itself (GanAdamFuse;
        needs plain-write kernel gradients: wgrads 'own' without accumulate, or 'wide'); self.adam_fused[key] lists those kernels.
        wire = data pointer of this network's bf16 wire buffer (data-parallel exchange, ParamSet offsets): every wgrad launch that can
        writes its gradient there in the wire format instead of the fp32 buffer (GanWgradDesc.dw_wire); self.wire_direct[key] lists
        those kernels - the caller casts (gan_grad_pack) only the rest of a bucket.
        The caller runs gan_adam_begin BEFORE this list and ParamSet.adam_rest_ops() for everything else after it, and must not
        start a layer's wgrad before every dgrad of the step that reads the layer's weights has been enqueued (staged order).
        Backward op list.  Wherever the launch shape allows it, the dgrad that produces the gradient w.r.t. a layer's
        activation starts that layer's backward in its epilogue (dz + partial sums; `fused` = chunk count) and the layer is
        finished by finalize + apply; otherwise the three-launch normalisation backward / act_bwd follows."""
        bd, P, C_, groups = self._bd, self.net.params, self.C, self.groups
        if wgrads == 'wide' and not self.guest_batch:
            raise ValueError("wgrads='wide' needs a host call (guest_batch > 0)")
        W = (lambda b_: b_.wide()) if wgrads == 'wide' else (lambda b_: b_)
        wacc = accumulate and wgrads != 'wide'

        fused_names, wire_names = [], []
        if adam is not None and wacc:
            raise ValueError("adam fusion needs plain-write kernel gradients")

        def wgrad(big_v, small_v, kname, big_c, small_c, stride):
            if wgrads == 'none':
                return
            af = P.adam_fuse_desc(kname, adam[0], adam[1]) if adam is not None else None
            wp = wire + 2 * P.entries[kname][0] if wire else None
            alt = kname in self.alt_wgrad
            ops.append(bd.wgrad(big_v, small_v, P.ptr(kname, 'grad'), big_c, small_c, stride, wacc, adam_fuse=af, wire_ptr=wp,
                                ws_ptr=self.ctx.ws_lanes[5].data_ptr() if alt else None, alt=alt))
            if af is not None and bd.last_adam_fused:
                fused_names.append(kname)
            if bd.last_wire_direct:
                wire_names.append(kname)
        ops = []
        ops.append(bd.act_bwd(self.out.view(), self.dgen.view(), self.dgen2.view() if use_dgen2 else None,
                              self.dpre.view(), 'tanh'))
        wgrad(W(self.dpre).view(), W(self.cat[6]).view(), 'last.kernel', C_, 128, 2)
        ops.append(bd.bias_grad(self.dpre.view(), P.ptr('last.bias', 'grad'), accumulate, side=bool(getattr(self, 'bias_grad_on_side', False))))

        def up_spec(j):         # backward of up j (ReLU [+ dropout] after the norm) on the leading G_UP[j] channels of dcat[j]
            mean, rstd = self.stats[f'up{j}']
            mptr = self.masks[j].data_ptr() if (self.masks is not None and j < 3) else None
            return bd.fuse_spec(self.y_up[j].view(), None, f'up{j}', groups, mean.data_ptr(), rstd.data_ptr(), 'relu', mptr, 512,
                                cols=G_UP[j])

        ops.append(bd.conv('convT_dgrad', self.dpre.view(), self.dcat[6].view(), P.tr['last.kernel'].data_ptr(), 128, 2, k_real=C_,
                           bwd_fuse=up_spec(6)))
        fused, fptr, full = bd.last_bwd_fused, bd.last_stats_ptr, False
        for j in range(6, -1, -1):
            name = f'up{j}'
            mean, rstd = self.stats[name]
            mptr = self.masks[j].data_ptr() if (self.masks is not None and j < 3) else None
            if full:         # the dgrad above finished this layer's backward in its slab-reduce kernel (GanNormFuse)
                pass
            elif fused:
                ops.append(bd.norm_bwd_fused(name, self.y_up[j].view(), self.dcat[j].view(0, G_UP[j]), self.dy_up[j].view(), groups,
                                             mean.data_ptr(), rstd.data_ptr(), fused, fptr, True, accumulate))
            else:
                ops.append(bd.norm_bwd(name, self.y_up[j].view(), self.dcat[j].view(0, G_UP[j]), None, self.dy_up[j].view(),
                                       groups, mean.data_ptr(), rstd.data_ptr(), 'relu', mptr, True, accumulate))
            xin = self.a7 if j == 0 else self.cat[j - 1]
            dxin = self.da7 if j == 0 else self.dcat[j - 1]
            cin = xin.c
            wgrad(W(self.dy_up[j]).view(), W(xin).view(), name + '.kernel', G_UP[j], cin, 2)
            if j > 0:
                spec, nfz = up_spec(j - 1), bd.bwd_norm_fuse(f'up{j - 1}', self.dy_up[j - 1].view(), True, accumulate)
            else:               # da7: gradient w.r.t. the bottleneck activation = down7's backward
                m7, r7 = self.stats['down7']
                spec = bd.fuse_spec(self.y_down[7].view(), None, 'down7', groups, m7.data_ptr(), r7.data_ptr(), 'lrelu')
                nfz = bd.bwd_norm_fuse('down7', self.dy_down[7].view(), True, accumulate)
            ops.append(bd.conv('convT_dgrad', self.dy_up[j].view(), dxin.view(), P.tr[name + '.kernel'].data_ptr(), cin, 2, bwd_fuse=spec,
                               norm_fuse=nfz))
            fused, fptr, full = bd.last_bwd_fused, bd.last_stats_ptr, bd.last_full
        dy0 = self.dy_down[0]
        for i in range(7, -1, -1):
            name = f'down{i}'
            if i == 7:
                da, da2 = self.da7.view(), None
            else:
                j = 6 - i
                da, da2 = self.dcat[j].view(G_UP[j], G_DOWN[i]), self.dA[i].view()
            if i == 0:
                if fused:
                    dy0 = self.dA[0]        # the epilogue of down1's dgrad already applied LeakyReLU' and the skip gradient
                else:
                    ops.append(bd.act_bwd(self.cat[6].view(G_UP[6], 64), da, da2, self.dy_down[0].view(), 'lrelu'))
            else:
                mean, rstd = self.stats[name]
                if full:
                    pass
                elif fused:
                    dz = self.da7.view() if i == 7 else self.dA[i].view()
                    ops.append(bd.norm_bwd_fused(name, self.y_down[i].view(), dz, self.dy_down[i].view(), groups,
                                                 mean.data_ptr(), rstd.data_ptr(), fused, fptr, True, accumulate))
                else:
                    ops.append(bd.norm_bwd(name, self.y_down[i].view(), da, da2, self.dy_down[i].view(), groups,
                                           mean.data_ptr(), rstd.data_ptr(), 'lrelu', None, True, accumulate))
            dyi = dy0 if i == 0 else self.dy_down[i]
            if i == 0:
                xin, cin_real = W(self.xin).view(), C_
                self.dy0_in_dA = dy0 is self.dA[0]        # (host and guest must agree on where down0's output gradient lives)
            elif i == 1:
                xin, cin_real = W(self.cat[6]).view(G_UP[6], 64), 64
            else:
                jj = 6 - (i - 1)
                xin, cin_real = W(self.cat[jj]).view(G_UP[jj], G_DOWN[i - 1]), G_DOWN[i - 1]
            wgrad(xin, W(dyi).view(), name + '.kernel', cin_real, G_DOWN[i], 2)
            fused, full = 0, False
            if i > 0:
                jb = 6 - (i - 1)                                     # layer below: down i-1, its skip gradient sits in dcat[jb]
                skip = self.dcat[jb].view(G_UP[jb], G_DOWN[i - 1])
                if i - 1 == 0:
                    spec = bd.fuse_spec(self.cat[6].view(G_UP[6], 64), skip, None, act='lrelu')
                else:
                    mb, rb = self.stats[f'down{i - 1}']
                    spec = bd.fuse_spec(self.y_down[i - 1].view(), skip, f'down{i - 1}', groups, mb.data_ptr(), rb.data_ptr(), 'lrelu')
                ops.append(bd.conv('conv_dgrad', dyi.view(), self.dA[i - 1].view(),
                                   P.nat[name + '.kernel'].data_ptr(), G_DOWN[i - 1], 2, bwd_fuse=spec,
                                   norm_fuse=bd.bwd_norm_fuse(f'down{i - 1}', self.dy_down[i - 1].view(), True, accumulate) if i - 1 > 0 else None))
                fused, fptr, full = bd.last_bwd_fused, bd.last_stats_ptr, bd.last_full
            elif need_dx:
                ops.append(bd.conv('conv_dgrad', dyi.view(), self.dxin.view(0, C_),
                                   P.nat[name + '.kernel'].data_ptr(), C_, 2))
        self.adam_fused[(use_dgen2, need_dx, accumulate, wgrads, adam)] = tuple(fused_names)
        self.wire_direct[(use_dgen2, need_dx, accumulate, wgrads, adam, wire)] = tuple(wire_names)
        return ops

    # public API ------------------------------------------------------------------------------
    def set_dropmasks(self, masks):
        """Explicit 0/1 masks (parity tests); disables the per-step generator."""
        for t, m in zip(self.masks, masks):
            t.copy_(torch.from_numpy(np.asarray(m)).to(torch.uint8))
        self.auto_masks = False

    def set_input(self, x_f32):
        """x: dense fp32 NHWC [B,S,S,C] device tensor -> typed, 8-channel-padded input buffer."""
        dst = self.xin.view(0, self.C)
        rc = self.ctx.lib.gan_pack(self.ctx.dt, x_f32.data_ptr(), C.byref(dst), self.ctx.stream())
        L.check(rc, "pack")

    def set_input_view(self, src_view):
        dst = self.xin.view(0, self.C)
        L.check(self.ctx.lib.gan_copy_view(self.ctx.dt, C.byref(src_view), C.byref(dst), self.ctx.stream()), "copy_view")

    def forward(self, inner_hook=None, masks_done=False):
        """inner_hook: called when the op list reaches the inner layers (down3): a place to start independent work on
        another lane that then runs beside the launch-latency-bound part of the generator.  masks_done: the caller has already
        enqueued self.mask_ops elsewhere (a side lane) and orders them before the decoder itself."""
        if self.auto_masks and not masks_done:
            self.ctx.run(self.mask_ops)
        if inner_hook is None:
            self.ctx.run(self.fwd_ops)
        else:
            self.ctx.run(self.fwd_ops[:self.fwd_inner_start])
            inner_hook()
            self.ctx.run(self.fwd_ops[self.fwd_inner_start:])

    def out_view(self, n0=0, n=None):
        return self.out.view(0, self.C, n0, n)

    def half(self, n0, n):
        """Samples [n0, n0 + n) of this call as a call-like object (CycleGAN batches two logical calls of one generator)."""
        return CallSlice(self, n0, n)

    def _bwd_key(self, use_dgen2, need_dx, accumulate, wgrads, adam, wire=None):
        if wire:
            return (use_dgen2, need_dx, accumulate, wgrads, adam, wire)
        return (use_dgen2, need_dx, accumulate) if wgrads == 'own' and adam is None else (use_dgen2, need_dx, accumulate, wgrads, adam)

    def backward(self, use_dgen2=False, need_dx=False, accumulate=False, defer_wgrads=False, wgrads='own', adam=None):
        """Upstream gradient(s) w.r.t. the tanh output must be in self.dgen (and self.dgen2).  wgrads / adam: see _build_bwd."""
        key = self._bwd_key(use_dgen2, need_dx, accumulate, wgrads, adam)
        if key not in self._bwd_cache:
            self._bwd_cache[key] = self._build_bwd(*key)
        ops = self._bwd_cache[key]
        if defer_wgrads == 'staged':
            # the wgrad GEMMs (they only feed Adam) run on `self.wgrad_stream` in a few coarse stages: those of the
            # layers before cut k start when the main dgrad/norm chain has passed cut k - per-op dependencies thrash
            # (two LDS-bound GEMMs on the same CUs), one stage at the very end leaves the tail serial
            main = self.ctx.lane_stream(0)
            for k, (main_ops, w_ops) in enumerate(self.bwd_stages(self.wgrad_cuts, use_dgen2, need_dx, accumulate, wgrads, adam)):
                self.ctx.run(main_ops)
                self.wgrad_stream.wait_stream(main)
                two = self.wgrad_stream2 is not None
                w2 = [o for o in w_ops if two and o[3].get('alt')]
                self.ctx.run_on([o for o in w_ops if not (two and o[3].get('alt'))], self.wgrad_stream)
                if w2:           # second wgrad lane (forked from the origin stream, like every side lane)
                    self.wgrad_stream2.wait_stream(main)
                    self.ctx.run_on(w2, self.wgrad_stream2)
                if getattr(self, 'stage_hook', None) is not None:
                    self.stage_hook(k)         # e.g. the optimiser step of the layers whose gradients are now complete
        else:
            mkey = ('all', key)
            if mkey not in self._bwd_cache:
                self._bwd_cache[mkey] = merge_stacks(self.ctx, ops)      # (adjacent launches only: wgrads between two dgrads end a run)
            self.ctx.run(self._bwd_cache[mkey])

    def bwd_ops(self, use_dgen2=False, need_dx=False, accumulate=False, wgrads='own', adam=None):
        """The (cached) backward op list; building it also settles self.dy0_in_dA and self.adam_fused."""
        key = self._bwd_key(use_dgen2, need_dx, accumulate, wgrads, adam)
        if key not in self._bwd_cache:
            self._bwd_cache[key] = self._build_bwd(*key)
        return self._bwd_cache[key]

    def bwd_stages(self, cuts, use_dgen2=False, need_dx=False, accumulate=False, wgrads='own', adam=None, wire=None):
        """The backward op list cut into coarse stages at the wgrad indices `cuts` (the same cuts as the 'staged'
        mode): [(main-chain ops, wgrad ops)] per stage.  The wgrad GEMMs of a stage only feed Adam, so a caller may
        run them beside the NEXT stage's main chain (gan_amd/steps.py data-parallel schedule)."""
        key = self._bwd_key(use_dgen2, need_dx, accumulate, wgrads, adam, wire)
        if key not in self._bwd_cache:
            self._bwd_cache[key] = self._build_bwd(use_dgen2, need_dx, accumulate, wgrads, adam, wire)
        ops = self._bwd_cache[key]
        is_w = lambda o: len(o) > 4 and o[4]
        idx = [i for i, o in enumerate(ops) if is_w(o) and o[2] == 'conv_wgrad']     # (other side ops ride in the stage they fall into)
        bounds = [0] + [idx[c] for c in cuts if 0 < c < len(idx)] + [len(ops)]
        skey = ('stages', key, tuple(bounds))          # (cached: merged layer stacks hold device plans, built outside any capture)
        if skey not in self._bwd_cache:
            self._bwd_cache[skey] = [(merge_stacks(self.ctx, [o for o in ops[lo:hi] if not is_w(o)]), [o for o in ops[lo:hi] if is_w(o)])
                                     for lo, hi in zip(bounds[:-1], bounds[1:])]
        return self._bwd_cache[skey]

    def output_f32(self):
        o = torch.empty((self.B, self.S, self.S, self.C), dtype=torch.float32, device=self.ctx.device)
        v = self.out_view()
        L.check(self.ctx.lib.gan_unpack(self.ctx.dt, C.byref(v), o.data_ptr(), self.ctx.stream()), "unpack")
        return o


class CallSlice:
    """A batch slice of a GenCall that stands for one logical generator invocation (masks, output, gradient slots)."""

    def __init__(self, call, n0, n):
        self.call, self.n0, self.n, self.C = call, n0, n, call.C

    def set_dropmasks(self, masks):
        for t, m in zip(self.call.masks, masks):
            t[self.n0:self.n0 + self.n].copy_(torch.from_numpy(np.asarray(m)).to(torch.uint8))
        self.call.auto_masks = False

    def out_view(self):
        return self.call.out.view(0, self.C, self.n0, self.n)

    def xin_view(self):
        return self.call.xin.view(0, self.C, self.n0, self.n)

    def dgen_view(self, second=False):
        return (self.call.dgen2 if second else self.call.dgen).view(0, self.C, self.n0, self.n)

    def output_f32(self):
        return self.call.output_f32()[self.n0:self.n0 + self.n]


# --------------------------------------------------------------------------------------------------
class DiscriminatorNet:
    """`GAN.Discriminator(norm_type, target)` (base_gan.py:124-166)."""

    def __init__(self, ctx, channels, target=True, norm='batchnorm', seed=1):
        self.ctx, self.channels, self.target, self.norm = ctx, channels, target, norm
        self.cin = channels * (2 if target else 1)
        self.spec = discriminator_spec(channels, target, norm)
        self.params = ParamSet(ctx, self.spec)
        self.params.load_numpy(init_params_numpy(self.spec, seed))

    def new_call(self, batch, size, calls=2, lane=0, params_lane=2):
        """lane / params_lane: whose workspace the forward + input-gradient chain / the parameter-gradient pass use."""
        return DiscCall(self, batch, size, calls, lane, params_lane)


class DiscCall:
    """`calls` invocations of the discriminator batched along N (e.g. D(real) ++ D(fake), pix2pix.py:202-203);
    BatchNormalization statistics stay per invocation (groups = calls).  Backward pass "A" covers the whole
    batch with parameter gradients (discriminator loss); pass "B" is the input-gradient-only chain over the
    last invocation (generator's adversarial loss through D(fake), pix2pix.py:210)."""

    LAYERS = [('down0', 64, 2), ('down1', 128, 2), ('down2', 256, 2), ('conv', 512, 1), ('last', 1, 1)]

    def __init__(self, net, B, S, calls, lane=0, params_lane=2):
        ctx, P = net.ctx, net.params
        self.net, self.ctx, self.B, self.S, self.calls = net, ctx, B, S, calls
        N = B * calls
        self.N = N
        bd = _Builder(ctx, P, net.norm, lane=lane)
        self._bd = bd
        self._bd2 = _Builder(ctx, P, net.norm, lane=params_lane)      # parameter-gradient pass: second chain
        bn = net.norm == 'batchnorm'
        groups = calls if bn else N
        self.groups = groups
        self.gper = 1 if bn else B           # stat groups per invocation
        s1, s2, s3 = S // 2, S // 4, S // 8
        s4, s5 = s3 - 1, s3 - 2
        self.xin = Buf(ctx, N, S, S, 8)
        self.a0 = Buf(ctx, N, s1, s1, 64)
        self.y = {'down1': Buf(ctx, N, s2, s2, 128), 'down2': Buf(ctx, N, s3, s3, 256), 'conv': Buf(ctx, N, s4, s4, 512)}
        self.a = {'down0': self.a0, 'down1': Buf(ctx, N, s2, s2, 128), 'down2': Buf(ctx, N, s3, s3, 256),
                  'conv': Buf(ctx, N, s4, s4, 512)}
        self.logits = Buf(ctx, N, s5, s5, 1, torch.float32)
        dev, f32 = ctx.device, torch.float32
        self.stats = {k: (torch.zeros(groups * c, dtype=f32, device=dev), torch.zeros(groups * c, dtype=f32, device=dev))
                      for k, c in [('down1', 128), ('down2', 256), ('conv', 512)]}
        fwd = [bd.conv('conv_fwd', self.xin.view(), self.a0.view(), P.tr['down0.kernel'].data_ptr(), 64, 2, None, 'lrelu', k_real=net.cin)]
        prev = self.a0
        for name, co, stride in self.LAYERS[1:4]:
            mean, rstd = self.stats[name]
            fwd.append(bd.conv('conv_fwd', prev.view(), self.y[name].view(), P.tr[name + '.kernel'].data_ptr(), co, stride,
                               stats_groups=groups, norm_fuse=bd.fwd_norm_fuse(name, self.a[name].view(), groups, mean, rstd, 'lrelu', None)))
            if not bd.last_full:
                fwd += bd.norm_fwd(name, self.y[name].view(), self.a[name].view(), groups, mean, rstd, 'lrelu', None,
                                   fused_chunks=bd.last_stats_chunks, fused_ptr=bd.last_stats_ptr)
            prev = self.a[name]
        fwd.append(bd.conv('conv_fwd', prev.view(), self.logits.view(), P.tr['last.kernel'].data_ptr(), 1, 1,
                           P.ptr('last.bias'), None, 1))
        self.fwd_ops = merge_stacks(ctx, fwd)        # (runs start after down3: fwd_inner_start stays valid)
        # backward buffers (pass B reuses the first B samples' worth of them)
        self.dlogits = Buf(ctx, N, s5, s5, 8)        # pass A: all invocations (channel 0 real)
        self.dlogits_b = Buf(ctx, B, s5, s5, 8)      # pass B: one invocation
        self.dA = {'conv': Buf(ctx, N, s4, s4, 512), 'down2': Buf(ctx, N, s3, s3, 256), 'down1': Buf(ctx, N, s2, s2, 128),
                   'down0': Buf(ctx, N, s1, s1, 64)}
        self.dy = {'conv': Buf(ctx, N, s4, s4, 512), 'down2': Buf(ctx, N, s3, s3, 256), 'down1': Buf(ctx, N, s2, s2, 128),
                   'down0': Buf(ctx, N, s1, s1, 64)}
        self.dxin = Buf(ctx, B, S, S, 8)
        self._cache = {}

    def forward(self):
        self.ctx.run(self.fwd_ops)

    def forward_part_ops(self, call, lane=0):
        """Forward of ONE invocation (batch slice `call`) as its own op list: the same results as the batched forward
        (statistics are per invocation anyway), usable on another lane - D(real) does not depend on the generator."""
        key = ('F', call, lane)
        if key not in self._cache:
            bd, P, net, B = (self._bd2 if lane == 2 else self._bd), self.net.params, self.net, self.B
            n0, gper = call * B, self.gper
            sv = lambda buf: buf.view(0, None, n0, B)
            ops = [bd.conv('conv_fwd', sv(self.xin), sv(self.a0), P.tr['down0.kernel'].data_ptr(), 64, 2, None, 'lrelu', k_real=net.cin)]
            prev = self.a0
            for name, co, stride in self.LAYERS[1:4]:
                mean, rstd = self.stats[name]
                lo, hi = call * gper * co, (call + 1) * gper * co
                ops.append(bd.conv('conv_fwd', sv(prev), sv(self.y[name]), P.tr[name + '.kernel'].data_ptr(), co, stride,
                                   stats_groups=gper,
                                   norm_fuse=bd.fwd_norm_fuse(name, sv(self.a[name]), gper, mean[lo:hi], rstd[lo:hi], 'lrelu', None)))
                if not bd.last_full:
                    ops += bd.norm_fwd(name, sv(self.y[name]), sv(self.a[name]), gper, mean[lo:hi], rstd[lo:hi], 'lrelu', None,
                                       fused_chunks=bd.last_stats_chunks, fused_ptr=bd.last_stats_ptr)
                prev = self.a[name]
            ops.append(bd.conv('conv_fwd', sv(prev), sv(self.logits), P.tr['last.kernel'].data_ptr(), 1, 1,
                               P.ptr('last.bias'), None, 1))
            self._cache[key] = merge_stacks(self.ctx, ops)
        return self._cache[key]

    def logits_view(self, call):
        """fp32 logits of invocation `call`: (ptr, count)."""
        per = self.B * self.logits.h * self.logits.w
        return self.logits.t.data_ptr() + 4 * call * per, per

    def dlogits_ptr(self, call):
        per = self.B * self.dlogits.h * self.dlogits.w * 8
        return self.dlogits.t.data_ptr() + call * per * self.dlogits.t.element_size()

    def _chain(self, n0, n, groups, stat_off, wgrads, need_dx, accumulate, dx_dst=None, dx_c0=0, wire=None):
        """Backward over samples [n0, n0+n).  Scratch gradients always live at samples [0, n) of the dA/dy
        buffers; saved forward tensors are read at [n0, n0+n)."""
        bd, P = (self._bd2 if wgrads else self._bd), self.net.params
        ops = []
        wire_names = []

        def wg(big_v, small_v, kname, big_c, small_c, stride):
            ops.append(bd.wgrad(big_v, small_v, P.ptr(kname, 'grad'), big_c, small_c, stride, accumulate,
                                wire_ptr=wire + 2 * P.entries[kname][0] if wire else None))
            if bd.last_wire_direct:
                wire_names.append(kname)
        sv = lambda buf: buf.view(0, None, n0, n)          # saved forward tensors
        gv = lambda buf: buf.view(0, None, 0, n)           # gradient scratch
        dl = self.dlogits.view(0, None, n0, n) if wgrads else self.dlogits_b.view()
        # last: conv s1 with bias, no activation
        if wgrads:
            wg(sv(self.a['conv']), dl, 'last.kernel', 512, 1, 1)
            ops.append(bd.bias_grad(dl, P.ptr('last.bias', 'grad'), accumulate))
        def spec_for(layer):          # backward of `layer` started in the epilogue of the dgrad that produces dA[layer]
            if layer == 'down0':
                return bd.fuse_spec(sv(self.a0), None, None, act='lrelu')
            m_, r_ = self.stats[layer]
            c_ = self.y[layer].c
            return bd.fuse_spec(sv(self.y[layer]), None, layer, groups, m_.data_ptr() + 4 * stat_off * c_,
                                r_.data_ptr() + 4 * stat_off * c_, 'lrelu')

        ops.append(bd.conv('conv_dgrad', dl, gv(self.dA['conv']), P.nat['last.kernel'].data_ptr(), 512, 1, k_real=1,
                           bwd_fuse=spec_for('conv')))
        fused, fptr, full = bd.last_bwd_fused, bd.last_stats_ptr, False
        order = [('conv', 'down2', 1, 256), ('down2', 'down1', 2, 128), ('down1', 'down0', 2, 64)]
        for name, prev, stride, cprev in order:
            mean, rstd = self.stats[name]
            c = self.y[name].c
            if full:
                pass
            elif fused:
                ops.append(bd.norm_bwd_fused(name, sv(self.y[name]), gv(self.dA[name]), gv(self.dy[name]), groups,
                                             mean.data_ptr() + 4 * stat_off * c, rstd.data_ptr() + 4 * stat_off * c, fused, fptr,
                                             wgrads, accumulate))
            else:
                ops.append(bd.norm_bwd(name, sv(self.y[name]), gv(self.dA[name]), None, gv(self.dy[name]), groups,
                                       mean.data_ptr() + 4 * stat_off * c, rstd.data_ptr() + 4 * stat_off * c, 'lrelu', None,
                                       wgrads, accumulate))
            if wgrads:
                wg(sv(self.a[prev]), gv(self.dy[name]), name + '.kernel', cprev, c, stride)
            ops.append(bd.conv('conv_dgrad', gv(self.dy[name]), gv(self.dA[prev]), P.nat[name + '.kernel'].data_ptr(), cprev, stride,
                               bwd_fuse=spec_for(prev),
                               norm_fuse=bd.bwd_norm_fuse(prev, gv(self.dy[prev]), wgrads, accumulate) if prev != 'down0' else None))
            fused, fptr, full = bd.last_bwd_fused, bd.last_stats_ptr, bd.last_full
        dy0 = self.dA['down0'] if fused else self.dy['down0']
        if not fused:
            ops.append(bd.act_bwd(sv(self.a0), gv(self.dA['down0']), None, gv(self.dy['down0']), 'lrelu'))
        if wgrads:
            wg(sv(self.xin), gv(dy0), 'down0.kernel', self.net.cin, 64, 2)
        if need_dx:
            if dx_dst is not None:      # only channels [dx_c0, dx_c0 + dx_dst.c) of the input gradient, written where the caller wants them
                wk = P.nat['down0.kernel']
                ops.append(bd.conv('conv_dgrad', gv(dy0), dx_dst, wk.data_ptr() + dx_c0 * wk.shape[2] * wk.element_size(), self.net.cin, 2))
            else:
                ops.append(bd.conv('conv_dgrad', gv(dy0), self.dxin.view(0, self.net.cin, 0, n),
                                   P.nat['down0.kernel'].data_ptr(), self.net.cin, 2))
        if wire:
            self.wire_direct = tuple(wire_names)
        return merge_stacks(self.ctx, ops)       # (adjacent launches only: a wgrad between two dgrads ends a run)

    def params_ops(self, accumulate=False, wire=None):
        """wire: data pointer of the network's bf16 wire buffer - wgrad launches that can write their gradient there directly
        (GanWgradDesc.dw_wire); self.wire_direct then lists those kernels."""
        key = ('A', accumulate, wire) if wire else ('A', accumulate)
        if key not in self._cache:
            self._cache[key] = self._chain(0, self.N, self.groups, 0, True, False, accumulate, wire=wire)
        return self._cache[key]

    def backward_params(self, accumulate=False):
        """Pass A: dlogits (all invocations, written by the loss kernels) -> parameter gradients."""
        key = ('A', accumulate)
        if key not in self._cache:
            self._cache[key] = self._chain(0, self.N, self.groups, 0, True, False, accumulate)
        self.ctx.run(self._cache[key])      # (the ops carry lane-2/3 workspaces either way)

    def backward_input(self, call, dst=None, c0=0):
        """Pass B: gradient w.r.t. the input of invocation `call`; its dlogits must be in self.dlogits_b.
        Result in self.dxin (8-channel padded), or - dst: a GanTensor view - input channels [c0, c0 + dst.c) straight into
        dst (the generator's upstream-gradient slot: no copy, and for Pix2Pix half the output channels of the layer)."""
        key = ('B', call, dst.ptr if dst is not None else None, c0)
        if key not in self._cache:
            self._cache[key] = self._chain(call * self.B, self.B, self.gper, call * self.gper, False, True, False, dst, c0)
        self.ctx.run(self._cache[key])
