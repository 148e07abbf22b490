"""Data-parallel gradient exchange: one process per GPU; the gradients of every network live in ONE flat fp32
buffer per network (ParamSet.grad: conv kernels first, in layer order, then the norm/bias vectors), so a BUCKET is
a contiguous range of that buffer.  The reference has no multi-GPU code (base_gan.py:18-19 only prints the GPU
count); semantics are defined here: per-replica BatchNorm statistics, gradient = mean over ranks
of the per-rank mean-loss gradients.

Exchange = one RCCL all-reduce per bucket (a few large collectives sized for xGMI's point-to-point links, not one per
tensor), started as soon as the bucket's last wgrad GEMM has been enqueued and overlapped with the rest of the
backward pass; Adam runs per bucket as it lands (gan_amd/steps.py).  Wire format: bf16 (57 M fp32 gradients =
229 MB per Pix2Pix step, 114 MB on the wire) written by our own cast kernel (gan_grad_pack, captured in the step's
graphs) and read back by gan_grad_unpack, which also applies the 1/world of the mean; fp32 exchanges in place.

`start(i, lo, hi)` enqueues the all-reduce of elements [lo, hi) of buffer i on the communicator's stream (after the
work already queued on the current stream) and returns a handle; `wait(h)` makes the CURRENT stream wait for it
(the host never blocks).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, grad_buffers, group=None, compress_bf16=False, lib=None, rehearse=False, exchange='allreduce'):
        """grad_buffers: list of flat fp32 tensors (ParamSet.grad of each network).  lib: the loaded C-ABI library
        (needed for the bf16 wire format on the GPU; CPU/gloo tests pass None and get a torch cast).  rehearse: issue the
        collectives even in a group of ONE rank (a one-GPU box can then run the whole data-parallel schedule over RCCL).
        exchange: 'allreduce' - one all-reduce per bucket in the wire format (bf16 wire: the SUM over ranks is taken in bf16);
        'rs_ag' - reduce-scatter of the fp32 gradients (fp32 accumulation over the ranks, each rank reduces 1/world of the
        bucket: the direct exchange on a fully connected xGMI mesh) followed by an all-gather of the reduced
        shards in the wire format (bf16 wire: 6 bytes per element on the links instead of 4, one rounding instead of
        world - 1)."""
        self.bufs = list(grad_buffers)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.active = self.world > 1 or (bool(rehearse) and dist.is_initialized())
        self.compress = bool(compress_bf16)
        self.lib = lib
        self.wire = [torch.zeros_like(b, dtype=torch.bfloat16) for b in self.bufs] if self.compress else None
        if exchange not in ('allreduce', 'rs_ag'):
            raise ValueError(f"exchange {exchange!r}: 'allreduce' or 'rs_ag'")
        self.exchange = exchange
        self._has_rs = dist.is_initialized() and dist.get_backend(group) != 'gloo'     # gloo has no reduce-scatter (CPU tests)
        self._pending = []

    @property
    def grad_scale(self):
        """What Adam must multiply the exchanged gradients by: fp32 exchanges are SUMs (1/world applied by Adam's
        grad_scale); bf16 exchanges come back through unpack(), which already applied it."""
        return 1.0 if self.compress else 1.0 / self.world

    # ---- wire-format kernels (enqueue-only; capturable) ----------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.bufs[0].device).cuda_stream if self.bufs[0].is_cuda else None

    def pack(self, i, lo=0, hi=None):
        """grad[i][lo:hi] (fp32) -> wire[i][lo:hi] (bf16).  No-op for fp32 exchanges and for 'rs_ag' (which reduces the fp32
        gradients and casts its own reduced shard inside start())."""
        if not self.compress or (self.exchange == 'rs_ag' and self.active):
            return
        hi = self.bufs[i].numel() if hi is None else hi
        if self.lib is not None and self.bufs[i].is_cuda:
            rc = self.lib.gan_grad_pack(self.bufs[i].data_ptr() + 4 * lo, self.wire[i].data_ptr() + 2 * lo, hi - lo, self._stream())
            if rc:
                raise RuntimeError(f"gan_grad_pack failed rc={rc}")
        else:
            self.wire[i][lo:hi].copy_(self.bufs[i][lo:hi])

    def unpack(self, i, lo=0, hi=None):
        """wire[i][lo:hi] (bf16, summed over ranks) * 1/world -> grad[i][lo:hi] (fp32).  No-op for fp32 exchanges."""
        if not self.compress:
            return
        hi = self.bufs[i].numel() if hi is None else hi
        if self.lib is not None and self.bufs[i].is_cuda:
            rc = self.lib.gan_grad_unpack(self.wire[i].data_ptr() + 2 * lo, self.bufs[i].data_ptr() + 4 * lo, hi - lo,
                                          1.0 / self.world, self._stream())
            if rc:
                raise RuntimeError(f"gan_grad_unpack failed rc={rc}")
        else:
            self.bufs[i][lo:hi].copy_(self.wire[i][lo:hi].float() / self.world)

    # ---- collectives -----------------------------------------------------------------------------------
    def start(self, i, lo=0, hi=None):
        """Begin the all-reduce (SUM) of elements [lo, hi) of buffer i in its wire format; returns a handle."""
        if not self.active:
            return None
        t = self.wire[i] if self.compress else self.bufs[i]
        hi = t.numel() if hi is None else hi
        if self.exchange == 'rs_ag':
            return self._start_rs_ag(i, lo, hi)
        return dist.all_reduce(t[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _start_rs_ag(self, i, lo, hi):
        """fp32 reduce-scatter (in place: rank r's shard of the bucket ends up holding the SUM over ranks), cast of that shard to
        the wire format, all-gather (in place in the wire / fp32 buffer).  Runs on the CURRENT stream's order: the caller issues
        it on the communicator launcher stream, which then waits for the reduce-scatter before the cast kernel."""
        n, w = hi - lo, self.world
        if n % (8 * w):
            raise ValueError(f"bucket of {n} elements: 'rs_ag' needs a multiple of {8 * w}")
        per = n // w
        src = self.bufs[i][lo:hi]
        mine = src[self.rank * per:(self.rank + 1) * per]
        if self._has_rs:
            h = dist.reduce_scatter_tensor(mine, src, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:           # gloo (CPU tests / one-GPU rehearsals of the schedule): the same sums through an all-reduce
            h = dist.all_reduce(src, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h.wait()
        if self.compress:
            a = lo + self.rank * per
            self.exchange, keep = 'allreduce', self.exchange          # (pack() is a no-op under 'rs_ag')
            try:
                self.pack(i, a, a + per)
            finally:
                self.exchange = keep
            out = self.wire[i][lo:hi]
        else:
            out = src
        return dist.all_gather_into_tensor(out, out[self.rank * per:(self.rank + 1) * per], group=self.group, async_op=True)

    def wait(self, handle):
        """The current stream waits for the collective (asynchronous for the host with RCCL)."""
        if handle is not None:
            handle.wait()

    # ---- whole-buffer convenience (eager path; CycleGAN's per-network exchange) --------------------------
    def start_all(self, i):
        self.pack(i)
        self._pending.append((i, self.start(i)))

    def finish(self, unpack=True):
        """Wait for the pending whole-buffer exchanges; unpack=False leaves the result in the wire buffers (Adam reads it
        there, gan_amd/steps.py) instead of casting it back into the fp32 gradient buffers."""
        for i, h in self._pending:
            self.wait(h)
            if unpack:
                self.unpack(i)
        self._pending = []

    def __call__(self, unpack=True):
        for i in range(len(self.bufs)):
            self.start_all(i)
        self.finish(unpack)


def shard_batch(global_batch, rank, world):
    """Even split of the global batch; returns (start, count)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    per = global_batch // world
    return rank * per, per


# ---- data-parallel runs of the CLIs (torchrun / torch.distributed.run: one process per GPU) ------------------------------
class DistInfo:
    """rank / world of this process and the device it owns; world == 1: a plain single-process run."""

    def __init__(self, rank=0, world=1, device=None):
        self.rank, self.world, self.device = rank, world, device

    @property
    def is_main(self):
        return self.rank == 0


def init_from_env(device: str = 'cuda:0', backend: str | None = None) -> DistInfo:
    """Join the process group a launcher described in the environment (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR/PORT) BEFORE
    the first GPU call of this process; each rank owns cuda:LOCAL_RANK.  Without a launcher: DistInfo(0, 1, device).
    backend: "nccl" (= RCCL on ROCm) unless given; CPU tests and one-GPU rehearsals pass "gloo"."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return DistInfo(0, 1, device)
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    backend = backend or 'nccl'
    ndev = torch.cuda.device_count()              # (does not create a GPU context)
    dev = f'cuda:{local % max(ndev, 1)}'
    kw = {}
    if backend == 'nccl':
        kw['device_id'] = torch.device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    if torch.cuda.is_available():
        torch.cuda.set_device(torch.device(dev))
    return DistInfo(rank, world, dev)


def shard_files(files, rank, world):
    """This rank's share of a file list (after the seeded split, which every rank computes identically): every world-th
    file, all shards cut to the same length so that every rank runs the same number of steps (each step is a collective)."""
    per = len(files) // world
    return list(files[rank:per * world:world]) if world > 1 else list(files)


def mean_over_ranks(acc, n, info: DistInfo):
    """Epoch mean of the per-step loss vectors over all ranks' steps: (sum over ranks of acc) / (sum over ranks of n)."""
    if info is None or info.world == 1 or acc is None:
        return (acc / n) if n else None
    t = torch.cat([acc.float().flatten(), torch.tensor([float(n)], device=acc.device)])
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t[:-1] / t[-1]


def assert_replicas_in_sync(param_sets, info: DistInfo):
    """Every rank must hold bit-identical weights after a data-parallel run (same start, same exchanged gradients): compares
    a checksum of every network's master weights across the ranks; raises on the first difference."""
    if info is None or info.world == 1:
        return
    sums = torch.stack([torch.stack([ps.master.double().sum(), ps.master.double().abs().sum()]) for ps in param_sets]).flatten()
    lo, hi = sums.clone(), sums.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"data-parallel replicas diverged: weight checksums differ across ranks ({lo.tolist()} .. {hi.tolist()})")


def shutdown(info: DistInfo, failed: bool = False):
    """Leave the process group.  Success path: together, behind a barrier (a rank that exits early makes its peers' pending
    collectives fail).  failed=True (this rank is unwinding an exception): NO barrier - its peers sit in a gradient exchange of
    another size and a mismatched collective hangs RCCL until the watchdog fires; the group is torn down (aborted where the
    backend can) and the exception propagates, so the launcher sees a non-zero exit and ends the other ranks."""
    if info is None or info.world <= 1 or not dist.is_initialized():
        return
    if failed:
        abort = getattr(dist.distributed_c10d, '_abort_process_group', None)     # (torch >= 2.6: ncclCommAbort, does not wait for peers)
        try:
            if abort is None:
                raise RuntimeError("torch.distributed has no _abort_process_group")
            abort()
        except Exception as e:        # nothing was torn down: say so (the launcher / the watchdog ends the peers, not this rank)
            import sys
            print(f"gan_amd.ddp: could not abort the process group on a failing rank ({e!r}); its peers block in their pending "
                  f"collectives until the launcher or the RCCL watchdog ends them", file=sys.stderr, flush=True)
        return
    try:
        dist.barrier()
    finally:
        dist.destroy_process_group()
