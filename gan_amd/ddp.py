"""Data-parallel gradient exchange: one process per GPU, gradients of every network live in ONE flat fp32
buffer per network (ParamSet.grad), so the exchange is a handful of large collectives — sized for xGMI's
point-to-point links rather than many small per-tensor all-reduces.  The reference has no multi-GPU code
(base_gan.py:18-19 only prints the GPU count); semantics are defined here: per-replica
BatchNorm statistics, gradient = mean over ranks of the per-rank mean-loss gradients.

Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, grad_buffers, group=None, compress_bf16=False, max_chunk_elems=64 << 20):
        """grad_buffers: list of flat fp32 tensors (ParamSet.grad of each network)."""
        self.bufs = list(grad_buffers)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.compress = compress_bf16
        self.max_chunk = max_chunk_elems
        self._stage = [torch.empty_like(b, dtype=torch.bfloat16) for b in self.bufs] if compress_bf16 else None

    @property
    def grad_scale(self):
        """Adam consumes SUM-reduced gradients scaled by 1/world (mean over ranks)."""
        return 1.0 / self.world

    def __call__(self):
        if self.world == 1:
            return
        for i, b in enumerate(self.bufs):
            if self.compress:
                s = self._stage[i]
                s.copy_(b)
                self._all_reduce_chunks(s)
                b.copy_(s)
            else:
                self._all_reduce_chunks(b)

    def _all_reduce_chunks(self, t):
        n = t.numel()
        for o in range(0, n, self.max_chunk):
            dist.all_reduce(t[o:min(n, o + self.max_chunk)], op=dist.ReduceOp.SUM, group=self.group)


def shard_batch(global_batch, rank, world):
    """Even split of the global batch; returns (start, count)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    per = global_batch // world
    return rank * per, per
