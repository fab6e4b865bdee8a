"""Data-parallel gradient exchange for the VAE training loop (the reference wraps the model in
``DistributedDataParallel``: ``vae_scripts/train_vae.py:282``; collectives C3/C4 of SURVEY.md §2.1).

Per-image forward/backward are independent (GroupNorm is per sample, attention per image), so the
batch shards across the GPUs of a node and ONE exchange happens per optimiser step: an fp32 SUM
all-reduce of all gradients, averaged by the optimiser (``grad_scale = 1/world``).  Because the
kernels write gradients into one flat arena in layer order, the exchange is a handful of large
contiguous buckets launched WHILE backward is still running: ``ready(start, end)`` is called by the
engine as each block's gradients become final (decoder tail first), full buckets go out with
``async_op=True`` (RCCL runs them on its own HIP stream, chained to the compute stream by events)
and ``finish()`` joins them before Adam.  There is no unused-parameter search (C5): every parameter
is used every step.  ``backend="nccl"`` is RCCL over xGMI; ``gloo`` is the CPU test double.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGradAllReducer:
    def __init__(self, grad_arena: torch.Tensor, group=None, bucket_bytes: int = 4 << 20):
        self.arena = grad_arena
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._lo = self._hi = None
        self._works = []
        self.launched = []          # (start, end) of every bucket sent this step, for tests/diagnostics

    def begin_step(self):
        self._lo = self._hi = None
        self._works.clear()
        self.launched.clear()

    def ready(self, start: int, end: int):
        """Gradients in arena[start:end] are final.  Ranges arrive in DESCENDING, adjacent order within
        a region (backward walks the layers in reverse); a gap flushes the pending bucket."""
        if self.world == 1 or end <= start:
            return
        if self._lo is None:
            self._lo, self._hi = start, end
        elif end == self._lo:
            self._lo = start
        elif start == self._hi:
            self._hi = end
        else:
            self._flush()
            self._lo, self._hi = start, end
        if self._hi - self._lo >= self.bucket_elems:
            self._flush()

    def _flush(self):
        if self._lo is None:
            return
        s, e = self._lo, self._hi
        self._lo = self._hi = None
        self.launched.append((s, e))
        self._works.append(dist.all_reduce(self.arena[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduce_range(self, start: int, end: int):
        """All-reduce arena[start:end] now, as ~bucket-sized asynchronous collectives (descending, like backward emits
        them).  Used by the HIP-graph step mode: a whole region's gradients are final when its captured half has been
        replayed, so there is nothing to coalesce -- the region goes out in ``bucket_bytes`` pieces that overlap the
        next captured half."""
        if self.world == 1 or end <= start:
            return
        self._flush()
        hi = end
        while hi > start:
            lo = max(start, hi - self.bucket_elems)
            if lo - start < self.bucket_elems // 4:      # do not leave a sliver as its own collective
                lo = start
            self.launched.append((lo, hi))
            self._works.append(dist.all_reduce(self.arena[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            hi = lo

    def finish(self):
        """Send what is pending and make the current stream wait for every bucket."""
        if self.world == 1:
            return
        self._flush()
        for w in self._works:
            w.wait()
        self._works.clear()


def broadcast_parameters(param_arena: torch.Tensor, group=None, src: int = 0):
    """DDP's construction-time ``_sync_module_states`` (C3) as one broadcast of the flat arena."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(param_arena, src=src, group=group)
