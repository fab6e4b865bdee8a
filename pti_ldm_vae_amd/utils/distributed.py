"""Process-group setup for the data-parallel loop: one process per MI355X under ``torchrun``.

Mirrors ``src/pti_ldm_vae/utils/distributed.py:8-29`` (``env://`` rendezvous, 10 h timeout, one
barrier, rank := LOCAL_RANK => single node).  Backend ``"nccl"`` on PyTorch-ROCm IS RCCL, running
over xGMI between the GPUs of a node; ``backend="gloo"`` is the CPU test double.
"""
from __future__ import annotations

import os
from datetime import timedelta

import torch
import torch.distributed as dist


def setup_ddp(rank: int, world_size: int, backend: str | None = None):
    """-> (dist, device).  Reads MASTER_ADDR / MASTER_PORT from the environment (``env://``)."""
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if "MASTER_ADDR" not in os.environ:
        raise KeyError("MASTER_ADDR is not set: launch with torchrun (env:// rendezvous)")
    if backend == "nccl":
        torch.cuda.set_device(rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://", timeout=timedelta(seconds=36000), rank=rank,
                                world_size=world_size)
    dist.barrier()
    device = torch.device(f"cuda:{rank}") if backend == "nccl" else torch.device("cpu")
    return dist, device
