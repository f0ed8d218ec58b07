"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference has no distributed code at all; the scheme here is the natural sharding of its hot path
(SURVEY.md §8e): every rank owns its envs, its HBM replay shard and its minibatch; parameters and optimizer
state are replicated; the ONLY exchange is one sum all-reduce of the flat f32 gradient per learn()
(4*P bytes = 333 KB at A=6), after which every rank applies the identical clip+Adam+Polyak with the 1/W folded
into the clip scale, so replicas stay in lock-step. BatchNorm statistics stay per rank (the reference's
per-learner semantics at per-GPU batch size).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """(rank, local_rank, world). Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def rank_seed(seed: int, rank: int, stream: int = 0) -> int:
    """Independent per-rank seeds for env / noise / sampler streams; rank 0, stream 0 keeps the user's seed so a
    1-GPU run is the reference-seeded run."""
    return (int(seed) + 7919 * int(rank) + 104729 * int(stream)) & 0xFFFFFFFFFFFFFFFF


def all_reduce_flat_grad(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce of the flat gradient. The mean is NOT taken here: naf_adam_polyak_fused multiplies by
    inv_world inside its clip scale, so the averaged gradient never makes a separate pass over memory."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def broadcast_parameters(theta2: torch.Tensor, src: int = 0, group=None) -> None:
    """Same seed already gives identical initial weights on every rank; broadcast anyway so a loaded checkpoint or a
    stray RNG draw cannot desynchronise the replicas."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(theta2, src=src, group=group)


def units_per_rank(total_envs: int, world: int, rank: int) -> int:
    """Weak scaling keeps envs/GPU fixed; for a fixed global env count the remainder goes to the low ranks."""
    base, rem = divmod(int(total_envs), int(world))
    return base + (1 if rank < rem else 0)
