"""Data-parallel plumbing for the hyperedge-classifier step: one process per GPU, the positive
hyperedges sharded by rank, every parameter replicated, ONE all-reduce of the flat gradient buffer per step
(RCCL over xGMI; ``torch.distributed`` backend "nccl" on ROCm, "gloo" in the CPU tests).

Why this reproduces the single-rank step on the global batch: rows are independent inside a step and the loss is
a mean over rows (main.py:56), so with equal shard sizes the global gradient is the average of the rank
gradients; tensors that received no gradient on ANY rank stay "grad None" (skipped by AdamW), which is decided on
the global batch by a MAX all-reduce of the ``touched`` flags.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world: int) -> np.ndarray:
    """Indices of the rows rank ``rank`` owns: a strided shard of a (pre-shuffled) list, truncated so that every
    rank owns the same number of rows (equal shards are what makes mean-of-means == global mean)."""
    per = n_rows // world
    return np.arange(rank, per * world, world, dtype=np.int64)


def shard_edges(edges: np.ndarray, weights: np.ndarray, rank: int, world: int):
    idx = shard_rows(len(edges), rank, world)
    return edges[idx], weights[idx]


def allreduce_gradients(gflat: torch.Tensor, touched: Optional[torch.Tensor], group=None) -> float:
    """SUM the flat gradient buffer and MAX the touched flags across ranks (in place).  Returns the scale
    (1/world) the optimizer must apply to the summed gradient."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world == 1:
        return 1.0
    dist.all_reduce(gflat, op=dist.ReduceOp.SUM, group=group)
    if touched is not None:
        dist.all_reduce(touched, op=dist.ReduceOp.MAX, group=group)
    return 1.0 / world


def broadcast_parameters(flat: torch.Tensor, src: int = 0, group=None):
    """Replicate rank ``src``'s flat parameter buffer (call once after building the model)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
