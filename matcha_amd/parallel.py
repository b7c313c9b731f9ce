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


def allreduce_bucket(gbuf: torch.Tensor, n_flat: int, touched: torch.Tensor, group=None, force: bool = False) -> float:
    """One collective per step: ``gbuf`` = [flat gradients (n_flat floats) | room for the touched flags].  The flags are
    copied behind the gradients as floats, the whole buffer is summed across ranks, and a flag is set again where the sum
    is positive -- "grad is None" is thereby decided on the GLOBAL batch.  Returns the 1/world scale the
    optimizer applies.  ``force`` runs the collective even for a single rank (bench: exercise RCCL under a 1-rank launch)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return 1.0
    tail = gbuf[n_flat:n_flat + touched.numel()]
    tail.copy_(touched)
    dist.all_reduce(gbuf, op=dist.ReduceOp.SUM, group=group)
    touched.copy_(tail > 0)
    return 1.0 / world


def sparse_exchange_pays(n_nodes: int, d: int, cap: int, world: int) -> bool:
    """Row-sparse exchange of the table gradient or dense all-reduce?  Bytes a rank receives: the
    all-gather of (id, row) lists is (world - 1) * cap * (d + 1) * 4; a ring all-reduce of the dense [N + 1, d] gradient moves
    2 * (world - 1) / world * (N + 1) * d * 4.  hg38 1 Mb (N = 3067) stays dense (1.4 MB against 0.6 GB of lists at 65 536
    rows per rank); BASELINE config 5 (1 M nodes, d = 256) goes sparse (0.94 GB of lists at 16 384 x 8 slots per rank on 8
    ranks against 1.8 GB)."""
    if world <= 1:
        return False
    return (world - 1) * cap * (d + 1) < 2 * (world - 1) / world * (n_nodes + 1) * d


def exchange_table_rows(ids: torch.Tensor, rows: torch.Tensor, group=None, out=None):
    """All-gather every rank's (node id, gradient row) list: ids int32 [n] (0 = unused entry), rows float [n, d] ->
    (ids_all [world * n], rows_all [world * n, d]) in rank order on every rank; ``out`` = preallocated (ids_all, rows_all) to
    receive into (the Trainer reuses one pair across steps).  Every rank passes the same ``n`` (the Trainer agrees on
    max_r(real tokens) + 1 beforehand, so the lists travel without their padding); xGMI is point-to-point, and an all-gather
    sends each list once to each peer (one hop), where a ring all-reduce of the dense table would forward
    2 (world - 1) / world table copies."""
    world = dist.get_world_size(group)
    if out is None:
        out = (ids.new_empty(world * ids.numel()), rows.new_empty((world * rows.shape[0], rows.shape[1])))
    ids_all, rows_all = out
    dist.all_gather_into_tensor(ids_all, ids.contiguous(), group=group)
    dist.all_gather_into_tensor(rows_all, rows.contiguous(), group=group)
    return ids_all, rows_all


def recon_grad_weight(m_local: torch.Tensor, beta: float, group=None) -> torch.Tensor:
    """Upstream gradient of the reconstruction loss for THIS rank so that the averaged gradient equals the single-rank
    one on the global batch.  The reference's recon loss is a mean over the m "other" tokens of the
    drawn chromosome (Modules.py:195-199); with m_r rows on rank r the global mean is sum_r(m_r * loss_r) / sum_r(m_r),
    so rank r contributes with weight  beta * m_r / sum(m) * world  (the optimizer later divides the summed gradient by
    world).  ``m_local`` is the 1-element device tensor ``losses[2:3]`` written by matcha_forward; no host sync."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return m_local.new_full((1,), float(beta))
    world = dist.get_world_size(group)
    total = m_local.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return float(beta) * world * m_local / torch.clamp(total, min=1.0)


def broadcast_parameters(flat: torch.Tensor, src: int = 0, group=None):
    """Replicate rank ``src``'s flat parameter buffer (call once after building the model)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
