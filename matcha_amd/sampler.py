"""Device-side negative sampling (reference main.py:361-459) over an exact hash set (utils.py:75-97 builds Bloom
filters; see matcha_amd/csrc/sampler.hip).  Thin wrappers over the C ABI; torch is used for memory only."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


class HyperedgeSet:
    """Exact membership set of known hyperedges (all sizes in one table).  ``edges``: int64 [M, L] zero-padded,
    each row ascending.  ``HyperedgeSet.empty(device, L)`` reproduces the reference's phase 1, where the
    "dict" is a list of empty python sets (main.py:589) and negatives come out equal to their positives."""

    def __init__(self, edges: torch.Tensor):
        lib = _lib.load()
        if not edges.is_cuda:
            raise _lib.MatchaHipError("HyperedgeSet needs a cuda tensor (no CPU fallback)")
        self.edges = edges.to(torch.long).contiguous()
        self.n, self.L = int(self.edges.shape[0]), int(self.edges.shape[1])
        nbytes = lib.matcha_hashset_bytes(self.n)
        self.table = torch.empty(nbytes, dtype=torch.uint8, device=edges.device)
        st = C.c_void_p(torch.cuda.current_stream(edges.device).cuda_stream)
        _lib.check(lib.matcha_hashset_build(_lib.ptr(self.table), nbytes, _lib.ptr(self.edges), self.n, self.L, st), "matcha_hashset_build")

    @classmethod
    def empty(cls, device, L: int):
        return cls(torch.zeros((0, L), dtype=torch.long, device=device))

    def contains(self, rows: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        rows = rows.to(device=self.edges.device, dtype=torch.long).contiguous()
        out = torch.empty(rows.shape[0], dtype=torch.int32, device=rows.device)
        st = C.c_void_p(torch.cuda.current_stream(rows.device).cuda_stream)
        _lib.check(lib.matcha_hashset_contains(_lib.ptr(self.table), _lib.ptr(self.edges), self.L, _lib.ptr(rows), rows.shape[0],
                                               rows.shape[1], _lib.ptr(out), st), "matcha_hashset_contains")
        return out.bool()


class NegativeSampler:
    """generate_negative (main.py:361-459): ``neg_num`` corrupted copies of every positive -- positions chosen with
    Binomial(k, 1/2) != 0 multiplicity, each replaced by a uniform bin of the SAME chromosome, redrawn until the row
    is duplicate-free, has all adjacent gaps > min_dis and is not a known hyperedge."""

    def __init__(self, hset: HyperedgeSet, node2chrom: np.ndarray, chrom_range: np.ndarray, neg_num: int = 3, min_dis: int = 0,
                 seed: int = 0):
        dev = hset.edges.device
        self.hset, self.neg_num, self.min_dis = hset, int(neg_num), int(min_dis)
        self.node2chrom = torch.as_tensor(np.asarray(node2chrom, dtype=np.int32), device=dev)
        self.chrom_range = torch.as_tensor(np.asarray(chrom_range, dtype=np.int32), device=dev).contiguous()
        self.seed = torch.full((1,), int(seed), dtype=torch.int64, device=dev)
        self.n_nodes = int(self.node2chrom.numel()) - 1
        self.n_chrom = int(self.chrom_range.shape[0])
        # device status word (include/matcha_hip.h): [0] bit 1 = a node without a chromosome was met, [1] = negatives whose
        # trials were exhausted (returned equal to their positive); read it with check_status() at a point that syncs anyway
        self.status = torch.zeros(4, dtype=torch.int32, device=dev)

    def sample_into(self, pos: torch.Tensor, neg_out: torch.Tensor, advance_seed: bool = True):
        """pos int64 [P,L] -> neg_out int64 [P*neg_num, L]; negatives of positive j at rows neg_num*j ... (main.py:383-428).
        Advances the seed on the device (graph-replay safe) unless the caller already did (``advance_seed=False``)."""
        lib = _lib.load()
        P, L = pos.shape
        st = C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
        if advance_seed:
            self.seed.add_(1)
        _lib.check(lib.matcha_neg_sample(_lib.ptr(self.hset.table), _lib.ptr(self.hset.edges), self.hset.n, self.hset.L, _lib.ptr(pos), P, L,
                                         self.neg_num, self.min_dis, _lib.ptr(self.node2chrom), self.n_nodes, _lib.ptr(self.chrom_range),
                                         self.n_chrom, _lib.ptr(self.seed), _lib.ptr(neg_out), _lib.ptr(self.status), st), "matcha_neg_sample")
        return neg_out

    def check_status(self) -> int:
        """Synchronising read of the status word: raises KeyError if a node without a chromosome reached the sampler (the
        reference's ``chrom_range[node2chrom[node]]`` raises there, main.py:401-403); returns the number of negatives whose
        65 536 redraws were exhausted since the last call (the reference would still be looping, main.py:392)."""
        st = self.status.tolist()
        self.status.zero_()
        _lib.raise_on_status(st, "NegativeSampler")
        return int(st[1])

    def sample(self, pos: torch.Tensor) -> torch.Tensor:
        pos = pos.to(torch.long).contiguous()
        out = torch.empty(pos.shape[0] * self.neg_num, pos.shape[1], dtype=torch.long, device=pos.device)
        return self.sample_into(pos, out)
