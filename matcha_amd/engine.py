"""Fused training step of the hyperedge classifier on one MI355X (+ data parallel over RCCL).

One step = what ``train_epoch``'s loop body does in the reference (main.py:155-183) for one batch:
forward (Classifier.forward + weighted BCE, main.py:54-56), ``loss = bce*alpha + recon*beta`` (:166),
backward (:179) and ``AdamW.step`` (:183) -- here as direct C-ABI calls on flat buffers: no autograd graph, one
AdamW launch over every live parameter, gradients zeroed inside the optimizer kernel, and no host
synchronisation (the reference's two ``.item()`` per step, main.py:187-188, are left to the caller), so the whole
step can be captured in a hipGraph (``Trainer.capture``).

Data parallel: one process per GPU; every rank runs the same step on its shard of the batch and
the flat gradient buffer is summed with one RCCL all-reduce (``torch.distributed`` backend "nccl" is RCCL on ROCm)
and scaled by 1/world inside the AdamW kernel; equal shard sizes then reproduce the single-rank step on the global
batch because the loss is a mean over rows (main.py:56).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .Modules import Classifier, _Runtime
from .parallel import allreduce_bucket, broadcast_parameters, recon_grad_weight


class Trainer:
    def __init__(self, model: Classifier, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, process_group=None, base_seed: int = 0):
        self.model = model
        self.rt: _Runtime = model._runtime()
        rt = self.rt
        self.lib = rt.lib
        self.lr, self.betas, self.eps, self.wd = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        dev = rt.device
        # gradients + (data parallel) a float copy of the `touched` flags behind them: ONE all-reduce bucket per step
        self._n_touched = rt.n_touched
        self.gbuf = torch.zeros(rt.n_flat + rt.n_touched, dtype=torch.float32, device=dev)
        self.gflat = self.gbuf[:rt.n_flat]
        self.exp_avg = torch.zeros_like(self.gflat)
        self.exp_avg_sq = torch.zeros_like(self.gflat)
        self.grads = rt.tensors_for(self.gflat)
        n_seg = len(rt.seg_group_list)
        self.n_seg = n_seg
        self.seg_step = torch.zeros(n_seg, dtype=torch.int32, device=dev)
        self.seg_coef = torch.zeros(3 * n_seg, dtype=torch.float32, device=dev)
        self.touched = torch.zeros(rt.n_touched, dtype=torch.int32, device=dev)
        self.losses = torch.zeros(3, dtype=torch.float32, device=dev)     # bce, recon, rows of the recon mean
        self.seed = torch.full((1,), int(base_seed), dtype=torch.int64, device=dev)
        self._ws = {}
        self._logits = {}
        self.pg = process_group
        self.world = 1
        self.force_collectives = False      # bench.py sets it under a 1-rank torchrun launch to exercise the RCCL path
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            broadcast_parameters(rt.flat, 0, process_group)      # every rank starts from rank 0's weights

    # ---- buffers ------------------------------------------------------------------------------------
    def _buffers(self, B: int, L: int):
        key = (B, L)
        if key not in self._ws:
            self._ws[key] = self.rt.workspace(B, L)
            self._logits[key] = torch.empty(B, dtype=torch.float32, device=self.rt.device)
        return self._ws[key], self._logits[key]

    def _opts(self, alpha: float, beta: float, random_chrom: int):
        o = _lib.StepOpts()
        o.training = 1 if self.model.training else 0
        o.p_drop_adj, o.p_drop_fc1, o.p_drop_pff = self.model._dropout_p()
        o.alpha, o.beta = float(alpha), float(beta)
        o.random_chrom = int(random_chrom)
        o.seed = self.seed.data_ptr()
        o.loss_in_forward = 1            # the loss is alpha*bce + beta*recon here: the tail's backward runs inside the forward kernel
        return o

    # ---- one step -------------------------------------------------------------------------------------
    def forward_backward(self, x, y, w, alpha=1.0, beta=0.001, random_chrom: int = 0):
        """forward + backward into the flat gradient buffer (accumulating).  x int64 [B,L]; y, w float [B] or [B,1]."""
        rt = self.rt
        if not rt.still_packed():
            raise _lib.MatchaHipError("model parameters moved after the Trainer was built; create a new Trainer")
        B, L = x.shape
        ws, logits = self._buffers(B, L)
        opts = self._opts(alpha, beta, random_chrom)
        st = rt.stream()
        self.seed.add_(1)                                   # new dropout masks every step (graph-replay safe)
        _lib.check(self.lib.matcha_forward(C.byref(rt.shape), C.byref(rt.params), C.byref(rt.frozen), C.byref(opts), _lib.ptr(x), B, L,
                                           _lib.ptr(y), _lib.ptr(w), _lib.ptr(logits), _lib.ptr(self.losses), _lib.ptr(ws), ws.numel(),
                                           st), "matcha_forward")
        drecon = None
        if (self.world > 1 or self.force_collectives) and rt.shape.mode == 1 and beta != 0.0:
            # adj front end under data parallel: the recon loss is a mean over a rank-dependent number of rows
            drecon = recon_grad_weight(self.losses[2:3], beta, self.pg)
        _lib.check(self.lib.matcha_backward(C.byref(rt.shape), C.byref(rt.params), C.byref(rt.frozen), C.byref(opts), _lib.ptr(x), B, L,
                                            _lib.ptr(y), _lib.ptr(w), None, _lib.ptr(drecon) if drecon is not None else None,
                                            C.byref(self.grads), _lib.ptr(self.touched), _lib.ptr(ws), ws.numel(), st), "matcha_backward")
        return logits

    def all_reduce(self):
        """One RCCL all-reduce per step over [gradients | touched flags] (matcha_amd/parallel.py::allreduce_bucket)."""
        if self.world > 1 or self.force_collectives:
            allreduce_bucket(self.gbuf, self.rt.n_flat, self.touched, self.pg, force=self.force_collectives)

    def optimizer_step(self):
        rt = self.rt
        _lib.check(self.lib.matcha_adamw_step(_lib.ptr(rt.flat), _lib.ptr(self.gflat), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                              rt.n_flat, _lib.ptr(rt.seg_off), self.n_seg, _lib.ptr(rt.seg_group), _lib.ptr(self.touched),
                                              _lib.ptr(self.seg_step), _lib.ptr(self.seg_coef), self.lr, self.betas[0], self.betas[1],
                                              self.eps, self.wd, 1.0 / self.world, rt.stream()), "matcha_adamw_step")

    def step(self, x, y, w, alpha=1.0, beta=0.001, random_chrom: int = 0):
        """One optimisation step.  Returns device tensors (bce [scalar view], recon [1], logits [B]); nothing syncs."""
        x = x.contiguous()
        y = y.reshape(-1).contiguous()
        w = w.reshape(-1).contiguous()
        logits = self.forward_backward(x, y, w, alpha, beta, random_chrom)
        self.all_reduce()
        self.optimizer_step()
        return self.losses[0], self.losses[1:2], logits

    # ---- hipGraph capture of the single-GPU step ---------------------------------------------------------
    def capture(self, x, y, w, alpha=1.0, beta=0.001, random_chrom: int = 0):
        """Capture forward+backward+AdamW on static input buffers; returns a callable that replays the graph.
        Refill ``x``, ``y``, ``w`` in place between replays."""
        if self.world > 1:
            raise RuntimeError("capture() covers the single-GPU step; with DP the all-reduce sits between two graphs")
        x = x.contiguous()
        y = y.reshape(-1).contiguous()
        w = w.reshape(-1).contiguous()
        side = torch.cuda.Stream(self.rt.device)
        side.wait_stream(torch.cuda.current_stream(self.rt.device))
        with torch.cuda.stream(side):
            for _ in range(2):                          # warm up allocations outside the capture
                self.step(x, y, w, alpha, beta, random_chrom)
        torch.cuda.current_stream(self.rt.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.step(x, y, w, alpha, beta, random_chrom)
        self._graph = graph

        def replay():
            graph.replay()
            return out
        return replay
