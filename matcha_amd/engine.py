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
import os
from typing import Optional

import torch

from . import _lib
from .Modules import Classifier, _Runtime
from .parallel import allreduce_bucket, broadcast_parameters, exchange_table_rows, recon_grad_weight, sparse_exchange_pays


class Trainer:
    def __init__(self, model: Classifier, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, process_group=None, base_seed: int = 0, table_exchange: str = "auto",
                 deterministic: bool = False):
        """``table_exchange`` (data parallel, table front end): "dense" = the table gradient rides in the flat all-reduce bucket,
        "sparse" = all-gather of per-token (id, row) lists + a deterministic local sum, "auto" = whichever
        moves fewer bytes for the batch shape (parallel.sparse_exchange_pays).
        ``deterministic`` (table front end): the embedding backward sorts the tokens by node id and sums each node's rows in
        token order (one writer per table row) instead of float atomics -- every parameter is then bitwise reproducible from
        run to run; costs ~60 us per 65 536-row step (2 %).  The row-sparse exchange always reduces this way."""
        self.deterministic = bool(deterministic)
        if table_exchange not in ("auto", "dense", "sparse"):
            raise ValueError("table_exchange must be 'auto', 'dense' or 'sparse'")
        self.table_exchange = table_exchange
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
        self.loss_in_forward = True               # the tail's backward inside the forward kernel (opts.loss_in_forward); tests set False for the separate kernels
        self.seed_advanced_by_caller = False      # train.py's captured step advances self.seed inside matcha_step_select (one launch less)
        self._ws = {}
        self._logits = {}
        self._xws = {}
        self.pg = process_group
        self.world = 1
        rank = 0
        self.force_collectives = False      # bench.py sets it under a 1-rank torchrun launch to exercise the RCCL path
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            rank = torch.distributed.get_rank(process_group)
            broadcast_parameters(rt.flat, 0, process_group)      # every rank starts from rank 0's weights
        # dropout masks are a function of (seed, row slot): ranks must not share a seed or every rank would drop the same
        # units of its own rows (rank folded in here so that callers cannot forget it)
        self.seed = torch.full((1,), (int(base_seed) + 0x9E3779B97F4A7C15 * rank) & 0x7FFFFFFFFFFFFFFF, dtype=torch.int64, device=dev)
        # table front end: the table is the FIRST field of the flat buffer; with the row-sparse exchange the all-reduce bucket
        # starts behind it
        self._n_table = (rt.field_off_after("table") if rt.mode == 0 else 0)
        self._sparse = False
        # exchange options (A/B in tests): overlap the encoder part of the bucket with the front-end backward; exchange the
        # row-sparse lists compacted to the real tokens
        self.overlap_exchange = True
        self.compact_exchange = True
        self._side = None
        self._enc_work = None
        self.comm_bytes = {}              # bytes each collective of the last step moved per rank (bench.py reports them)

    # ---- buffers ------------------------------------------------------------------------------------
    def _buffers(self, B: int, L: int):
        key = (B, L)
        if key not in self._ws:
            self._ws[key] = self.rt.workspace(B, L)
            if os.environ.get("MATCHA_POISON_WS"):       # development: a read of workspace bytes nobody wrote shows up as NaN / garbage
                self._ws[key].fill_(0xFF if os.environ["MATCHA_POISON_WS"] == "nan" else 0x5B)
            self._logits[key] = torch.empty(B, dtype=torch.float32, device=self.rt.device)
        return self._ws[key], self._logits[key]

    def _opts(self, alpha: float, beta: float, random_chrom):
        o = _lib.StepOpts()
        o.training = 1 if self.model.training else 0
        o.p_drop_adj, o.p_drop_fc1, o.p_drop_pff = self.model._dropout_p()
        o.alpha, o.beta = float(alpha), float(beta)
        if isinstance(random_chrom, torch.Tensor):
            # a device int32 cell: the kernels read the chromosome when they run, so a captured step replays with whatever the
            # caller wrote there last (matcha_step_opts.random_chrom_dev)
            if random_chrom.dtype != torch.int32 or not random_chrom.is_cuda or random_chrom.numel() != 1:
                raise ValueError("random_chrom as a tensor must be ONE int32 on the model's device")
            o.random_chrom = 0
            if self.rt.mode == 1:
                o.random_chrom_dev = random_chrom.data_ptr()
        else:
            o.random_chrom = int(random_chrom)
        o.seed = self.seed.data_ptr()
        o.loss_in_forward = 1 if self.loss_in_forward else 0      # the loss is alpha*bce + beta*recon here: the tail's backward can run inside the forward kernel
        o.status = self.rt.status.data_ptr()
        o.sparse_table_grad = 1 if self._sparse else 0
        o.deterministic = 1 if self.deterministic else 0
        return o

    def _use_sparse(self, B: int, L: int) -> bool:
        rt = self.rt
        if rt.mode != 0 or not (self.world > 1 or self.force_collectives):
            return False
        if self.table_exchange != "auto":
            return self.table_exchange == "sparse"
        return sparse_exchange_pays(rt.n_nodes, rt.d, B * L + 1, self.world)

    def check_status(self):
        """Synchronising check of the device status word: raises IndexError if a node id outside [0, N] reached a step since
        the last check (the kernels read such ids as the padding id; the reference's nn.Embedding raises at once).  Call it
        where the loop synchronises anyway (end of an epoch: train.train_epoch does)."""
        self.rt.check_status("Trainer.step")

    # ---- one step -------------------------------------------------------------------------------------
    def forward_backward(self, x, y, w, alpha=1.0, beta=0.001, random_chrom=0, max_tokens: Optional[int] = None):
        """forward + backward into the flat gradient buffer (accumulating).  x int64 [B,L]; y, w float [B] or [B,1].
        ``max_tokens`` (row-sparse exchange): the largest number of real tokens (x != 0) any rank holds this step, when the caller knows
        it on the host -- every rank must pass the same value -- so that the lists travel compacted WITHOUT the device-side count
        exchange and its host wait (all_reduce).  The driver knows it: every rank holds the whole global batch on the host."""
        rt = self.rt
        if not rt.still_packed():
            raise _lib.MatchaHipError("model parameters moved after the Trainer was built; create a new Trainer")
        B, L = x.shape
        ws, logits = self._buffers(B, L)
        self._sparse = self._use_sparse(B, L)
        self._last_shape = (B, L)
        opts = self._opts(alpha, beta, random_chrom)
        st = rt.stream()
        self.comm_bytes = {}
        self._host_max_tokens = None if max_tokens is None else int(max_tokens)
        self._begin_exchange(x)
        overlap = self._overlap()
        if overlap:
            opts.encoder_done_event = self._enc_event.cuda_event
        if not self.seed_advanced_by_caller:
            self.seed.add_(1)                               # new dropout masks every step (graph-replay safe)
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
        if overlap:
            self._start_encoder_allreduce()
        return logits

    # ---- data-parallel exchange ----------------------------------------------------------------------
    def _exchange_setup(self):
        """Side stream + event of the overlapped exchange (table front end): the encoder part of the flat bucket (everything from
        ln_q_g on, the contiguous tail) is all-reduced on ``_side`` as soon as matcha_backward records ``_enc_event``, while the
        front-end backward, the embedding scatter / table-gradient sort still run on the main stream."""
        if getattr(self, "_side", None) is None:
            dev = self.rt.device
            self._side = torch.cuda.Stream(dev)
            self._enc_event = torch.cuda.Event()
            self._enc_event.record(torch.cuda.current_stream(dev))       # creates the hipEvent_t behind the handle
            self._cnt_event = torch.cuda.Event()
            self._enc_work = None
            self._cnt_host = torch.zeros(self.world, dtype=torch.int32).pin_memory()
            self._n_enc_off = self.rt.field_off["ln_q_g"]

    def _overlap(self) -> bool:
        """Overlapped two-part bucket: table front end only (there the touched flags are constant, so nothing rides behind the
        gradients), and not while a hipGraph is being captured."""
        return (self.overlap_exchange and self.rt.mode == 0 and (self.world > 1 or self.force_collectives)
                and not torch.cuda.is_current_stream_capturing())

    def _begin_exchange(self, x):
        """Called before matcha_forward.  Row-sparse exchange: start the all-gather of the ranks' real-token counts on the side
        stream (tokens = non-zero slots of x: the plan's Tr, known long before the backward finishes), so that the lists can be
        exchanged COMPACTED to max_r(Tr_r) + 1 entries instead of the fixed capacity B*L + 1 -- the padding fraction (38 % at k in
        {2..8}, L = 8) never crosses xGMI -- without stalling the host: by the time the backward is enqueued the 4-byte counts
        are in pinned memory."""
        if not (self.world > 1 or self.force_collectives):
            return
        self._exchange_setup()
        if self._sparse and self.compact_exchange and self._host_max_tokens is None:
            main = torch.cuda.current_stream(self.rt.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                cnt = (x != 0).sum().to(torch.int32).view(1)
                if self.world > 1:
                    cnt_all = torch.empty(self.world, dtype=torch.int32, device=self.rt.device)
                    torch.distributed.all_gather_into_tensor(cnt_all, cnt, group=self.pg)
                else:
                    cnt_all = cnt
                self._cnt_host.copy_(cnt_all, non_blocking=True)
                self._cnt_event.record(self._side)
            x.record_stream(self._side)

    def _start_encoder_allreduce(self):
        """Right after matcha_backward was enqueued: all-reduce gflat[ln_q_g:] on the side stream behind the encoder_done event."""
        rt = self.rt
        self._side.wait_event(self._enc_event)
        with torch.cuda.stream(self._side):
            part = self.gflat[self._n_enc_off:rt.n_flat]
            self._enc_work = torch.distributed.all_reduce(part, op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=True)
        self.comm_bytes["encoder_allreduce"] = 4 * (rt.n_flat - self._n_enc_off)

    def _timed(self, name: str, fn):
        """bench.py --gpus N: HIP events around one collective on the stream it runs on (self.time_collectives); read back with
        collective_ms() after a synchronisation."""
        if not getattr(self, "time_collectives", False):
            return fn()
        st = torch.cuda.current_stream(self.rt.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        out = fn()
        e1.record(st)
        self.__dict__.setdefault("_coll_events", []).append((name, e0, e1))
        return out

    def collective_ms(self):
        """{collective: [ms per call]} of the timed collectives since the last call (synchronises)."""
        torch.cuda.synchronize(self.rt.device)
        out = {}
        for name, e0, e1 in self.__dict__.pop("_coll_events", []):
            out.setdefault(name, []).append(e0.elapsed_time(e1))
        return out

    def all_reduce(self):
        """Gradient exchange of one step (matcha_amd/parallel.py).  Dense: the flat bucket [gradients | touched flags] in ONE RCCL
        all-reduce -- or, with the table front end, in two parts: the encoder tail already in flight on the side stream
        (``_start_encoder_allreduce``) and [table | attribute_nn | next_w] here.  Row-sparse: the bucket starts BEHIND the table
        gradient, and the table's (id, row) lists are all-gathered (compacted to the largest real-token count of any rank) and
        summed locally in a fixed order (matcha_scatter_rows): every rank ends with the same dense table gradient."""
        if not (self.world > 1 or self.force_collectives):
            return
        rt = self.rt
        dev = rt.device
        lo = self._n_table if self._sparse else 0
        if self._enc_work is not None:
            # table mode: both touched flags are set by every rank's backward -- nothing to exchange behind the gradients
            front = self.gflat[lo:self._n_enc_off]
            if front.numel():
                self._timed("front_allreduce", lambda: torch.distributed.all_reduce(front, op=torch.distributed.ReduceOp.SUM, group=self.pg))
            self.comm_bytes["front_allreduce"] = 4 * front.numel()
            self._enc_work.wait()                                          # main stream waits for the side stream's collective
            torch.cuda.current_stream(dev).wait_stream(self._side)
            self._enc_work = None
        else:
            self._timed("bucket_allreduce", lambda: allreduce_bucket(self.gbuf[lo:], rt.n_flat - lo, self.touched, self.pg, force=self.force_collectives))
            self.comm_bytes["bucket_allreduce"] = 4 * (rt.n_flat - lo + self._n_touched)
        if not self._sparse:
            return
        B, L = self._last_shape
        ws = self._ws[(B, L)]
        p_ids, p_rows, p_n, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        _lib.check(self.lib.matcha_table_grad_rows(C.byref(rt.shape), B, L, _lib.ptr(ws), ws.numel(), C.byref(p_ids), C.byref(p_rows),
                                                   C.byref(p_n), C.byref(cap)), "matcha_table_grad_rows")
        cap, d, base = int(cap.value), rt.d, ws.data_ptr()
        n_send = cap
        if self.compact_exchange and self._host_max_tokens is not None:
            n_send = min(cap, (self._host_max_tokens + 1 + 63) // 64 * 64)        # the caller knew the count: no device read-back, no host wait
        elif self.compact_exchange:
            # the ranks' counts were all-gathered on the side stream at the start of the step; the host waits for that 4-byte copy -- i.e.
            # it cannot run more than one step ahead of the GPU in this mode (pass max_tokens to avoid it)
            self._cnt_event.synchronize()
            n_send = min(cap, (int(self._cnt_host.max()) + 1 + 63) // 64 * 64)   # real tokens of the fullest rank + the padding token
        ids = ws[p_ids.value - base:p_ids.value - base + 4 * n_send].view(torch.int32)
        rows = ws[p_rows.value - base:p_rows.value - base + 4 * n_send * d].view(torch.float32).view(n_send, d)
        key = (cap, d)
        if key not in self._xws:          # receive buffers + sort scratch, sized once for the full capacity and reused every step
            world = self.world
            self._xws[key] = (torch.empty(world * cap, dtype=torch.int32, device=dev), torch.empty(world * cap * d, dtype=torch.float32, device=dev),
                              torch.empty(self.lib.matcha_scatter_rows_workspace_bytes(world * cap, d, rt.n_nodes), dtype=torch.uint8, device=dev))
        ids_buf, rows_buf, xws = self._xws[key]
        n = self.world * n_send
        ids_all, rows_all = self._timed("table_rows_allgather", lambda: exchange_table_rows(ids, rows, self.pg, out=(ids_buf[:n], rows_buf[:n * d].view(n, d))))
        self.comm_bytes["table_rows_allgather"] = 4 * (self.world - 1) * n_send * (d + 1)
        self.comm_bytes["table_rows_fill"] = n_send / cap
        _lib.check(self.lib.matcha_scatter_rows(_lib.ptr(ids_all), _lib.ptr(rows_all), n, d, rt.n_nodes, _lib.ptr(self.gflat), _lib.ptr(xws),
                                                xws.numel(), rt.stream()), "matcha_scatter_rows")

    def eval_forward(self, x, y, w, random_chrom=0):
        """Forward + weighted BCE + reconstruction loss only (eval_epoch, main.py:200-258): no dropout, nothing kept for a backward, the
        compact forward workspace.  Returns the logits [B]; ``self.losses`` holds (bce, recon, rows of the recon mean).  Nothing
        synchronises and nothing allocates after the first call of a shape, so the call can be captured in a hipGraph (train.py)."""
        rt = self.rt
        if not rt.still_packed():
            raise _lib.MatchaHipError("model parameters moved after the Trainer was built; create a new Trainer")
        B, L = x.shape
        key = (B, L, "eval")
        if key not in self._ws:
            self._ws[key] = rt.workspace(B, L, forward_only=True)
            self._logits[key] = torch.empty(B, dtype=torch.float32, device=rt.device)
        ws, logits = self._ws[key], self._logits[key]
        opts = self._opts(1.0, 1.0, random_chrom)
        opts.training, opts.forward_only, opts.loss_in_forward = 0, 1, 0
        opts.sparse_table_grad, opts.deterministic = 0, 0
        _lib.check(self.lib.matcha_forward(C.byref(rt.shape), C.byref(rt.params), C.byref(rt.frozen), C.byref(opts), _lib.ptr(x), B, L,
                                           _lib.ptr(y), _lib.ptr(w), _lib.ptr(logits), _lib.ptr(self.losses), _lib.ptr(ws), ws.numel(),
                                           rt.stream()), "matcha_forward")
        return logits

    def optimizer_step(self):
        rt = self.rt
        _lib.check(self.lib.matcha_adamw_step(_lib.ptr(rt.flat), _lib.ptr(self.gflat), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                              rt.n_flat, _lib.ptr(rt.seg_off), self.n_seg, _lib.ptr(rt.seg_group), _lib.ptr(self.touched),
                                              _lib.ptr(self.seg_step), _lib.ptr(self.seg_coef), self.lr, self.betas[0], self.betas[1],
                                              self.eps, self.wd, 1.0 / self.world, rt.stream()), "matcha_adamw_step")

    def step(self, x, y, w, alpha=1.0, beta=0.001, random_chrom=0, max_tokens: Optional[int] = None):
        """One optimisation step.  Returns device tensors (bce [scalar view], recon [1], logits [B]); nothing syncs.
        ``random_chrom``: the chromosome of the reconstruction branch (Modules.py:192) as an int, or as a one-element int32 device
        tensor the kernels read when they run (what a captured step needs, ``capture``)."""
        x = x.contiguous()
        y = y.reshape(-1).contiguous()
        w = w.reshape(-1).contiguous()
        logits = self.forward_backward(x, y, w, alpha, beta, random_chrom, max_tokens=max_tokens)
        self.all_reduce()
        self.optimizer_step()
        return self.losses[0], self.losses[1:2], logits

    # ---- hipGraph capture of the single-GPU step ---------------------------------------------------------
    def supports_device_chrom(self) -> bool:
        """Whether this model's kernels take the reconstruction branch's chromosome from device memory (what capturing an adj step
        needs): the fused adj front end of embed_dim 64 does, the table front end never reads it."""
        return bool(self.lib.matcha_random_chrom_dev_supported(C.byref(self.rt.shape), C.byref(self.rt.frozen)))

    def capture(self, x, y, w, alpha=1.0, beta=0.001, random_chrom=0):
        """Capture forward+backward+AdamW on static input buffers; returns a callable that replays the graph.
        Refill ``x``, ``y``, ``w`` (and the ``random_chrom`` cell) in place between replays."""
        if self.world > 1 or self.force_collectives:
            raise RuntimeError("capture() covers the single-GPU step; with DP the all-reduce sits between two graphs")
        if self.rt.mode == 1 and beta != 0.0 and not (isinstance(random_chrom, torch.Tensor) and self.supports_device_chrom()):
            # as an int, random_chrom is a launch parameter of the reconstruction branch: a graph would replay ONE chromosome forever
            # and only recon[r] would ever train; the reference draws a new one per forward (Modules.py:192)
            raise RuntimeError("capture() would freeze random_chrom (adj front end with beta != 0): pass it as a one-element int32 "
                               "device tensor (fused adj front end, embed_dim 64), replay the step eagerly, or capture with beta = 0.  "
                               "alpha and beta are baked into a captured graph as well.")
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
