"""Training driver: the flow of the reference's ``Code/main.py`` on the MI355X-native step.

Same inputs (``./config.JSON`` keys; ``temp_dir/{chrom_range.npy, node2chrom.npy, all_<k>_counter.npy,
all_<k>_freq_counter.npy, intra_adj.npy, inter_adj.npy}``, main.py:538-571), same two phases (main.py:637-643 and
:671-679), same outputs (``temp_dir/model.chkpt`` = {'model_link': state_dict, 'epoch'}, ``temp_dir/model2load`` =
the pickled module, ``../embeddings.npy`` float32 [N,d]; main.py:316-322, :476, :685).  What differs: batches stay on
the GPU, negatives are drawn by the device sampler, and one step is forward+backward+AdamW through the C ABI
(matcha_amd.engine.Trainer) with no per-step host synchronisation.

    cd <dir with config.JSON> && python -m matcha_amd.train [--front-end adj|table] [--epochs1 3 --epochs2 30]

Data parallel (the reference is single-process): launched with
``python -m torch.distributed.run --nproc-per-node N -m matcha_amd.train ...`` every rank runs this same flow on its own GPU.
Everything drawn from numpy's global generator -- the 80/20 split, the epoch shuffles, ``random_chrom`` -- is identical on
every rank (rank 0's seed is broadcast), each global batch of ``world * 96`` positives is cut strided by rank
(parallel.shard_rows), each rank draws its own negatives, gradients are exchanged inside ``Trainer.step``, and only rank 0
writes files.  Work that draws from numpy's global stream (``save_embeddings`` with the adj front end) therefore runs on EVERY
rank -- only the file write is root-only -- so the streams never drift apart (tests/test_hip_data_parallel.py asserts it).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import time
from typing import List, Optional, Tuple

import numpy as np
import torch

import ctypes as C

from . import _lib, utils as U
from .Modules import Classifier, DataGenerator, MultipleEmbedding, Wrap_Embedding
from .engine import Trainer
from .parallel import shard_rows
from .sampler import HyperedgeSet, NegativeSampler


def _dist():
    """(rank, world) of the initialised default process group, (0, 1) without one."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def _barrier():
    if _dist()[1] > 1:
        torch.distributed.barrier()

STATS = {"train_graph_replays": 0}      # captured training graphs replayed by this process (tests: the data-parallel driver does not fall back)
GRAPH_EPOCHS = os.environ.get("MATCHA_TRAIN_GRAPH", "1") != "0"     # single GPU: the epoch loop replays one captured step (Session.graph_epoch)
NEG_NUM = 3          # main.py:527
BATCH_SIZE = 96      # main.py:528 (positives per step)
MODEL_NAME = "model.chkpt"   # main.py:530


def get_attributes(num: List[int]) -> np.ndarray:
    """[N+1, C+1] attribute table: one-hot chromosome || (bin index)/num[0]; row 0 (padding) zeros (main.py:497-512)."""
    C = len(num)
    blocks = [np.zeros((1, C + 1), dtype=np.float32)]
    for i, n in enumerate(num):
        a = np.zeros((n, C + 1), dtype=np.float32)
        a[:, i] = 1.0
        a[:, C] = np.arange(n, dtype=np.float32) / np.float32(num[0])
        blocks.append(a)
    return np.concatenate(blocks, axis=0)


def load_kmers(temp_dir: str, size_list: List[int], cutoff: float) -> Tuple[np.ndarray, np.ndarray]:
    """all_<k>_counter.npy / all_<k>_freq_counter.npy -> (edges int64 [M, max_k] zero-padded, weights float32 [M]) keeping
    rows whose quantile-transformed frequency exceeds ``cutoff`` (main.py:551-566, :649-660).  The transform runs on the
    device (positives.py / csrc/quantile.hip), fitted on every row -- no scikit-learn, no random subsample."""
    from . import positives
    edges, weights = positives.load_kmers(temp_dir, size_list, cutoff)
    return edges.cpu().numpy(), weights.cpu().numpy()


def build_features(temp_dir: str, chrom_range: np.ndarray):
    """Per-chromosome np.corrcoef of the intra-chromosomal contact block, NaN -> 0 (main.py:569-577) + raw inter matrix, both
    built and kept on the device (features.py / csrc/features.hip); ``MultipleEmbedding`` z-scores the inter rows there."""
    from . import features
    return features.build_features(temp_dir, chrom_range)


@torch.no_grad()
def save_embeddings(model: Classifier, n_nodes: int, path: Optional[str] = "../embeddings.npy", batch_size: int = 4096) -> np.ndarray:
    """Eval-mode get_node_embeddings for ids 1..N -> float32 [N,d], row i = node id i+1 (main.py:462-479)."""
    was = model.training
    model.eval()
    dev = model.layer_norm1.weight.device
    ids = torch.arange(1, n_nodes + 1, dtype=torch.long, device=dev).view(-1, 1)
    with model.deferred_id_check():                     # one status read-back for the sweep instead of one per 4096 ids
        out = [model.get_node_embeddings(ids[j:j + batch_size])[:, 0, :] for j in range(0, n_nodes, batch_size)]
    emb = torch.cat(out, dim=0).cpu().numpy()
    if path is not None:
        np.save(path, emb)
    model.train(was)
    return emb


@torch.no_grad()
def predict(model: Classifier, rows, batch_size: int = 100000) -> np.ndarray:
    """Logits [n,1] for a list / array of hyperedges, zero-padded per chunk like pad_sequence (main.py:482-494)."""
    model.eval()
    dev = model.layer_norm1.weight.device
    outs = []
    with model.deferred_id_check():
        for j in range(0, len(rows), batch_size):
            x = U.pad_rows(rows[j:j + batch_size]).to(dev, non_blocking=True)
            outs.append(model(x))
    return np.concatenate([o.cpu().numpy() for o in outs], axis=0) if outs else np.zeros((0, 1), dtype=np.float32)


class Session:
    """Everything one training run needs on the device: model, fused trainer, positive set, negative sampler."""

    def __init__(self, model: Classifier, node2chrom: np.ndarray, chrom_range: np.ndarray, min_size: int, max_size: int, min_dis: int,
                 seed: int = 0, deterministic: bool = False):
        self.deterministic = deterministic
        self.model = model
        self.dev = model.layer_norm1.weight.device
        self.min_size, self.max_size, self.min_dis = min_size, max_size, min_dis
        self.node2chrom, self.chrom_range = node2chrom, chrom_range
        self.n_chrom = len(chrom_range)
        self.rank, self.world = _dist()
        self.trainer = Trainer(model, lr=1e-3, base_seed=seed, deterministic=deterministic)      # AdamW(lr=1e-3), main.py:630
        self._generation = 0          # bumped whenever the trainer or the sampler is replaced: a captured step belongs to ONE pair of them
        self.set_known(None)
        self.rng = np.random.default_rng(seed)

    def new_optimizer(self):
        """main.py:671 builds a fresh AdamW (moments and step counts reset) for phase 2."""
        self.trainer = Trainer(self.model, lr=1e-3, base_seed=int(self.rng.integers(1 << 30)), deterministic=self.deterministic)
        self._generation += 1
        self.__dict__.pop("_graph_state", None)
        self.__dict__.pop("_eval_state", None)

    def set_known(self, edges: Optional[np.ndarray]):
        """The 'dict' negatives are checked against (main.py:589 empty sets in phase 1; build_hash at :664)."""
        if edges is None or len(edges) == 0:
            hs = HyperedgeSet.empty(self.dev, self.max_size)
        else:
            hs = HyperedgeSet(torch.from_numpy(np.ascontiguousarray(edges)).to(self.dev))
        # the draw is rank-shared (numpy's global stream is); the rank offset gives every rank its own negative stream
        self.sampler = NegativeSampler(hs, self.node2chrom, self.chrom_range, neg_num=NEG_NUM, min_dis=self.min_dis,
                                       seed=int(np.random.randint(1 << 30)) + 7919 * self.rank)
        self._generation = getattr(self, "_generation", 0) + 1
        self.__dict__.pop("_graph_state", None)         # a captured step has the old sampler's tables and seed baked in
        self.__dict__.pop("_eval_state", None)

    def make_batch(self, pos: torch.Tensor, pos_w: torch.Tensor):
        """generate_negative's output (main.py:443-448): x = [pos; neg], y = [1..; 0..], w = [pos_w..; 1..]."""
        P = pos.shape[0]
        neg = self.sampler.sample(pos)
        x = torch.cat([pos, neg], dim=0)
        y = torch.cat([torch.ones(P, device=self.dev), torch.zeros(neg.shape[0], device=self.dev)])
        w = torch.cat([pos_w.to(torch.float32), torch.ones(neg.shape[0], device=self.dev)])
        sizes = (x != 0).sum(dim=1)
        return x, y, w, sizes

    def random_chrom(self) -> int:
        # Modules.py:192 draws np.random.choice(np.arange(C), 1); randint consumes numpy's global stream too, at a sixth of the cost
        # (the draw sits on the per-step host path)
        return int(np.random.randint(self.n_chrom)) if self.n_chrom else 0

    # ---- one epoch as replays of ONE captured step (single GPU) -------------------------------------------------------------------
    # The reference's batch is 96 positives + 288 negatives (main.py:527-528): ~0.25 ms of kernels per step, against which a step
    # enqueued call by call from Python (slicing, three torch.cat, the sampler, ~30 launches through ctypes) costs the host several
    # times that.  Everything a step does is already on the device and free of synchronisation, so the step -- positives gathered by a
    # device-side counter, negative sampling, batch assembly, forward, backward, AdamW, the epoch's running sums and its row in the
    # prediction table -- is captured once into a hipGraph and the epoch loop is `graph.replay()` per step.
    def graph_ok(self, beta: float, training: bool = True) -> bool:
        """Whether the epoch loop can replay captured steps.  Data parallel: the gradient exchange sits BETWEEN two graphs (A: selection, negative
        sampling, forward, backward; collectives, eager; B: AdamW + the epoch's records) -- unless the step has a collective of its own in the
        middle (adj front end with beta != 0: the reconstruction mean's row count is exchanged between forward and backward), which stays call
        by call.  A forward-only (evaluation) step has no collective at all."""
        tr = self.trainer
        if tr.force_collectives and self.world == 1:         # bench.py's one-rank RCCL exercise: call by call
            return False
        if self.world > 1 and training and not (tr.rt.mode == 0 or beta == 0.0):
            return False
        return GRAPH_EPOCHS and (tr.rt.mode == 0 or beta == 0.0 or tr.supports_device_chrom())

    def graph_epoch(self, e: torch.Tensor, w: torch.Tensor, n_batch: int, P: int, alpha: float, beta: float, tok_max=None):
        """e int64 [>= n_batch * P, L], w float32: the epoch's shuffled positives (data parallel: THIS RANK's rows of every global batch, in
        step order).  Returns (bce_sum, recon_sum, preds [n_batch, B], labels [B], sizes [n_batch, B]) as device tensors; nothing has
        synchronised.  ``tok_max`` (data parallel): per step the largest real-token count of any rank (host arithmetic of the driver), for
        the compacted row-sparse exchange."""
        dev, L, B = self.dev, int(e.shape[1]), P * (1 + NEG_NUM)
        dp = self.world > 1
        # the chromosome of every step's reconstruction branch, drawn exactly as the step-by-step loop draws them (Modules.py:192)
        chroms = np.asarray([self.random_chrom() for _ in range(n_batch)], dtype=np.int32)
        # (a monotonically increasing generation, not id(): CPython reuses the id of a freed Trainer / sampler)
        key = (n_batch, P, L, float(alpha), float(beta), self._generation)
        st = self.__dict__.get("_graph_state")
        if st is not None and not self.trainer.rt.still_packed():
            # parameters moved since the capture (model.to(), a rebuilt runtime): the graph would train the old buffers -- as
            # Trainer.forward_backward refuses to
            raise _lib.MatchaHipError("model parameters moved after the step was captured; create a new Session / Trainer")
        if st is None or st["key"] != key:
            st = dict(key=key, graph=None,
                      pos=torch.empty((n_batch * P, L), dtype=torch.long, device=dev), w=torch.empty(n_batch * P, dtype=torch.float32, device=dev),
                      chroms=torch.empty(n_batch, dtype=torch.int32, device=dev), cell=torch.zeros(1, dtype=torch.int32, device=dev),
                      it=torch.zeros(1, dtype=torch.long, device=dev),
                      x=torch.zeros((B, L), dtype=torch.long, device=dev),
                      y=torch.cat([torch.ones(P, device=dev), torch.zeros(B - P, device=dev)]),           # main.py:444-445
                      ww=torch.ones(B, dtype=torch.float32, device=dev),                                  # main.py:446-447
                      preds=torch.empty((n_batch, B), dtype=torch.float32, device=dev), sizes=torch.empty((n_batch, B), dtype=torch.long, device=dev),
                      sums=torch.zeros(2, dtype=torch.float32, device=dev))
            self._graph_state = st
        st["pos"].copy_(e[:n_batch * P]); st["w"].copy_(w[:n_batch * P]); st["chroms"].copy_(torch.from_numpy(chroms))
        st["it"].zero_(); st["sums"].zero_()

        lib = _lib.load()
        n_rows = n_batch * P

        tr = self.trainer

        def part_a(step_i: int):
            # everything that changes from step to step lives on the device: the step counter `it` picks the step's positives, weights
            # and reconstruction chromosome (matcha_step_select) and the row of the epoch's prediction / size buffers (matcha_step_record)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.matcha_step_select(_lib.ptr(st["pos"]), _lib.ptr(st["w"]), n_rows, L, _lib.ptr(st["it"]), P, _lib.ptr(st["x"]),
                                              _lib.ptr(st["ww"]), _lib.ptr(st["chroms"]), n_batch, _lib.ptr(st["cell"]),
                                              _lib.ptr(self.sampler.seed), _lib.ptr(tr.seed), stream),
                       "matcha_step_select")
            self.sampler.sample_into(st["x"][:P], st["x"][P:], advance_seed=False)
            tr.seed_advanced_by_caller = True
            try:
                # (data parallel: a host-side token bound keeps the count exchange -- a collective -- out of the step; the value that
                # matters is set per step in front of the exchange, below)
                st["logits"] = tr.forward_backward(st["x"], st["y"], st["ww"], alpha, beta, st["cell"],
                                                   max_tokens=None if tok_max is None else tok_max[step_i])
            finally:
                tr.seed_advanced_by_caller = False

        def exchange(step_i: int):
            if dp:
                tr._host_max_tokens = None if tok_max is None else int(tok_max[step_i])
                tr.all_reduce()

        def part_b():
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            tr.optimizer_step()
            _lib.check(lib.matcha_step_record(_lib.ptr(st["logits"]), _lib.ptr(tr.losses), _lib.ptr(st["x"]), B, L, _lib.ptr(st["it"]),
                                              n_batch, _lib.ptr(st["sums"]), _lib.ptr(st["preds"]), _lib.ptr(st["sizes"]), stream),
                       "matcha_step_record")

        def one_step(step_i: int):
            part_a(step_i); exchange(step_i); part_b()

        done = 0
        tm = self.__dict__.setdefault("timing", {})
        tm["graph_replays"] = 0
        if os.environ.get("MATCHA_TRAIN_GRAPH") == "steps":      # development: the same device-side step function, enqueued call by call
            for i in range(n_batch):
                one_step(i)
            return st["sums"][0], st["sums"][1], st["preds"], st["y"], st["sizes"]
        if st["graph"] is None:
            # the first steps of the epoch run call by call on a side stream (they ARE steps of the epoch: buffers get allocated,
            # kernels loaded), then the same functions are captured; capturing enqueues nothing
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(min(2, n_batch)):
                    one_step(done)
                    done += 1
            torch.cuda.current_stream(dev).wait_stream(side)
            if dp:
                # data parallel: the gradient exchange (RCCL all-reduce / all-gather, or gloo on the test box) sits between two graphs
                ga, gb_ = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga):
                    part_a(min(done, n_batch - 1))
                with torch.cuda.graph(gb_):
                    part_b()
                st["graph"] = (ga, gb_)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    one_step(min(done, n_batch - 1))
                st["graph"] = g
        if dp:
            ga, gb_ = st["graph"]
            for i in range(done, n_batch):
                ga.replay()
                exchange(i)
                gb_.replay()
                tm["graph_replays"] += 2
                STATS["train_graph_replays"] += 2
        else:
            for _ in range(n_batch - done):
                st["graph"].replay()
                tm["graph_replays"] += 1
                STATS["train_graph_replays"] += 1
        return st["sums"][0], st["sums"][1], st["preds"], st["y"], st["sizes"]


def _graph_eval(sess: Session, e: torch.Tensor, w: torch.Tensor, n_batch: int, P: int):
    """eval_epoch's loop as replays of ONE captured forward-only step (round 5): device-side batch selection
    (matcha_step_select), negative sampling, matcha_forward(forward_only) with the loss inside, matcha_step_record.  Same device-side
    state machine as Session.graph_epoch; returns (bce_sum, recon_sum, preds [n_batch, B], labels [B], sizes [n_batch, B])."""
    dev, L, B = sess.dev, int(e.shape[1]), P * (1 + NEG_NUM)
    # the reconstruction chromosome of every forward, drawn exactly as Classifier.forward draws it (Modules.py:192: np.random.choice on numpy's
    # global stream, adj front end only) so that the call-by-call loop and this one leave the stream in the same state
    if sess.trainer.rt.mode == 1:
        chroms = np.asarray([int(np.random.choice(np.arange(sess.n_chrom), 1)[0]) for _ in range(n_batch)], dtype=np.int32)
    else:
        chroms = np.zeros(n_batch, dtype=np.int32)
    key = (n_batch, P, L, sess._generation)
    st = sess.__dict__.get("_eval_state")
    if st is not None and not sess.trainer.rt.still_packed():
        # parameters moved since the capture (model.to(), a rebuilt runtime): a replay would evaluate the old buffers (as graph_epoch refuses to)
        raise _lib.MatchaHipError("model parameters moved after the evaluation step was captured; create a new Session / Trainer")
    if st is None or st["key"] != key:
        st = dict(key=key, graph=None,
                  pos=torch.empty((n_batch * P, L), dtype=torch.long, device=dev), w=torch.empty(n_batch * P, dtype=torch.float32, device=dev),
                  chroms=torch.empty(n_batch, dtype=torch.int32, device=dev), cell=torch.zeros(1, dtype=torch.int32, device=dev),
                  it=torch.zeros(1, dtype=torch.long, device=dev), x=torch.zeros((B, L), dtype=torch.long, device=dev),
                  y=torch.cat([torch.ones(P, device=dev), torch.zeros(B - P, device=dev)]), ww=torch.ones(B, dtype=torch.float32, device=dev),
                  preds=torch.empty((n_batch, B), dtype=torch.float32, device=dev), sizes=torch.empty((n_batch, B), dtype=torch.long, device=dev),
                  sums=torch.zeros(2, dtype=torch.float32, device=dev))
        sess._eval_state = st
    st["pos"].copy_(e[:n_batch * P]); st["w"].copy_(w[:n_batch * P]); st["chroms"].copy_(torch.from_numpy(chroms))
    st["it"].zero_(); st["sums"].zero_()
    lib = _lib.load()
    n_rows = n_batch * P
    tr = sess.trainer

    def one_step():
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.matcha_step_select(_lib.ptr(st["pos"]), _lib.ptr(st["w"]), n_rows, L, _lib.ptr(st["it"]), P, _lib.ptr(st["x"]), _lib.ptr(st["ww"]),
                                          _lib.ptr(st["chroms"]), n_batch, _lib.ptr(st["cell"]), _lib.ptr(sess.sampler.seed), None, stream),
                   "matcha_step_select")          # (no dropout in a forward-only step: the trainer's dropout seed stays where the call-by-call loop leaves it)
        sess.sampler.sample_into(st["x"][:P], st["x"][P:], advance_seed=False)
        logits = tr.eval_forward(st["x"], st["y"], st["ww"], random_chrom=st["cell"])
        _lib.check(lib.matcha_step_record(_lib.ptr(logits), _lib.ptr(tr.losses), _lib.ptr(st["x"]), B, L, _lib.ptr(st["it"]), n_batch, _lib.ptr(st["sums"]),
                                          _lib.ptr(st["preds"]), _lib.ptr(st["sizes"]), stream), "matcha_step_record")

    done = 0
    if st["graph"] is None:
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(min(2, n_batch)):
                one_step()
                done += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one_step()
        st["graph"] = g
    for _ in range(n_batch - done):
        st["graph"].replay()
    return st["sums"][0], st["sums"][1], st["preds"], st["y"], st["sizes"]


def train_epoch(sess: Session, edges: np.ndarray, weights: np.ndarray, alpha: float, beta: float, batch_size: int = BATCH_SIZE):
    """main.py:119-197: shuffle, floor(len/batch) steps of (negatives, forward, backward, AdamW), epoch metrics."""
    dev = sess.dev
    model = sess.model
    model.train()
    e = torch.from_numpy(edges).to(dev)
    w = torch.from_numpy(weights.astype(np.float32)).to(dev)
    perm_host = np.random.permutation(len(e))                                    # sync_shuffle, utils.py:142-149 (rank-shared stream)
    perm = torch.from_numpy(perm_host).to(dev)
    e, w = e[perm], w[perm]
    gb = batch_size * sess.world                                                # global batch: every rank steps on batch_size positives
    n_batch = len(e) // gb
    mine = torch.from_numpy(shard_rows(gb, sess.rank, sess.world)).to(dev)      # this rank's rows of a global batch (strided)
    # data parallel: every rank holds the epoch's whole (shuffled) positive list on the host, a negative has the size of its positive, so
    # the largest token count of any rank in step i is host arithmetic -- the row-sparse exchange then needs no device-side count
    tok_max = None
    if sess.world > 1:
        k_host = (edges != 0).sum(1)[perm_host][:n_batch * gb].reshape(n_batch, batch_size, sess.world)        # [step, row, rank]: strided shards
        tok_max = ((1 + NEG_NUM) * k_host.sum(1).max(1)).tolist()
    tm = sess.__dict__.setdefault("timing", {})          # wall clock of the epoch's phases (tools/epoch_bench.py reads it)
    t_loop = time.perf_counter()
    if n_batch > 0 and sess.graph_ok(beta):
        if sess.world > 1:
            # this rank's rows of every global batch (strided shards, as `mine` below), contiguous in step order
            e_r = e[:n_batch * gb].reshape(n_batch, batch_size, sess.world, -1)[:, :, sess.rank].reshape(n_batch * batch_size, -1).contiguous()
            w_r = w[:n_batch * gb].reshape(n_batch, batch_size, sess.world)[:, :, sess.rank].reshape(-1).contiguous()
        else:
            e_r, w_r = e, w
        bce_sum, rec_sum, p2, y1, s2 = sess.graph_epoch(e_r, w_r, n_batch, batch_size, alpha, beta, tok_max=tok_max)
        if sess.world > 1:                                                       # epoch means over all ranks, as in the call-by-call loop
            both = torch.stack([bce_sum, rec_sum])
            torch.distributed.all_reduce(both)
            bce_sum, rec_sum = both[0] / sess.world, both[1] / sess.world
        pred, label, size = p2.reshape(-1), y1.repeat(n_batch), s2.reshape(-1)
        torch.cuda.synchronize(dev)                                              # the epoch's one synchronisation
        tm["loop_s"] = time.perf_counter() - t_loop
        return _epoch_metrics(sess, bce_sum, rec_sum, pred, label, size, n_batch)
    bce_sum = torch.zeros((), device=dev)
    rec_sum = torch.zeros((), device=dev)
    preds, labels, sizes = [], [], []
    for i in range(n_batch):
        pos, pw = e[i * gb:(i + 1) * gb][mine], w[i * gb:(i + 1) * gb][mine]
        x, y, ww, s = sess.make_batch(pos, pw)
        bce, recon, logits = sess.trainer.step(x, y, ww, alpha=alpha, beta=beta, random_chrom=sess.random_chrom(),
                                               max_tokens=None if tok_max is None else tok_max[i])
        bce_sum += bce
        rec_sum += recon[0]
        preds.append(torch.sigmoid(logits).clone())                             # main.py:58
        labels.append(y)
        sizes.append(s)
    if sess.world > 1:                                                           # epoch means over all ranks; metrics below: this rank's rows
        both = torch.stack([bce_sum, rec_sum])
        torch.distributed.all_reduce(both)
        bce_sum, rec_sum = both[0] / sess.world, both[1] / sess.world
    pred, label, size = torch.cat(preds), torch.cat(labels), torch.cat(sizes)
    torch.cuda.synchronize(dev)                                                  # the epoch's one synchronisation
    tm["loop_s"] = time.perf_counter() - t_loop
    return _epoch_metrics(sess, bce_sum, rec_sum, pred, label, size, n_batch)


def _epoch_metrics(sess: Session, bce_sum, rec_sum, pred, label, size, n_batch: int):
    t_m = time.perf_counter()
    try:
        return _epoch_metrics_impl(sess, bce_sum, rec_sum, pred, label, size, n_batch)
    finally:
        sess.__dict__.setdefault("timing", {})["metrics_s"] = time.perf_counter() - t_m


def _epoch_metrics_impl(sess: Session, bce_sum, rec_sum, pred, label, size, n_batch: int):
    sess.trainer.check_status()                                                  # IndexError if a node id outside [0, N] reached a step
    exhausted = sess.sampler.check_status()                                      # KeyError for nodes without a chromosome
    if exhausted:
        print(f"warning: {exhausted} negatives could not be redrawn within 65536 trials and were returned equal to their positive")
    auc, aupr = U.roc_auc_cuda(label, pred, size, sess.max_size)
    acc = U.accuracy(pred, label, size, sess.max_size)
    return float(bce_sum) / max(n_batch, 1), float(rec_sum) / max(n_batch, 1), acc, auc, aupr


@torch.no_grad()
def eval_epoch(sess: Session, edges: np.ndarray, weights: np.ndarray, batch_size: int = BATCH_SIZE, max_rows: int = 10000):
    """main.py:200-258: <= 10000 shuffled validation positives, fresh negatives, forward + loss only."""
    dev = sess.dev
    model = sess.model
    model.eval()
    import ctypes as C
    from . import _lib
    e = torch.from_numpy(edges).to(dev)
    w = torch.from_numpy(weights.astype(np.float32)).to(dev)
    perm = torch.from_numpy(np.random.permutation(len(e))[:max_rows]).to(dev)   # rank-shared stream: every rank evaluates the same rows
    e, w = e[perm], w[perm]
    n_batch = len(e) // batch_size
    tm = sess.__dict__.setdefault("timing", {})
    t_loop = time.perf_counter()
    if n_batch > 0 and sess.graph_ok(1.0, training=False):
        # the forward-only step captured once and replayed per batch (the training epoch's mechanism); metrics as in _epoch_metrics
        bce_t, rec_t, p2, y1, s2 = _graph_eval(sess, e, w, n_batch, batch_size)
        pred, label, size = p2.reshape(-1), y1.repeat(n_batch), s2.reshape(-1)
        torch.cuda.synchronize(dev)
        tm["eval_loop_s"] = time.perf_counter() - t_loop
        sess.trainer.check_status()
        auc, aupr = U.roc_auc_cuda(label, pred, size, sess.max_size)
        acc = U.accuracy(pred, label, size, sess.max_size)
        return float(bce_t) / n_batch, float(rec_t) / n_batch, acc, auc, aupr
    bce_sum, rec_sum = torch.zeros((), device=dev), torch.zeros((), device=dev)
    preds, labels, sizes = [], [], []
    with model.deferred_id_check():                     # nothing synchronises inside the loop; ids are checked once behind it
        for i in range(n_batch):
            pos, pw = e[i * batch_size:(i + 1) * batch_size], w[i * batch_size:(i + 1) * batch_size]
            x, y, ww, s = sess.make_batch(pos, pw)
            logits, recon = model(x, return_recon=True)
            bce_sum += torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), y, weight=ww)
            rec_sum += recon[0]
            preds.append(torch.sigmoid(logits.view(-1)))
            labels.append(y)
            sizes.append(s)
    bce_sum, rec_sum = float(bce_sum), float(rec_sum)
    pred, label, size = torch.cat(preds), torch.cat(labels), torch.cat(sizes)
    auc, aupr = U.roc_auc_cuda(label, pred, size, sess.max_size)
    acc = U.accuracy(pred, label, size, sess.max_size)
    return bce_sum / max(n_batch, 1), rec_sum / max(n_batch, 1), acc, auc, aupr


def train(sess: Session, training_data, validation_data, epochs: int, alpha: float, beta: float, temp_dir: str, n_nodes: int,
          batch_size: int = BATCH_SIZE, batches_per_epoch: int = 1000, emb_path: Optional[str] = "../embeddings.npy", log=print):
    """main.py:261-342: per epoch save embeddings, draw batches_per_epoch*batch rows per size, train, validate, checkpoint.
    (The reference's 'best' checkpoint is in fact the last one -- it compares the parsed size label, main.py:313-322 --
    so every epoch is saved.)"""
    edges, weights = training_data
    rows = [r[r != 0] for r in edges]
    root = sess.rank == 0
    log = log if root else (lambda *a, **k: None)
    # an "iteration" serves batches_per_epoch GLOBAL batches: world times the rows of the single-process run, same number of steps
    gen = DataGenerator(rows, weights, int(batch_size) * sess.world, batches_per_epoch, min_size=sess.min_size, max_size=sess.max_size)
    for epoch in range(epochs):
        # every rank runs the sweep (the adj front end draws from numpy's rank-shared global stream once per chunk, Modules.py:192:
        # a root-only call would leave rank 0's stream ahead of the others and the "global batch" would stop being one); root writes
        save_embeddings(sess.model, n_nodes, emb_path if root else None)
        t0 = time.time()
        e_part, w_part = gen.next_iter()
        bce, rec, acc, auc, aupr = train_epoch(sess, e_part, w_part, alpha, beta, batch_size)
        log(f"[ Epoch {epoch} of {epochs} ]  - (Training)   bce: {bce:7.4f}, recon: {rec:7.4f} acc: {acc}, auc: {auc}, aupr: {aupr}, "
            f"elapse: {time.time() - t0:3.3f} s")
        t0 = time.time()
        vb, vr, vacc, vauc, vaupr = eval_epoch(sess, validation_data[0], validation_data[1], batch_size)
        log(f"  - (Validation-hyper) bce: {vb:7.4f}, recon: {vr:7.4f},  acc: {vacc}, auc: {vauc}, aupr: {vaupr}, elapse: {time.time() - t0:3.3f} s")
        if root:
            torch.save({"model_link": sess.model.state_dict(), "epoch": epoch}, os.path.join(temp_dir, MODEL_NAME))   # main.py:316-321
            torch.save(sess.model, os.path.join(temp_dir, "model2load"))                                               # main.py:322
    _barrier()                                                                  # rank 0's checkpoint is on disk
    ck = torch.load(os.path.join(temp_dir, MODEL_NAME), map_location=sess.dev, weights_only=False)              # main.py:326-327
    sess.model.load_state_dict(ck["model_link"])


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, str]:
    """Join the process group a ``torch.distributed.run`` launch describes (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the
    environment): one process per GPU, backend "nccl" (= RCCL on ROCm) unless MATCHA_DIST_BACKEND / ``backend`` says otherwise
    (the two-ranks-on-one-GPU test uses gloo).  Returns (rank, world, device); (0, 1, "cuda") outside such a launch."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or "RANK" not in os.environ:
        return 0, 1, "cuda"
    backend = backend or os.environ.get("MATCHA_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else int(os.environ.get("MATCHA_LOCAL_DEVICE", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not torch.distributed.is_initialized():
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        torch.distributed.init_process_group(backend, **kw)
    return torch.distributed.get_rank(), world, f"cuda:{local}"


def run(config: dict, front_end: str = "adj", epochs1: int = 3, epochs2: int = 30, batches_per_epoch: int = 1000, device: str = "cuda",
        emb_path: Optional[str] = "../embeddings.npy", log=print, deterministic: bool = False) -> Classifier:
    """The script body of main.py:516-685 (on every rank of a data-parallel launch, see the module docstring)."""
    rank, world = _dist()
    if world > 1:
        # numpy's global generator drives the split, the epoch shuffles and random_chrom: one seed for all ranks
        seed = torch.tensor([int(np.random.randint(1 << 31)) if rank == 0 else 0], dtype=torch.int64, device=device)
        torch.distributed.broadcast(seed, 0)
        np.random.seed(int(seed.item()))
        torch.manual_seed(int(seed.item()))          # identical initial weights before the Trainer's broadcast, too
    d = int(config["embed_dim"])
    size_list = [int(v) for v in config["k-mer_size"]]
    min_size, max_size = min(size_list), max(size_list)
    temp_dir = config["temp_dir"]
    min_dis = int(config["min_distance"])
    chrom_range = np.load(os.path.join(temp_dir, "chrom_range.npy")).astype(np.int64)
    n2c_dict = np.load(os.path.join(temp_dir, "node2chrom.npy"), allow_pickle=True).item()
    num = [int(v[1] - v[0]) for v in chrom_range]
    N = int(np.sum(num))
    node2chrom = np.full(N + 1, -1, dtype=np.int32)
    for k, v in n2c_dict.items():
        if 0 < int(k) <= N:
            node2chrom[int(k)] = int(v)

    data, weight = load_kmers(temp_dir, size_list, float(config["quantile_cutoff_for_positive"]))
    weight = weight / np.mean(weight) * NEG_NUM                                  # main.py:594-595
    idx = np.arange(len(data))
    np.random.shuffle(idx)                                                        # main.py:598
    split = int(0.8 * len(idx))
    train_data, test_data = data[idx[:split]], data[idx[split:]]
    train_w, test_w = weight[idx[:split]], weight[idx[split:]]

    attr = get_attributes(num)
    if front_end == "adj":
        feats, inter = build_features(temp_dir, chrom_range)
        ne = MultipleEmbedding(feats, d, False, torch.as_tensor(np.cumsum(num)), chrom_range, inter)   # main.py:609-613
    else:
        ne = Wrap_Embedding(N + 1, d, padding_idx=0)
    model = Classifier(n_head=8, d_model=d, d_k=d, d_v=d, node_embedding=ne, diag_mask=True, bottle_neck=d, attribute_dict=attr).to(device)
    save_embeddings(model, N, emb_path if rank == 0 else None)                     # main.py:625 (every rank: keeps the numpy stream shared)

    sess = Session(model, node2chrom, chrom_range.astype(np.int32), min_size, max_size, min_dis, deterministic=deterministic)
    # phase 1: alpha 0, beta 1, empty dict (negatives == positives)             main.py:637-643
    train(sess, (train_data, train_w), (test_data, test_w), epochs1, 0.0, 1.0, temp_dir, N, batches_per_epoch=batches_per_epoch,
          emb_path=emb_path, log=log)
    # phase 2: dict of all k-mers above the unlabel cutoff, fresh AdamW, alpha 1, beta 1e-3   main.py:646-679
    dict_data, _ = load_kmers(temp_dir, size_list, float(config["quantile_cutoff_for_unlabel"]))
    sess.set_known(dict_data)
    sess.new_optimizer()
    train(sess, (train_data, train_w), (test_data, test_w), epochs2, 1.0, 0.001, temp_dir, N, batches_per_epoch=batches_per_epoch,
          emb_path=emb_path, log=log)
    save_embeddings(model, N, emb_path if rank == 0 else None)                     # main.py:684
    if rank == 0:
        torch.save(model, os.path.join(temp_dir, "model2load"))                  # main.py:685
    _barrier()
    return model


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", default="./config.JSON")
    ap.add_argument("--front-end", choices=["adj", "table"], default="adj")
    ap.add_argument("--epochs1", type=int, default=3)
    ap.add_argument("--epochs2", type=int, default=30)
    ap.add_argument("--batches-per-epoch", type=int, default=1000)
    ap.add_argument("--deterministic", action="store_true", help="table front end: sorted embedding backward (bitwise reproducible table)")
    a = ap.parse_args(argv)
    rank, world, device = init_distributed()
    run(U.get_config(a.config), a.front_end, a.epochs1, a.epochs2, a.batches_per_epoch, device=device, deterministic=a.deterministic)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
