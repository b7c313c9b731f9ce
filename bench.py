#!/usr/bin/env python3
"""Benchmark of the hyperedge-classifier training step on MI355X (BASELINE.json metric:
"training hyperedges/sec at k in {2..5}, embed_dim=64; 1/2/4/8 MI355X").

One step = one pass of the hot path over one batch resident in HBM: draw a batch of positives from the
device-resident pool, generate neg_num=3 negatives per positive on the GPU (main.py:361-459), forward + weighted
BCE, backward, [RCCL exchange of the gradients], fused AdamW -- i.e. main.py:155-183 for one batch.  Rows =
positives + negatives, every one of which goes through forward+backward+update.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)
    python bench.py --layout c5 --dim 256 --ks 2,3,4,5,6,7,8 --rows 16384 --edges 100000000     (BASELINE configs[4])

Prints ONE JSON line on rank 0:
  value / ms_per_step   the headline workload (configs[2]: hg38 1 Mb, mixed k in {2..5}, d = 64, table front end)
  roofline              the dominant kernel class, timed live with HIP events on the launch stream inside the timed region
  roofline_gather       (N = 1) the embedding-row gather alone on 1 M x 64, 16 M x 64 and 1 M x 256 tables with uniform ids, timed
                        live the same way: achieved = rows * (4 d + 8) / t against the 8 TB/s HBM-read roof (north_star target 40 %)
  extra_points          (N = 1, default run only) the other workloads the survey names, each a short run of the same step: adj front
                        end at configs[2], the reference's own 384-row batch, configs[3] (d = 128) and configs[4] (C5, d = 256)
  cpu_baseline          the oracle ("port" of the reference's PyTorch-CPU step, pinned to the reference by tests/golden) timed on
                        this box's host cores on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import ctypes as C
import gc
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from matcha_amd import _lib, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # MI355X_MICROARCH.md: measured float4 copy (read + write)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA peak (no xf32/TF32 on gfx950)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)
# Round 5: the fused embed_dim-64 / 128 kernels and the wide GEMMs of embed_dim >= 128 compute every f32 product as SIX bf16 MFMAs over three bf16
# planes per operand (v = h + m + l to 2^-27, f32 accumulate: fp32-accurate) -- 6 x 32 cycles on the matrix pipe instead of
# 8 x 64 on the vector ALUs.  Round 6: a roofline `frac` is EXECUTED work over the peak of the pipe that executes it (mfma_roofline below):
# plane-product flops (6 per executed f32 flop) against the dense bf16 peak for these classes; the contract's algorithmic figure (the reference
# formulation's f32 flops / time / the f32 MFMA peak) is reported beside it as frac_reference_f32 and may exceed 1.
BF16X3_FLOPS_PER_F32_FLOP = 6.0
# MFMA-bound kernel classes.  The fused kernels count ALGORITHMIC GEMM flops only: fused_fwd 4 projections per
# head + the two pff GEMMs; fused_bwd 8 GEMMs per head (dO, dWfc1, 3 dW', 3 d x_hat terms) -- the Q/K/V recompute of the
# backward kernel and the O(k) attention arithmetic are not counted.
GEMM_CLASSES = ("gemm_nt", "gemm_nn", "gemm_tn", "fused_fwd", "fused_bwd", "adj_encode", "adj_recon", "adj_bwd")
# Merged heads (embed_dim 64, round 3): with d_k = d_v = d_model the four per-head products of the reference collapse into two in the
# forward (r = B_h x, dyn += M_h z) and eight into four in the backward.  `achieved` stays what the contract asks for -- the reference
# formulation's ALGORITHMIC flops over the kernel's time -- and `executed_fraction` says what share of them the matrix cores really
# run (so achieved x executed_fraction is the executed rate, the one bounded by the f32 MFMA peak).
EXECUTED_FRACTION_MERGED = {"fused_fwd": (8 * 2 + 2) / (8 * 4 + 2.0), "fused_bwd": 4 / 8.0}
# embed_dim 128 (enc128.hip): the fused attention block alone -- 4 projections per head forward (the two pff GEMMs stay separate launches), 8 GEMMs
# per head backward; merged heads execute half of each
EXECUTED_FRACTION_MERGED_128 = {"fused_fwd": 0.5, "fused_bwd": 0.5}
METRIC = "training hyperedges/sec at k∈{2..5}, embed_dim=64; 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=7, help="timed windows of --steps steps each; ms_per_step / value are the MEDIAN window "
                                                             "(min and max are reported beside it)")
    ap.add_argument("--rows", type=int, default=65536, help="rows (positives+negatives) per GPU per step")
    ap.add_argument("--front-end", choices=["table", "adj"], default="table")
    ap.add_argument("--layout", default="hg38_1mb", help="hg38_1mb | hg38_100kb | c1 | c5 (matcha_amd/synth.py LAYOUTS)")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--ks", default="2,3,4,5")
    ap.add_argument("--edges-per-k", type=int, default=100000)
    ap.add_argument("--edges", type=int, default=0, help="layout c5: total number of known hyperedges (default 100 M), built on the device")
    ap.add_argument("--prof", default="auto", help="kernel class for the live roofline (see matcha_amd/_lib.py PROF) or 'none'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_gather and extra_points (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (single GPU)")
    ap.add_argument("--table-exchange", choices=["auto", "dense", "sparse"], default="auto")
    ap.add_argument("--deterministic", action="store_true", help="sorted (bitwise reproducible) embedding backward instead of float atomics")
    ap.add_argument("--zipf", action="store_true", help="layout c5: node ids of the known hyperedges drawn Zipf(1.0) instead of uniform (cached regime, SURVEY §8 d2)")
    return ap.parse_args()


def attribute_table(num):
    """[N+1, C+1] one-hot chromosome || normalised position, row 0 zeros (what main.py:497-512 builds)."""
    C_ = len(num)
    out = np.zeros((int(np.sum(num)) + 1, C_ + 1), dtype=np.float32)
    lo = 1
    for i, n in enumerate(num):
        out[lo:lo + n, i] = 1.0
        out[lo:lo + n, C_] = np.arange(n, dtype=np.float32) / np.float32(num[0])
        lo += n
    return out


def make_model(front_end, dim, num, device):
    import Modules as M
    d = dim
    N = int(np.sum(num))
    torch.manual_seed(0)
    if front_end == "table":
        ne = M.Wrap_Embedding(N + 1, d, padding_idx=0)
    elif N > 10000:
        # hg38 100 kb (BASELINE configs[3] in the reference's own mode, main.py:609-613): correlation-like features (one [n_c, n_c] block
        # per chromosome, 190 MB) and a sparse positive inter matrix (3.7 GB), generated on the device -- the host generator of
        # synth.make_adjacency would need two dense N x N float32 matrices
        gen = torch.Generator(device=device).manual_seed(9)
        feats = [torch.rand((n, n), device=device, generator=gen) * 2 - 1 for n in num]
        inter = torch.rand((N, N), device=device, generator=gen) * (torch.rand((N, N), device=device, generator=gen) < 0.3)
        ne = M.MultipleEmbedding(feats, d, False, torch.as_tensor(np.cumsum(num)), synth.chrom_range(num), inter)
    else:
        intra, inter = synth.make_adjacency(np.random.default_rng(2), num)
        cr = synth.chrom_range(num)
        feats = []
        for lo, hi in cr:
            with np.errstate(invalid="ignore", divide="ignore"):
                c = np.corrcoef(intra[lo - 1:hi - 1, lo - 1:hi - 1]).astype(np.float32)
            c[np.isnan(c)] = 0.0
            feats.append(c)
        ne = M.MultipleEmbedding(feats, d, False, torch.as_tensor(np.cumsum(num)), cr, inter)
    clf = M.Classifier(n_head=8, d_model=d, d_k=d, d_v=d, node_embedding=ne, diag_mask=True, bottle_neck=d,
                       attribute_dict=attribute_table(num))
    return clf.to(device)


def executed_fraction(lib, kernel_class, dim):
    """Share of a kernel class's algorithmic GEMM flops that the merged-heads kernels execute (1.0 for everything else)."""
    if dim == 64 and kernel_class in EXECUTED_FRACTION_MERGED and lib.matcha_get_option(b"disable_merged") == 0 \
            and lib.matcha_get_option(b"disable_fused") == 0:
        return EXECUTED_FRACTION_MERGED[kernel_class]
    if dim == 128 and kernel_class in EXECUTED_FRACTION_MERGED_128 and lib.matcha_get_option(b"disable_merged") == 0 \
            and lib.matcha_get_option(b"disable_fused") == 0:
        return EXECUTED_FRACTION_MERGED_128[kernel_class]
    return 1.0


def bf16x3_class(lib, kernel_class, dim):
    """Does this kernel class run its products as 3 x bf16 split-operand MFMAs (round 5)?"""
    if dim == 64:
        return kernel_class in ("fused_fwd", "fused_bwd") and lib.matcha_get_option(b"disable_fused") == 0 and lib.matcha_get_option(b"disable_merged") == 0
    if dim == 128 and kernel_class in ("fused_fwd", "fused_bwd"):
        return lib.matcha_get_option(b"disable_fused") == 0 and lib.matcha_get_option(b"disable_merged") == 0
    return dim >= 128 and kernel_class in ("gemm_nt", "gemm_nn", "gemm_tn") and lib.matcha_get_option(b"disable_wide_gemm") == 0


def mfma_roofline(lib, kernel_class, dim, ach_ref_tflops):
    """The roofline entry of an MFMA-bound kernel class.  `ach_ref_tflops` = the reference formulation's ALGORITHMIC f32 flops (SURVEY.md
    §8 d4) over the kernel's time.  `frac` is EXECUTED work over the peak of the pipe that executes it (round-5 review: the algorithmic
    figure over the f32 MFMA peak exceeded 1 once the products had left that pipe, and a fraction above 1 is not a roofline fraction):
      * bf16x3 classes (the fused embed_dim 64 / 128 kernels, the wide GEMMs): merged heads execute `executed_fraction` of the reference's
        products, each as SIX bf16 plane products (3 planes per f32 operand, f32 accumulate) -> achieved = plane-product TFLOP/s against the
        dense bf16 MFMA peak;
      * f32-MFMA classes: executed f32 TFLOP/s against the f32 MFMA peak.
    The contract's figure stays beside it as achieved_reference_f32 / frac_reference_f32 (it may exceed 1 for the bf16x3 classes)."""
    ex = executed_fraction(lib, kernel_class, dim)
    out = dict(bound="mfma", kernel=kernel_class)
    if bf16x3_class(lib, kernel_class, dim):
        a = ach_ref_tflops * ex * BF16X3_FLOPS_PER_F32_FLOP
        out.update(achieved=round(a, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(a / MFMA_BF16_PEAK_TFLOPS, 4),
                   peak_is="dense bf16 MFMA (the pipe that executes the products: 3 bf16 planes per f32 operand, 6 plane products per f32 product, "
                           "f32 accumulate -- fp32-accurate, DESIGN.md 4.0)",
                   achieved_is="executed bf16 plane-product flops / kernel time")
    else:
        a = ach_ref_tflops * ex
        out.update(achieved=round(a, 3), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(a / MFMA_F32_PEAK_TFLOPS, 4),
                   peak_is="f32 MFMA", achieved_is="executed f32 flops / kernel time")
    out.update(executed_fraction=round(ex, 4), achieved_reference_f32=round(ach_ref_tflops, 3), peak_reference_f32=MFMA_F32_PEAK_TFLOPS,
               frac_reference_f32=round(ach_ref_tflops / MFMA_F32_PEAK_TFLOPS, 4),
               reference_note="achieved_reference_f32 = the reference formulation's algorithmic f32 flops (SURVEY.md 8 d4) / kernel time -- the contract's "
                              "`achieved`; over the f32 MFMA peak it can exceed 1 where the products run on the bf16 pipe",
               pmc=None)
    return out


def csrc_sha16():
    """Fingerprint of the kernel sources: PMC traffic figures committed under profiles/ are only quoted for the tree they
    were measured on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "matcha_amd", "csrc", "*.h*"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class Dist:
    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # torch.distributed.run / torchrun
        # plumbing tests on a one-GPU box: MATCHA_DIST_BACKEND=gloo MATCHA_LOCAL_DEVICE=0 runs every rank on device 0 (as
        # matcha_amd.train.init_distributed does); the driver's runs use neither: one rank per GPU over RCCL
        self.backend = os.environ.get("MATCHA_DIST_BACKEND", "nccl")
        if "MATCHA_LOCAL_DEVICE" in os.environ:
            self.local_rank = int(os.environ["MATCHA_LOCAL_DEVICE"])
        self.device = torch.device("cuda", self.local_rank)

    def barrier(self):
        if self.world > 1 or self.launched:
            torch.distributed.barrier()
        torch.cuda.synchronize(self.device)


def run_workload(dist, *, layout, dim, ks, rows, front_end, steps, warmup, prof="auto", graph=False, edges_per_k=100000, edges=0,
                 table_exchange="auto", model_only=True, deterministic=False, zipf=False, windows=1):
    """Build the workload on the device, run `warmup` untimed + `steps` timed steps, return the measurements."""
    from matcha_amd.engine import Trainer
    from matcha_amd.sampler import HyperedgeSet, NegativeSampler
    device, world, rank = dist.device, dist.world, dist.rank
    num = synth.LAYOUTS[layout]
    N = int(np.sum(num))
    L = max(ks)
    neg_num = 3                                   # main.py:527
    P = rows // (1 + neg_num)                     # positives per GPU per step
    B = P * (1 + neg_num)

    # ---- synthetic positives (same pool on every rank; rank r trains on pool[r::world]) -----------------------
    rng = np.random.default_rng(2)
    pool = None
    if layout == "c5":
        # BASELINE configs[4]: 1 M nodes, k uniform in {2..8}; the known set is built on the device and this
        # rank's shard is kept as int32 CSR; the positives of a step are expanded to the int64 rows the sampler / model take
        n_edges = edges or 100_000_000
        pool_all = synth.make_edges_device(N, n_edges, ks=tuple(ks), seed=5, device=device, zipf=zipf)
        csr_off, csr_ids = synth.edges_to_csr(pool_all, rank, world)
        M_shard = csr_off.numel() - 1
        g = torch.Generator(device=device)
        g.manual_seed(11 + rank)
        shard_w = torch.rand(M_shard, generator=g, device=device) * 0.4 + 0.6
        shard_w = shard_w / shard_w.mean() * neg_num
        workload_edges = n_edges
    else:
        pools = [np.pad(synth.make_edges_fast(rng, N, k, edges_per_k), ((0, 0), (0, L - k))) for k in ks]
        pool = np.concatenate(pools, axis=0)
        wts = rng.uniform(0.6, 1.0, size=len(pool)).astype(np.float32)      # quantile-transformed weights > cutoff 0.6 (main.py:555-556)
        wts = wts / wts.mean() * neg_num                                     # main.py:594-595
        perm = rng.permutation(len(pool))
        pool, wts = pool[perm], wts[perm]
        pool_all = torch.from_numpy(pool).to(device)
        shard = torch.from_numpy(np.ascontiguousarray(pool[rank::world])).to(device).long().contiguous()
        shard_w = torch.from_numpy(np.ascontiguousarray(wts[rank::world])).to(device).float().contiguous()
        M_shard = shard.shape[0]
        workload_edges = len(pool)

    hset = HyperedgeSet(pool_all)                                         # replicated exact set of known hyperedges
    sampler = NegativeSampler(hset, synth.node2chrom(num), synth.chrom_range(num), neg_num=neg_num, min_dis=0, seed=1234 + rank)

    clf = make_model(front_end, dim, num, device)
    clf.train()                                                           # dropout ON, as in the reference's training step
    trainer = Trainer(clf, lr=1e-3, base_seed=99, table_exchange=table_exchange, deterministic=deterministic)
    trainer.force_collectives = dist.launched and world == 1       # 1-rank torchrun: still go through RCCL
    trainer.time_collectives = world > 1 or dist.launched           # HIP events around every collective: the first multi-GPU run diagnoses itself

    x = torch.zeros((B, L), dtype=torch.long, device=device)
    y = torch.cat([torch.ones(P, device=device), torch.zeros(B - P, device=device)])      # main.py:444-445
    w = torch.ones(B, device=device)                                                        # main.py:446-447
    cursor = torch.zeros(1, dtype=torch.long, device=device)
    ar = torch.arange(P, device=device)
    n_chrom = len(num)
    chrom_rng = np.random.default_rng(7)

    step_no = torch.zeros(1, dtype=torch.long, device=device)
    lib_ = _lib.load()

    def one_step():
        rc = int(chrom_rng.integers(n_chrom)) if front_end == "adj" else 0
        if layout == "c5":
            idx = (cursor + ar) % M_shard                   # next P positives of this rank's (pre-shuffled) shard: CSR rows expanded to [P, L]
            cursor.add_(P)
            x[:P] = synth.csr_to_padded(csr_off, csr_ids, L, idx)
            torch.index_select(shard_w, 0, idx, out=w[:P])
            sampler.sample_into(x[:P], x[P:])
            return trainer.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=rc)     # phase-2 weighting (main.py:672-673)
        # the next P positives (wrapping around the shard), their weights and both RNG seeds + 1 in one launch: what matcha_amd.train's
        # captured step does (matcha_step_select, main.py:160-161)
        st_ = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        _lib.check(lib_.matcha_step_select(_lib.ptr(shard), _lib.ptr(shard_w), M_shard, L, _lib.ptr(step_no), P, _lib.ptr(x), _lib.ptr(w), None, 0,
                                           None, _lib.ptr(sampler.seed), _lib.ptr(trainer.seed), st_), "matcha_step_select")
        step_no.add_(1)
        sampler.sample_into(x[:P], x[P:], advance_seed=False)
        trainer.seed_advanced_by_caller = True
        try:
            return trainer.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=rc)     # phase-2 weighting (main.py:672-673)
        finally:
            trainer.seed_advanced_by_caller = False

    runner = one_step
    if graph and world == 1 and front_end == "table":
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(2):
                one_step()
        torch.cuda.current_stream(device).wait_stream(side)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            one_step()
        runner = cg.replay

    lib = _lib.load()
    for _ in range(warmup):
        runner()
    dist.barrier()

    # which kernel class is dominant?  measure every class over two steps each (outside the timed region)
    prof_cls = prof
    class_ms = {}
    class_work = {}
    if prof_cls == "auto" and not graph:
        for name, cid in _lib.PROF.items():
            lib.matcha_profile_select(cid)
            runner()
            runner()
            ms, n, wk = C.c_double(), C.c_int64(), C.c_double()
            _lib.check(lib.matcha_profile_read(C.byref(ms), C.byref(n), C.byref(wk)))
            if n.value:
                class_ms[name] = ms.value / 2.0
                class_work[name] = (wk.value / n.value, n.value / 2.0)       # work per launch (launch bound B*L + 1 rows), launches per step
        lib.matcha_profile_select(0)
        prof_cls = max(class_ms, key=class_ms.get) if class_ms else "none"
    if graph:
        prof_cls = "none"
    dist.barrier()

    if prof_cls != "none":
        lib.matcha_profile_select(_lib.PROF[prof_cls])
    # `windows` timed windows of EXACTLY `steps` steps each, every one bracketed by barrier + synchronize on both sides; the line
    # reports the median window (run-to-run noise of one 36 ms window exceeded the differences being resolved), min and max beside it
    win = []
    for _ in range(max(1, windows)):
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner()
        dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t[0])
        win.append(dt)
    elapsed = float(np.median(win))
    steps_timed = steps * len(win)
    roof = None
    if prof_cls != "none":
        ms, n, wk = C.c_double(), C.c_int64(), C.c_double()
        _lib.check(lib.matcha_profile_read(C.byref(ms), C.byref(n), C.byref(wk)))
        lib.matcha_profile_select(0)
        # The library sizes launches for the upper bound B*L + 1 token rows and counts work for that bound; the ragged
        # path only computes the real tokens (+ 1 shared padding token), so scale to the ALGORITHMIC work actually done.
        if prof_cls not in ("adamw", "neg_sample"):
            fill = (float((x != 0).sum().item()) + 1.0) / (B * L + 1.0)
            wk = C.c_double(wk.value * fill)
        if n.value and ms.value > 0:
            per_launch_ms = ms.value / n.value
            if prof_cls in GEMM_CLASSES:
                ach = wk.value / (ms.value * 1e-3) / 1e12
                roof = mfma_roofline(lib, prof_cls, dim, ach)
                roof.update(traffic=None, launches_per_step=n.value / steps_timed, avg_launch_ms=round(per_launch_ms, 5), work_per_launch=wk.value / n.value)
            else:
                ach = wk.value / (ms.value * 1e-3) / 1e9
                roof = dict(bound="hbm", kernel=prof_cls, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, launches_per_step=n.value / steps_timed,
                            avg_launch_ms=round(per_launch_ms, 5), work_per_launch=wk.value / n.value)
    if roof is not None:
        # HBM bytes per launch from committed PMC passes of this same command (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # runs, FETCH doubled per the gfx950 note; tools/collect_profiles.sh + summarize_profiles.py).  Quoted only when the
        # file was measured on THIS tree's kernels (fingerprint of matcha_amd/csrc) and on this workload; otherwise null.
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                with open(f) as fh:
                    pmc = json.load(fh)
                if pmc.get("csrc_sha16") == csrc_sha16() and pmc.get("workload") == [layout, dim, front_end, rows]:
                    c = pmc["classes"].get(prof_cls)
                    if c:
                        roof["traffic"] = round(c["hbm_bytes_per_launch"], 1)
                        roof["traffic_source"] = os.path.basename(f)
                    mf = os.path.join(os.path.dirname(f), os.path.basename(f).replace("pmc_traffic", "pmc_mfma"))
                    if "pmc" in roof and os.path.exists(mf):
                        # pipe occupancy of the same kernel from the committed counter passes (tools/summarize_profiles.py): matrix pipe, vector
                        # ALUs, their co-execution, LDS -- what says WHICH resource a kernel below its roof is waiting on
                        with open(mf) as fh2:
                            mm = json.load(fh2)
                        if mm.get("csrc_sha16") == csrc_sha16():
                            k_ = mm.get("kernels", {}).get(prof_cls, {})
                            roof["pmc"] = dict(k_.get("derived", {}), source=os.path.basename(mf)) if k_ else None
                    break
            except (OSError, ValueError, KeyError):
                pass
    # the model step alone (forward + backward [+ exchange] + AdamW on the last batch; no positive
    # gather, no negative sampling) -- reported beside the headline, never as `value`
    model_only_ms = None
    if model_only and not graph:
        k2 = max(1, steps)
        trainer.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=0)
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            trainer.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=0)
        dist.barrier()
        model_only_ms = (time.perf_counter() - t1) / k2 * 1e3
    trainer.check_status()
    coll = {}
    if trainer.time_collectives:
        # per-collective time on this rank (mean over the calls of the timed windows) and what every rank saw of the process group
        mine = {k: float(np.mean(v)) * 1e3 for k, v in trainer.collective_ms().items()}
        names = sorted(mine)
        t = torch.tensor([mine[k] for k in names] + [float(torch.distributed.get_world_size()), float(torch.cuda.current_device())],
                         dtype=torch.float64, device=device)
        allr = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(allr, t)
        coll = {"collective_us_per_call_by_rank": {k: [round(float(a[i]), 1) for a in allr] for i, k in enumerate(names)},
                "world_size_seen_by_rank": [int(a[len(names)]) for a in allr], "device_by_rank": [int(a[len(names) + 1]) for a in allr],
                "backend": torch.distributed.get_backend()}
    exhausted = sampler.check_status()
    losses = trainer.losses.cpu().tolist()
    # every MFMA- / HBM-bound kernel class against its roofline, from the same two-step measurement that picked the dominant one
    # (work scaled from the launch bound to the real tokens like `roofline`; the sampler's and AdamW's work is exact)
    fill_all = (float((x != 0).sum().item()) + 1.0) / (B * L + 1.0)
    roof_all = {}
    for name, ms_step in class_ms.items():
        wpl, lps = class_work[name]
        work = wpl * lps * (1.0 if name in ("adamw", "neg_sample") else fill_all)
        if ms_step <= 0 or work <= 0:
            continue
        if name in GEMM_CLASSES:
            ach = work / (ms_step * 1e-3) / 1e12
            r_ = mfma_roofline(lib, name, dim, ach)
            roof_all[name] = dict(bound="mfma", ms_per_step=round(ms_step, 4), achieved=r_["achieved"], unit="TFLOP/s", peak=r_["peak"], frac=r_["frac"],
                                  pipe="bf16 x3 planes" if r_["peak"] == MFMA_BF16_PEAK_TFLOPS else "f32", executed_fraction=r_["executed_fraction"],
                                  frac_reference_f32=r_["frac_reference_f32"])
        else:
            ach = work / (ms_step * 1e-3) / 1e9
            roof_all[name] = dict(bound="hbm", ms_per_step=round(ms_step, 4), achieved=round(ach, 1), unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))
            # SURVEY.md 8 d4 prices the gather at k (4 d + 8) READ bytes per row and the backward scatter at k 4 d bytes added; the fused front-end
            # kernels also write X (+ the pre-activation rows) / read the d x_hat, dXs, x0 rows, which `frac` counts: both figures, labelled
            tok = wpl / (B * L + 1.0)
            if name == "front_fwd" and tok > 0:
                roof_all[name]["frac_is"] = "bytes read + written per token (ids, table row, attribute row, X [, x0]) / time / 8 TB/s"
                roof_all[name]["frac_d4_read_bytes"] = round(ach * (4.0 * dim + 8.0) / tok / HBM_PEAK_GBS, 4)
            if name == "front_bwd" and tok > 0:
                roof_all[name]["frac_is"] = "bytes read + written per token (d x_hat slabs, dXs, x0, X, ids, attribute row, gradient row) / time / 8 TB/s"
                roof_all[name]["frac_d4_scatter_bytes"] = round(ach * (4.0 * dim) / tok / HBM_PEAK_GBS, 4)
            if name == "adamw":
                # SURVEY §8 d4 prices the update at 28 B per element (p, g, m, v read; p, m, v written); the kernel also zeroes g (zero_grad): 32 B moved
                roof_all[name]["frac_incl_grad_zeroing_32B"] = round(ach * 32.0 / 28.0 / HBM_PEAK_GBS, 4)
    out = dict(B=B, P=P, L=L, N=N, elapsed=elapsed, windows=win, roof=roof, roof_all=roof_all, class_ms=class_ms, model_only_ms=model_only_ms, losses=losses,
               pool=pool, wts=None if pool is None else wts, num=num, known_edges=workload_edges, sparse_exchange=bool(trainer._sparse),
               exhausted_negatives=exhausted, comm_bytes=dict(trainer.comm_bytes), overlap=bool(trainer._side is not None and trainer._overlap()),
               collectives=coll)
    del trainer, clf, sampler, hset, pool_all
    gc.collect()
    torch.cuda.empty_cache()
    return out


def gather_roofline(device):
    """The embedding-row gather alone (Wrap_Embedding.forward / get_node_embeddings, Modules.py:33-34, :252-259) through the C ABI
    with uniform random ids; gather_rows_kernel timed live with HIP events on its stream (back-to-back launches, one event pair).  ALGORITHMIC bytes per row = 4 d + 8 read
    (SURVEY.md §8 d4); the rows are also written (4 d) because this surface materialises them, so the kernel as a whole is bound by
    the ~6.3 TB/s read + write copy ceiling: `frac` (read bytes against the 8 TB/s read roof) tops out near 0.39 for an
    HBM-resident table, and `copy_frac` says how close the kernel is to THAT ceiling."""
    lib = _lib.load()
    out = []
    for name, N, d, T, resident in (("1M x 64 (244 MiB table: Infinity-Cache sized)", 1 << 20, 64, 1 << 24, "mall"),
                                    ("16M x 64 (4 GiB table)", 1 << 24, 64, 1 << 24, "hbm"),
                                    ("C5 1M x 256 (1 GiB table)", 1 << 20, 256, 1 << 22, "hbm"),
                                    # the same two HBM-resident tables at the launch size a training step has (its token count): the
                                    # gathered rows (80 / 128 MiB) then stay in the 256 MiB Infinity Cache instead of streaming out
                                    ("16M x 64 (4 GiB table), one step's tokens per launch (65536 x 5)", 1 << 24, 64, 65536 * 5, "hbm"),
                                    ("C5 1M x 256 (1 GiB table), one C5 step's tokens per launch (16384 x 8)", 1 << 20, 256, 16384 * 8, "hbm")):
        table = torch.randn(N + 1, d, device=device)
        ids = torch.randint(1, N + 1, (T,), device=device, dtype=torch.int64)
        rows = torch.empty(T, d, device=device)
        shp = _lib.Shape(d, 1, N, 1, 0, 0)
        par, fro = _lib.Tensors(), _lib.Frozen()
        par.table = table.data_ptr()
        st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

        def run():
            _lib.check(lib.matcha_node_embeddings(C.byref(shp), C.byref(par), C.byref(fro), _lib.ptr(ids), T, _lib.ptr(rows), None, 0, None, st), "gather")
        for _ in range(3):
            run()
        torch.cuda.synchronize(device)
        # one HIP event pair around `reps` back-to-back launches on the stream they are launched on (torch's current stream here): a
        # per-launch event pair costs ~3 us, which is 10 % of a 30 us step-sized launch
        reps = 10 if T >= (1 << 22) else 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize(device)
        assert torch.equal(rows[:4096], table[ids[:4096]])
        t = e0.elapsed_time(e1) * 1e-3 / reps
        read = T * (4.0 * d + 8.0)
        out.append(dict(table=name, d=d, rows_per_launch=T, resident=resident, bound="hbm", kernel="gather_rows", avg_launch_ms=round(t * 1e3, 4),
                        achieved=round(read / t / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(read / t / 1e9 / HBM_PEAK_GBS, 4),
                        read_plus_write_gbs=round((read + T * 4.0 * d) / t / 1e9, 1),
                        copy_frac=round((read + T * 4.0 * d) / t / 1e9 / HBM_COPY_GBS, 4)))
        del table, ids, rows
        torch.cuda.empty_cache()
    return out


def visible_gpus() -> int:
    """Number of GPUs a child would see, WITHOUT a HIP / HSA call in this process (torch.cuda.device_count() falls back to
    hipGetDeviceCount on ROCm builds without amdsmi, which opens the GPU): the KFD topology lists one node per agent, GPUs are the nodes
    with SIMDs; *_VISIBLE_DEVICES narrows it."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(f) as fh:
                for line in fh:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    return n


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed.run launch: start the N ranks ourselves -- one process per GPU
    over RCCL, exactly the command the driver would have used -- as a CHILD process, before anything in this process has touched
    the GPU (no HIP call, no torch.cuda.is_available(); the devices are counted from the KFD topology, visible_gpus), and hand its exit code back.
    Rank 0's JSON line reaches stdout through the inherited descriptor.  Never falls back to one rank."""
    import socket
    import subprocess
    n_dev = visible_gpus()
    if n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible; refusing to report a {args.gpus}-GPU line", file=sys.stderr)
        return 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def front_gather_roofline(device):
    """The gather the model step actually executes, on tables that do not fit any cache (north_star: "coalesced HBM reads of embedding
    rows into LDS tiles"): (a) embed_dim 64 -- front_fwd_kernel gathers the node rows and attribute rows of a 64-token tile straight into
    LDS, runs attribute_nn and next_w on them and writes only X (an inference forward; the training forward also writes the
    pre-activation rows) -- on a 16 M x 64 table; (b) embed_dim 256 -- embed_fwd_kernel (gather + attribute_nn, two tokens in flight per
    16-lane group) on BASELINE configs[4]'s 1 M x 256 table.  Timed live with HIP events through the library's per-class profiler inside
    real forward calls; achieved = tokens * (8 + 4 d + 4 n_attr) READ bytes / time against the 8 TB/s HBM-read roof."""
    import Modules as M
    lib = _lib.load()
    out = []
    for name, num, d, B, L, cls in (("front_fwd (fused front end, d = 64) on a 16 M x 64 table (4 GiB)", [(1 << 24) // 23] * 23, 64, 65536, 5, "front_fwd"),
                                    ("embed_fwd (d = 256) on the C5 table, 1 M x 256 (1 GiB)", synth.LAYOUTS["c5"], 256, 16384, 8, "embed_fwd")):
        N = int(np.sum(num))
        clf = make_model("table", d, num, device).eval()
        n_attr = len(num) + 1
        g = torch.Generator(device=device)
        g.manual_seed(3)
        x = torch.randint(1, N + 1, (B, L), generator=g, device=device, dtype=torch.int64)
        reps = 10
        with torch.no_grad(), clf.deferred_id_check():
            for _ in range(2):
                clf(x)
            lib.matcha_profile_select(_lib.PROF[cls])
            for _ in range(reps):
                clf(x)
            ms, n, wk = C.c_double(), C.c_int64(), C.c_double()
            _lib.check(lib.matcha_profile_read(C.byref(ms), C.byref(n), C.byref(wk)))
            lib.matcha_profile_select(0)
        torch.cuda.synchronize(device)
        tokens = B * L + 1
        t = ms.value * 1e-3 / max(n.value, 1)
        # attribute rows: both kernels rebuild them from the node id when the table has get_attributes' structure (attr_mode 1: ONE random
        # row per token; front_fwd2_kernel since round 6); any other table is read as rows padded to one 128-byte fetch unit (csrc/attr_src.hpp)
        attr_read = 0 if clf._runtime().attr_mode == 1 else 4 * n_attr
        read = tokens * (8.0 + 4.0 * d + attr_read)
        written = tokens * 4.0 * d
        rec = dict(table=name, d=d, tokens_per_launch=tokens, resident="hbm", bound="hbm", kernel=cls, avg_launch_ms=round(t * 1e3, 4),
                   achieved=round(read / t / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(read / t / 1e9 / HBM_PEAK_GBS, 4),
                   read_bytes_per_token=8 + 4 * d + attr_read, written_bytes_per_token=4 * d,
                   # SURVEY.md 8 d4's own figure: k (4 d + 8) read bytes per row = (4 d + 8) per token -- the node row and its int64 id only
                   frac_d4_read_bytes=round(tokens * (4.0 * d + 8.0) / t / 1e9 / HBM_PEAK_GBS, 4),
                   read_plus_write_gbs=round((read + written) / t / 1e9, 1),
                   # the kernel moves as many bytes out as in: the measured copy ceiling (6.29 TB/s read + write) is its HBM bound,
                   # i.e. frac <= ~0.39-0.40 whatever the access pattern
                   copy_frac=round((read + written) / t / 1e9 / HBM_COPY_GBS, 4))
        if cls == "front_fwd":
            # ... and it is not a pure gather: attribute_nn (K = 32) and next_w (K = 64) run on the gathered tile, 2 x 64 x 96 flop per
            # token = 24 flop per byte moved, the machine's own balance (157 TFLOP/s : 6.3 TB/s): both roofs bind at once
            fl = tokens * 2.0 * 64 * (32 + 64)
            rec["mfma_tflops"] = round(fl / t / 1e12, 2)
            rec["mfma_frac"] = round(fl / t / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
        out.append(rec)
        del clf, x
        gc.collect()
        torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    dist = Dist()
    if args.gpus != dist.world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {dist.world} (launch with `python bench.py --gpus N`, or with "
                         f"torch.distributed.run --nproc-per-node N bench.py --gpus N)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(dist.local_rank)
    if dist.world > 1 or dist.launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dist.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dist.device)
        else:
            torch.distributed.init_process_group(dist.backend)
    world, rank = dist.world, dist.rank
    ks = [int(v) for v in args.ks.split(",")]

    m = run_workload(dist, layout=args.layout, dim=args.dim, ks=ks, rows=args.rows, front_end=args.front_end, steps=args.steps,
                     warmup=args.warmup, prof=args.prof, graph=args.graph, edges_per_k=args.edges_per_k, edges=args.edges,
                     table_exchange=args.table_exchange, deterministic=args.deterministic, zipf=args.zipf, windows=args.windows)
    B, P, L, N = m["B"], m["P"], m["L"], m["N"]
    elapsed = m["elapsed"]
    result = {
        "metric": METRIC,
        "value": round(B * world * args.steps / elapsed, 1),
        "unit": "hyperedges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "timed_windows": {"n": len(m["windows"]), "steps_each": args.steps, "statistic": "median",
                          "ms_per_step_min": round(min(m["windows"]) / args.steps * 1e3, 4),
                          "ms_per_step_max": round(max(m["windows"]) / args.steps * 1e3, 4)},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "dtype_note": "f32 in, f32 accumulate; the matrix products run as 3 x bf16 split-operand MFMAs (fp32-accurate: tests/test_cpu_bf16x3.py)",
        "data": "synthetic",
        "config": {"workload": f"{args.layout} bins (N={N}), k in {{{args.ks}}} mixed-k zero-padded to L={L}, embed_dim={args.dim}, "
                               f"front end={args.front_end}, neg_num=3, dropout on, AdamW lr=1e-3, {m['known_edges']} known hyperedges",
                   "rows_per_gpu_per_step": B, "positives_per_gpu_per_step": P, "global_rows_per_step": B * world,
                   "parallelism": f"dp{world}", "hipgraph": bool(args.graph),
                   "table_gradient_exchange": "row-sparse all-gather" if m["sparse_exchange"] else ("flat all-reduce" if (world > 1 or dist.launched) else "none"),
                   "embedding_backward": "sorted, one writer per row (bitwise reproducible)" if args.deterministic else "float atomics",
                   # bytes each collective of one step moves per rank (payload; a ring all-reduce sends and receives 2 (N-1)/N of it,
                   # the all-gather receives what is listed), and whether the encoder part overlaps the front-end backward
                   "collective_payload_bytes_per_step": m["comm_bytes"], "exchange_overlapped": m["overlap"], "collectives": m["collectives"]},
        "positives_per_s": round(P * world * args.steps / elapsed, 1),
        "last_bce": round(m["losses"][0], 5),
        "model_step_only": None if m["model_only_ms"] is None else {"ms_per_step": round(m["model_only_ms"], 4),
                                                                     "hyperedges_per_s": round(B * world / (m["model_only_ms"] * 1e-3), 1)},
        "roofline": m["roof"],
        "kernel_class_ms_per_step": {k: round(v, 4) for k, v in sorted(m["class_ms"].items(), key=lambda kv: -kv[1])},
        "roofline_by_kernel_class": m["roof_all"],
        "csrc_sha16": csrc_sha16(),
    }
    default_run = (args.layout, args.dim, args.front_end, args.rows, args.ks) == ("hg38_1mb", 64, "table", 65536, "2,3,4,5")
    if rank == 0 and world == 1 and not args.no_extras and not args.graph:
        result["roofline_gather"] = gather_roofline(dist.device)
        result["roofline_gather_in_step"] = front_gather_roofline(dist.device)
        if default_run:
            extras = {}
            for key, kw in (("deterministic_embedding_backward", dict(layout="hg38_1mb", dim=64, ks=[2, 3, 4, 5], rows=65536, front_end="table", deterministic=True)),
                            ("adj_front_end_configs2", dict(layout="hg38_1mb", dim=64, ks=[2, 3, 4, 5], rows=65536, front_end="adj", prof="auto")),
                            ("reference_batch_384_rows", dict(layout="hg38_1mb", dim=64, ks=[2, 3, 4, 5], rows=384, front_end="table")),
                            # the same launch-bound step replayed from ONE hipGraph (Trainer.capture path: nothing in the step allocates or synchronises)
                            ("reference_batch_384_rows_hipgraph", dict(layout="hg38_1mb", dim=64, ks=[2, 3, 4, 5], rows=384, front_end="table", graph=True)),
                            # the reference's own set-up (main.py:609-613 builds MultipleEmbedding = the adj front end; 96 positives x (1 + 3) rows per step;
                            # 80 ms per such step on 8 CPU cores)
                            ("reference_batch_384_rows_adj", dict(layout="hg38_1mb", dim=64, ks=[2, 3, 4, 5], rows=384, front_end="adj")),
                            ("configs3_hg38_100kb_d128", dict(layout="hg38_100kb", dim=128, ks=[2, 3, 4, 5], rows=65536, front_end="table", prof="auto")),
                            # the same configuration in the mode the reference itself would run there (MultipleEmbedding: per-chromosome gather-GEMMs
                            # over up to 2 491 feature columns + the reconstruction branch); embed_dim 128 takes the unfused adj kernels (adj_frontend.hip) in front of
                            # the fused attention block (enc128.hip)
                            ("configs3_hg38_100kb_d128_adj", dict(layout="hg38_100kb", dim=128, ks=[2, 3, 4, 5], rows=65536, front_end="adj", prof="auto")),
                            # BASELINE configs[4] at its FULL size: 1 M nodes, 100 M known hyperedges (generated, hashed and CSR-sharded on the
                            # device), k in {2..8}, d = 256.  kernel classes on: the HBM-bound part of this step is the dense AdamW over
                            # the 1 M x 256 table + the gather + the scatter (hbm_bound_share)
                            ("configs4_c5_1M_nodes_d256", dict(layout="c5", dim=256, ks=[2, 3, 4, 5, 6, 7, 8], rows=16384, front_end="table",
                                                                edges=100_000_000, prof="auto")),
                            # a small batch on the same table: the encoder shrinks with the rows, the 7.2 GB AdamW stream over the table
                            # does not -- the configuration in which config 5 IS the HBM-bound stress its name promises
                            ("configs4_c5_2048_rows", dict(layout="c5", dim=256, ks=[2, 3, 4, 5, 6, 7, 8], rows=2048, front_end="table",
                                                            edges=10_000_000, prof="auto")),
                            # Zipf(1.0) node ids: the cached regime of the gather and the contended one of the scatter
                            ("configs4_c5_zipf_ids", dict(layout="c5", dim=256, ks=[2, 3, 4, 5, 6, 7, 8], rows=16384, front_end="table",
                                                           edges=10_000_000, zipf=True))):
                kw = dict(kw)
                kw.setdefault("prof", "none")
                try:
                    e = run_workload(dist, steps=8, warmup=3, model_only=False, windows=3, **kw)
                except Exception as exc:      # an extra point never costs the headline line
                    extras[key] = {"error": repr(exc)[:300]}
                    gc.collect()
                    torch.cuda.empty_cache()
                    continue
                extras[key] = {"hyperedges_per_s": round(e["B"] * 8 / e["elapsed"], 1), "ms_per_step": round(e["elapsed"] / 8 * 1e3, 4),
                               "rows_per_step": e["B"], "known_hyperedges": e["known_edges"], "steps": 8, "windows": 3,
                               "exhausted_negatives": e["exhausted_negatives"]}
                if e["class_ms"]:
                    top = sorted(e["class_ms"].items(), key=lambda kv: -kv[1])[:6]
                    extras[key]["kernel_class_ms_per_step"] = {k: round(v, 4) for k, v in top}
                    hb = sum(e["class_ms"].get(k, 0.0) for k in ("adamw", "embed_fwd", "embed_scatter", "front_fwd", "front_bwd"))
                    extras[key]["hbm_bound_share"] = round(hb / (e["elapsed"] / 8 * 1e3), 4)
                    extras[key]["roofline_by_kernel_class"] = {k: v for k, v in e["roof_all"].items() if k in ("adamw", "embed_fwd", "embed_scatter", "adj_encode", "adj_recon",
                                                                                                            "adj_bwd", "gemm_nt", "gemm_nn", "gemm_tn", "attn_fwd", "attn_bwd", "fused_fwd", "fused_bwd")}
            # the reference's DRIVER flow on the clock (main.py:119-197, :261-342; tools/epoch_bench.py): matcha_amd.train running a phase-2
            # epoch at the reference's batch (96 + 288 rows per step) -- DataGenerator, batch assembly, sampler, step, metrics, save_embeddings;
            # a quarter epoch here (250 steps per size), the full 4 x 1000 steps with the CPU estimate in profiles/rNN_epoch_bench.jsonl
            try:
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                import epoch_bench
                extras["train_driver_epoch"] = {fe: epoch_bench.one_case(fe, 250, True) for fe in ("table", "adj")}
            except Exception as exc:      # the driver flow is an extra: never lose the headline line over it
                extras["train_driver_epoch"] = {"error": repr(exc)[:200]}
            result["extra_points"] = extras
    if rank == 0 and world == 1 and not args.no_cpu_baseline and m["pool"] is not None:
        result["cpu_baseline"] = cpu_baseline(args, m["num"], ks, L, m["pool"], m["wts"], 3)
    if rank == 0:
        print(json.dumps(result, ensure_ascii=False))
    if world > 1 or dist.launched:
        torch.distributed.destroy_process_group()


def cpu_baseline(args, num, ks, L, pool, wts, neg_num):
    """The reference's step on the host cores, via the oracle port (oracle/ is the checker/baseline only):
    python negative sampling + PyTorch-CPU forward/backward/AdamW at the reference's own batch (96 positives + 288
    negatives, main.py:527-528) and at a large batch, bounded to ~args.cpu_seconds in total (and at least 20 timed steps per point at
    the reference's batch).  Thread sweep 8 / 16 / 32 / 64 (torch.set_num_threads); `value` is the fastest point, `cores` its threads."""
    from oracle import hypersagnn as O
    from oracle import sampler as OS
    attr = attribute_table(num)
    sd = synth.make_state_dict(np.random.default_rng(0), num, args.dim, "table" if args.front_end == "table" else "adj", attr)
    if args.front_end == "table":
        fe = O.FrontEnd(mode="table", bounds=synth.bounds(num))
    else:
        intra, inter = synth.make_adjacency(np.random.default_rng(2), num)
        fe = O.FrontEnd(mode="adj", bounds=synth.bounds(num), feats=[torch.from_numpy(f) for f in O.corrcoef_features(intra, synth.chrom_range(num))],
                        inter=torch.from_numpy(O.zscore_inter(inter)))
    known = {tuple(int(v) for v in r if v) for r in pool}             # the FULL known set, as the device sampler sees it
    n2c, cr = synth.node2chrom(num), synth.chrom_range(num)
    n_cpu = os.cpu_count() or 1
    # thread sweep (a 256-thread intra-op pool on [384, 64]-sized operands is pure oversubscription: 22 rows/s in round 3's one-step
    # sample -- dropped): 8 / 16 / 32 / 64 threads where the host has them, >= 20 steps each at the reference's batch
    thread_cfgs = sorted({t for t in (8, 16, 32, 64) if t <= n_cpu} or {n_cpu})
    best, notes, best_cores = 0.0, [], n_cpu
    budget = args.cpu_seconds / (2 * len(thread_cfgs))
    for cores in thread_cfgs:
        torch.set_num_threads(cores)
        P_ = {k: torch.from_numpy(np.array(v)).requires_grad_(not k.startswith("attribute_dict")) for k, v in sd.items()}
        opt = O.AdamWRef()
        rng = np.random.default_rng(5)
        for pos_n, label, min_steps in ((96, "B=384 (reference batch)", 21), (2048, "B=8192", 4)):
            t_spent, rows, steps = 0.0, 0, 0
            while (t_spent < budget or steps < min_steps) and steps < 200:
                sel = rng.integers(0, len(pool), size=pos_n)
                t0 = time.perf_counter()
                neg = OS.sample_negatives(pool[sel], known, n2c, cr, neg_num, 0, seed=steps)
                xb, yb, wb = OS.assemble_batch(pool[sel], wts[sel], neg)
                T = xb.size
                masks = {"fc1": torch.from_numpy((rng.random((T, args.dim)) >= O.P_DROP_FC1).astype(np.float32) / (1 - O.P_DROP_FC1)),
                         "pff": torch.from_numpy((rng.random((T, args.dim)) >= O.P_DROP_PFF).astype(np.float32) / (1 - O.P_DROP_PFF))}
                O.train_step(P_, fe, opt, torch.from_numpy(xb), torch.from_numpy(yb), torch.from_numpy(wb), 1.0, 0.001,
                             random_chrom=int(rng.integers(len(num))), masks=masks)
                dt = time.perf_counter() - t0
                if steps > 0:                      # first step = warm-up
                    t_spent += dt
                    rows += len(xb)
                steps += 1
            rate = rows / t_spent if t_spent > 0 else 0.0
            notes.append(f"{cores} threads, {label}: {rate:.0f} rows/s over {steps - 1} steps")
            if rate > best:
                best, best_cores = rate, cores
    return {"value": round(best, 1), "unit": "hyperedges/s", "cores": best_cores, "host_cores": n_cpu, "kind": "port",
            "sample": "oracle port of the reference step (python negative sampling against the full known set + PyTorch-CPU fwd/bwd/AdamW, "
                      "dropout on); " + "; ".join(notes) + f"; torch {torch.__version__}"}


if __name__ == "__main__":
    main()
