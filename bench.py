#!/usr/bin/env python3
"""Benchmark of the hyperedge-classifier training step on MI355X (BASELINE.json metric:
"training hyperedges/sec at k in {2..5}, embed_dim=64; 1/2/4/8 MI355X").

One step = one pass of the hot path over one batch resident in HBM: draw a batch of positives from the
device-resident pool, generate neg_num=3 negatives per positive on the GPU (main.py:361-459), forward + weighted
BCE, backward, [RCCL all-reduce of the flat gradient], fused AdamW -- i.e. main.py:155-183 for one batch.  Rows =
positives + negatives, every one of which goes through forward+backward+update.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel class, timed live with HIP events on the
launch stream inside the timed region (matcha_profile_select/_read); `cpu_baseline` is the oracle ("port" of the
reference's PyTorch-CPU step, pinned to the reference by tests/golden) timed on this box's host cores on a bounded
sample (rank 0, N=1 only).
"""
import argparse
import ctypes as C
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
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA peak (no xf32/TF32 on gfx950)
# MFMA-bound kernel classes.  The fused kernels count ALGORITHMIC GEMM flops only: fused_fwd 4 projections per
# head + the two pff GEMMs; fused_bwd 8 GEMMs per head (dO, dWfc1, 3 dW', 3 d x_hat terms) -- the Q/K/V recompute of the
# backward kernel and the O(k) attention arithmetic are not counted.
GEMM_CLASSES = ("gemm_nt", "gemm_nn", "gemm_tn", "fused_fwd", "fused_bwd")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=65536, help="rows (positives+negatives) per GPU per step")
    ap.add_argument("--front-end", choices=["table", "adj"], default="table")
    ap.add_argument("--layout", default="hg38_1mb")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--ks", default="2,3,4,5")
    ap.add_argument("--edges-per-k", type=int, default=100000)
    ap.add_argument("--prof", default="auto", help="kernel class for the live roofline (see matcha_amd/_lib.py PROF) or 'none'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (single GPU)")
    return ap.parse_args()


def attribute_table(num):
    """[N+1, C+1] one-hot chromosome || normalised position, row 0 zeros (what main.py:497-512 builds)."""
    C_ = len(num)
    rows = []
    for i, n in enumerate(num):
        a = np.zeros((n, C_ + 1), dtype=np.float32)
        a[:, i] = 1.0
        a[:, C_] = np.arange(n, dtype=np.float32) / np.float32(num[0])
        rows.append(a)
    return np.concatenate([np.zeros((1, C_ + 1), np.float32)] + rows, axis=0)


def make_model(args, num, device):
    import Modules as M
    d = args.dim
    N = int(np.sum(num))
    torch.manual_seed(0)
    if args.front_end == "table":
        ne = M.Wrap_Embedding(N + 1, d, padding_idx=0)
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


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # torch.distributed.run / torchrun
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=device)

    from matcha_amd.engine import Trainer
    from matcha_amd.sampler import HyperedgeSet, NegativeSampler

    num = synth.LAYOUTS[args.layout]
    N = int(np.sum(num))
    ks = [int(v) for v in args.ks.split(",")]
    L = max(ks)
    neg_num = 3                                   # main.py:527
    P = args.rows // (1 + neg_num)                # positives per GPU per step
    B = P * (1 + neg_num)

    # ---- synthetic positives (same pool on every rank; rank r trains on pool[r::world]) -----------------------
    rng = np.random.default_rng(2)
    pools = [np.pad(synth.make_edges_fast(rng, N, k, args.edges_per_k), ((0, 0), (0, L - k))) for k in ks]
    pool = np.concatenate(pools, axis=0)
    wts = rng.uniform(0.6, 1.0, size=len(pool)).astype(np.float32)      # quantile-transformed weights > cutoff 0.6 (main.py:555-556)
    wts = wts / wts.mean() * neg_num                                     # main.py:594-595
    perm = rng.permutation(len(pool))
    pool, wts = pool[perm], wts[perm]
    pool_all = torch.from_numpy(pool).to(device)
    shard = torch.from_numpy(pool[rank::world]).to(device)
    shard_w = torch.from_numpy(wts[rank::world]).to(device)
    M_shard = shard.shape[0]

    hset = HyperedgeSet(pool_all)                                         # replicated exact set of known hyperedges
    sampler = NegativeSampler(hset, synth.node2chrom(num), synth.chrom_range(num), neg_num=neg_num, min_dis=0, seed=1234 + rank)

    clf = make_model(args, num, device)
    clf.train()                                                           # dropout ON, as in the reference's training step
    trainer = Trainer(clf, lr=1e-3, base_seed=99 + rank)
    trainer.force_collectives = launched and world == 1        # 1-rank torchrun: still go through RCCL

    x = torch.zeros((B, L), dtype=torch.long, device=device)
    y = torch.cat([torch.ones(P, device=device), torch.zeros(B - P, device=device)])      # main.py:444-445
    w = torch.ones(B, device=device)                                                        # main.py:446-447
    cursor = torch.zeros(1, dtype=torch.long, device=device)
    ar = torch.arange(P, device=device)
    n_chrom = len(num)
    chrom_rng = np.random.default_rng(7)

    def one_step():
        idx = (cursor + ar) % M_shard                       # next P positives of this rank's (pre-shuffled) shard
        cursor.add_(P)
        torch.index_select(shard, 0, idx, out=x[:P])
        torch.index_select(shard_w, 0, idx, out=w[:P])
        sampler.sample_into(x[:P], x[P:])
        rc = int(chrom_rng.integers(n_chrom)) if args.front_end == "adj" else 0
        return trainer.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=rc)     # phase-2 weighting (main.py:672-673)

    runner = one_step
    if args.graph and world == 1 and args.front_end == "table":
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(2):
                one_step()
        torch.cuda.current_stream(device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            one_step()
        runner = graph.replay

    def barrier():
        if world > 1 or launched:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    lib = _lib.load()
    for _ in range(args.warmup):
        runner()
    barrier()

    # which kernel class is dominant?  measure every class over two steps each (outside the timed region)
    prof_cls = args.prof
    class_ms = {}
    if prof_cls == "auto" and not args.graph:
        for name, cid in _lib.PROF.items():
            lib.matcha_profile_select(cid)
            runner()
            runner()
            ms, n, wk = C.c_double(), C.c_int64(), C.c_double()
            _lib.check(lib.matcha_profile_read(C.byref(ms), C.byref(n), C.byref(wk)))
            if n.value:
                class_ms[name] = ms.value / 2.0
        lib.matcha_profile_select(0)
        prof_cls = max(class_ms, key=class_ms.get) if class_ms else "none"
    if args.graph:
        prof_cls = "none"
    barrier()

    if prof_cls != "none":
        lib.matcha_profile_select(_lib.PROF[prof_cls])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner()
    barrier()
    elapsed = time.perf_counter() - t0
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
                roof = dict(bound="mfma", kernel=prof_cls, achieved=round(ach, 3), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=round(ach / MFMA_F32_PEAK_TFLOPS, 4), traffic=None, launches_per_step=n.value / args.steps,
                            avg_launch_ms=round(per_launch_ms, 5), work_per_launch=wk.value / n.value)
            else:
                ach = wk.value / (ms.value * 1e-3) / 1e9
                roof = dict(bound="hbm", kernel=prof_cls, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, launches_per_step=n.value / args.steps,
                            avg_launch_ms=round(per_launch_ms, 5), work_per_launch=wk.value / n.value)
    if roof is not None:
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE; profiles/): separate
        # rocprofv3 --pmc runs of this same command, see profiles/r01_pmc_traffic.json
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)["classes"].get(prof_cls)
            if pmc and args.rows == 65536 and args.front_end == "table":
                roof["traffic"] = round(pmc["hbm_bytes_per_launch"], 1)
        except (OSError, ValueError, KeyError):
            pass
    # the model step alone (forward + backward [+ all-reduce] + AdamW on the last batch; no positive
    # gather, no negative sampling) -- reported beside the headline, never as `value`
    model_only_ms = None
    if not args.graph:
        k2 = max(1, args.steps)
        trainer.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=0)
        barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            trainer.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=0)
        barrier()
        model_only_ms = (time.perf_counter() - t1) / k2 * 1e3
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0])

    losses = trainer.losses.cpu().tolist()
    result = {
        "metric": "training hyperedges/sec at k∈{2..5}, embed_dim=64; 1/2/4/8 MI355X",
        "value": round(B * world * args.steps / elapsed, 1),
        "unit": "hyperedges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.layout} bins (N={N}), k in {{{args.ks}}} mixed-k zero-padded to L={L}, embed_dim={args.dim}, "
                               f"front end={args.front_end}, neg_num=3, dropout on, AdamW lr=1e-3",
                   "rows_per_gpu_per_step": B, "positives_per_gpu_per_step": P, "global_rows_per_step": B * world,
                   "parallelism": f"dp{world}", "hipgraph": bool(args.graph)},
        "positives_per_s": round(P * world * args.steps / elapsed, 1),
        "last_bce": round(losses[0], 5),
        "model_step_only": None if model_only_ms is None else {"ms_per_step": round(model_only_ms, 4), "hyperedges_per_s": round(B * world / (model_only_ms * 1e-3), 1)},
        "roofline": roof,
        "kernel_class_ms_per_step": {k: round(v, 4) for k, v in sorted(class_ms.items(), key=lambda kv: -kv[1])},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, num, ks, L, pool, wts, neg_num)
    if rank == 0:
        print(json.dumps(result, ensure_ascii=False))
    if world > 1 or launched:
        torch.distributed.destroy_process_group()


def cpu_baseline(args, num, ks, L, pool, wts, neg_num):
    """The reference's step on the host cores, via the oracle port (oracle/ is the checker/baseline only):
    python negative sampling + PyTorch-CPU forward/backward/AdamW at the reference's own batch (96 positives + 288
    negatives, main.py:527-528) and at a large batch, bounded to ~args.cpu_seconds in total."""
    from oracle import hypersagnn as O
    from oracle import sampler as OS
    # A 256-thread intra-op pool on [384, 64]-sized operands is slower than a handful of threads (oversubscription),
    # so the port runs with the thread count the reference's own survey timing used (8) unless the host has fewer.
    cores = min(8, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    attr = attribute_table(num)
    sd = synth.make_state_dict(np.random.default_rng(0), num, args.dim, "table" if args.front_end == "table" else "adj", attr)
    P_ = {k: torch.from_numpy(np.array(v)).requires_grad_(not k.startswith("attribute_dict")) for k, v in sd.items()}
    if args.front_end == "table":
        fe = O.FrontEnd(mode="table", bounds=synth.bounds(num))
    else:
        intra, inter = synth.make_adjacency(np.random.default_rng(2), num)
        fe = O.FrontEnd(mode="adj", bounds=synth.bounds(num), feats=[torch.from_numpy(f) for f in O.corrcoef_features(intra, synth.chrom_range(num))],
                        inter=torch.from_numpy(O.zscore_inter(inter)))
    known = {tuple(int(v) for v in r if v) for r in pool[:20000]}
    n2c, cr = synth.node2chrom(num), synth.chrom_range(num)
    opt = O.AdamWRef()
    rng = np.random.default_rng(5)
    best, notes = 0.0, []
    budget = args.cpu_seconds
    for pos_n, label in ((96, "B=384 (reference batch)"), (2048, "B=8192")):
        t_spent, rows, steps = 0.0, 0, 0
        while t_spent < budget / 2 and steps < 200:
            sel = rng.integers(0, 20000, size=pos_n)
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
        notes.append(f"{label}: {rate:.0f} rows/s over {steps - 1} steps")
        best = max(best, rate)
    return {"value": round(best, 1), "unit": "hyperedges/s", "cores": cores, "kind": "port",
            "sample": "oracle port of the reference step (python negative sampling + PyTorch-CPU fwd/bwd/AdamW, dropout on), "
                      + "; ".join(notes) + f"; torch {torch.__version__}, {cores} threads"}


if __name__ == "__main__":
    main()
