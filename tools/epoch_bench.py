#!/usr/bin/env python3
"""The reference's driver flow on the clock (main.py:119-197, :261-342): wall-clock of ONE phase-2 epoch of matcha_amd.train at the
reference's own settings -- 96 positives + 288 negatives per step, 1000 steps per hyperedge size, DataGenerator, device sampler, metrics,
save_embeddings -- on BASELINE configs[2]'s workload (hg38 1 Mb, k in {2..5}, embed_dim 64), for both front ends, with the epoch loop
replaying one captured step (default) and enqueuing every step call by call (MATCHA_TRAIN_GRAPH=0 semantics).  Prints one JSON line per
case: epoch wall clock, per-step wall clock, the device time of a step (HIP events around replays) and the host's share.

    python tools/epoch_bench.py [--steps-per-k 1000] [--front-ends table,adj] [--cpu-steps 40]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from matcha_amd import synth, train as T


def one_case(front_end, steps_per_k, graph, ks=(2, 3, 4, 5), d=64, layout="hg38_1mb"):
    dev = torch.device("cuda", 0)
    num = synth.LAYOUTS[layout]
    N, L = int(np.sum(num)), max(ks)
    rng = np.random.default_rng(2)
    pools = [np.pad(synth.make_edges_fast(rng, N, k, 100000), ((0, 0), (0, L - k))) for k in ks]
    edges = np.concatenate(pools, axis=0)
    weights = rng.uniform(0.6, 1.0, size=len(edges)).astype(np.float32)
    weights = weights / weights.mean() * T.NEG_NUM
    np.random.seed(1)
    clf = bench.make_model(front_end, d, num, dev)
    clf.train()
    T.GRAPH_EPOCHS = graph
    sess = T.Session(clf, synth.node2chrom(num), synth.chrom_range(num).astype(np.int32), min(ks), max(ks), 0, seed=3)
    sess.set_known(edges)
    rows = [r[r != 0] for r in edges]
    gen = T.DataGenerator(rows, weights, T.BATCH_SIZE, steps_per_k, min_size=min(ks), max_size=max(ks))
    # the validation set of the reference's 80 / 20 split (main.py:569-592): eval_epoch takes <= 10 000 shuffled rows of it = 104 batches of 96
    val_n = min(len(edges) // 5, 20000)
    val_e, val_w = edges[-val_n:], weights[-val_n:]
    import tempfile
    ckdir = tempfile.mkdtemp(prefix="matcha_epoch_bench_")
    out = {}
    for ep in range(2):                       # epoch 0 pays the capture / first-touch costs; epoch 1 is the steady state that is reported
        t0 = time.perf_counter()
        T.save_embeddings(clf, N, None)
        t_emb = time.perf_counter() - t0
        t0 = time.perf_counter()
        e_part, w_part = gen.next_iter()
        t_gen = time.perf_counter() - t0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T.train_epoch(sess, e_part, w_part, 1.0, 0.001)
        torch.cuda.synchronize()
        t_train = time.perf_counter() - t0
        n_steps = len(e_part) // T.BATCH_SIZE
        # ... and the rest of the reference's epoch (main.py:300-322): the validation pass and the two checkpoint files
        t0 = time.perf_counter()
        T.eval_epoch(sess, val_e, val_w)
        torch.cuda.synchronize()
        t_eval = time.perf_counter() - t0
        clf.train()
        t0 = time.perf_counter()
        torch.save({"model_link": clf.state_dict(), "epoch": ep}, os.path.join(ckdir, T.MODEL_NAME))
        torch.save(clf, os.path.join(ckdir, "model2load"))
        t_ck = time.perf_counter() - t0
        tm = sess.timing
        # loop = upload + shuffle of the epoch's positives, the step loop, the copy of predictions / sizes back (the epoch's one sync);
        # metrics = the reference's per-epoch sklearn AUROC / AUPR + accuracy per size on the host (utils.py:32-72)
        out = dict(save_embeddings_s=round(t_emb, 4), data_generator_s=round(t_gen, 4), train_epoch_s=round(t_train, 4),
                   loop_s=round(tm.get("loop_s", 0.0), 4), epoch_metrics_s=round(tm.get("metrics_s", 0.0), 4), steps=n_steps,
                   wall_us_per_step=round(tm.get("loop_s", t_train) / n_steps * 1e6, 1),
                   eval_epoch_s=round(t_eval, 4), eval_loop_s=round(tm.get("eval_loop_s", 0.0), 4), eval_batches=min(len(val_e), 10000) // T.BATCH_SIZE,
                   checkpoint_s=round(t_ck, 4),
                   full_epoch_s=round(t_emb + t_gen + t_train + t_eval + t_ck, 4))
    # device time of one step: replays (or eager steps) back to back between two events, no host work in between that the GPU waits for
    st = sess.__dict__.get("_graph_state")
    if graph and st is not None and st["graph"] is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k = min(500, st["chroms"].numel())
        best = float("inf")
        for _ in range(3):                     # the fastest of three passes (a pass right after the host-side metrics can catch the GPU clocked down)
            st["it"].zero_()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(k):
                st["graph"].replay()
            ev1.record()
            torch.cuda.synchronize()
            best = min(best, ev0.elapsed_time(ev1) / k * 1e3)
        out["device_us_per_step"] = round(best, 1)
        out["host_overhead_over_device"] = round(out["wall_us_per_step"] / out["device_us_per_step"] - 1.0, 3)
    out.update(front_end=front_end, epoch_loop="hipGraph replay of one captured step" if graph else "call by call",
               rows_per_step=T.BATCH_SIZE * (1 + T.NEG_NUM), hyperedges_per_s=round(T.BATCH_SIZE * (1 + T.NEG_NUM) / (out["wall_us_per_step"] * 1e-6), 1))
    T.GRAPH_EPOCHS = True
    import shutil
    shutil.rmtree(ckdir, ignore_errors=True)
    return out


def cpu_epoch_estimate(front_end, n_steps, cpu_steps, ks=(2, 3, 4, 5), d=64, layout="hg38_1mb"):
    """The oracle port of the reference step on the host cores (8 threads, the survey's configuration), timed over cpu_steps steps of the
    reference's batch and scaled to the epoch's step count (oracle/ is the checker / baseline only)."""
    import argparse as _a
    num = synth.LAYOUTS[layout]
    N, L = int(np.sum(num)), max(ks)
    rng = np.random.default_rng(2)
    pool = np.concatenate([np.pad(synth.make_edges_fast(rng, N, k, 100000), ((0, 0), (0, L - k))) for k in ks], axis=0)
    wts = rng.uniform(0.6, 1.0, size=len(pool)).astype(np.float32)
    args = _a.Namespace(dim=d, front_end=front_end, cpu_seconds=cpu_steps * 0.05)
    r = bench.cpu_baseline(args, num, list(ks), L, pool, wts, 3)
    return dict(front_end=front_end, cpu_hyperedges_per_s=r["value"], cpu_threads=r["cores"],
                cpu_epoch_s_estimate=round(n_steps * 384 / max(r["value"], 1e-9), 1), sample=r["sample"][:200])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps-per-k", type=int, default=1000)
    ap.add_argument("--front-ends", default="table,adj")
    ap.add_argument("--cpu-steps", type=int, default=0)
    a = ap.parse_args()
    for fe in a.front_ends.split(","):
        for graph in (True, False):
            print(json.dumps(one_case(fe, a.steps_per_k, graph)), flush=True)
        if a.cpu_steps:
            print(json.dumps(cpu_epoch_estimate(fe, 4 * a.steps_per_k, a.cpu_steps)), flush=True)
