"""k-mer generation on the device: the flow of the reference's ``generate_kmers.py``.

``generate_kmers(clusters, size, ...)`` returns the k-mers of one size and their frequencies; ``main()`` reads
``config.JSON`` / ``temp_dir/edge_list.npy`` / ``chrom_range.npy`` and writes ``all_<k>_counter.npy`` (int [M, k]) and
``all_<k>_freq_counter.npy`` ([M]) for every configured size, the files ``matcha_amd/train.py`` (main.py:538-566) reads.
Rows come out in lexicographic order (the reference's order depends on process scheduling, generate_kmers.py:108-129).
There is no CPU path: the counting runs in libmatcha_hip.so (csrc/kmers.hip)."""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib

MAX_CLUSTER_LEN = 64          # csrc/kmers.hip kMaxClusterLen (the reference's max_cluster_size is 25, config.JSON)
MAX_COMBOS = (1 << 32) - 2    # candidate k-subsets per launch


def _csr(clusters: Sequence[Sequence[int]]):
    lens = np.fromiter((len(c) for c in clusters), dtype=np.int64, count=len(clusters))
    offsets = np.zeros(len(clusters) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    ids = np.concatenate([np.asarray(c, dtype=np.int32) for c in clusters]) if len(clusters) else np.zeros(0, dtype=np.int32)
    return ids, offsets, lens


def generate_kmers(clusters: Sequence[Sequence[int]], size: int, min_dis: int, max_size: int, min_freq_cutoff: int, n_nodes: int = 0,
                   device="cuda", max_combos_per_launch: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """(kmers int64 [M, size], freq int64 [M]) for one k-mer size (generate_kmers.py:8-69, :86-96).  ``clusters``: sorted
    unique node-id lists (process.py:66-77).  Launches are split so that each counts fewer than 2^32 candidate subsets AND
    fits half of the free device memory (``max_combos_per_launch`` lowers the bound further: tests); partial results of
    different launches are merged on the device (a k-mer may occur in several launches)."""
    if max_size > MAX_CLUSTER_LEN:
        raise _lib.MatchaHipError(f"max_cluster_size {max_size} exceeds the device kernel's limit {MAX_CLUSTER_LEN}")
    lib = _lib.load()
    ids, offsets, lens = _csr(clusters)
    if n_nodes <= 0:
        n_nodes = int(ids.max()) if len(ids) else 1
    per = np.array([math.comb(int(n), size) if size <= n <= max_size else 0 for n in lens], dtype=np.int64)     # :88
    if per.sum() == 0:
        return np.zeros((0, size), dtype=np.int64), np.zeros((0,), dtype=np.int64)
    dev = torch.device(device)
    ids_d = torch.from_numpy(ids).to(dev)
    off_d = torch.from_numpy(offsets).to(dev)
    parts = []
    start = 0
    # A launch holds every candidate subset at once (keys, their sorted copy, run lengths, the unpacked rows): bound the
    # candidates per launch by the FREE device memory as well as by the 2^32 - 2 index limit, so that a realistic data set
    # (millions of clusters of up to 25 bins: 1e10 candidates at k = 5) splits further instead of running out of memory
    probe = 1 << 20
    per_cand = lib.matcha_kmer_workspace_bytes(probe, size, n_nodes) / probe + 8.0 * size + 8.0
    free_bytes = torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" else 1 << 34
    launch_cap = int(min(MAX_COMBOS, max(1 << 16, 0.5 * free_bytes / max(per_cand, 1.0))))
    if max_combos_per_launch:
        launch_cap = min(launch_cap, int(max_combos_per_launch))
    while start < len(per):                                             # greedy split into launches of <= launch_cap candidates
        run, end = 0, start
        while end < len(per) and run + per[end] <= launch_cap:
            run += int(per[end])
            end += 1
        if end == start:
            raise _lib.MatchaHipError(f"one cluster alone has {int(per[start])} candidate k-subsets: more than a launch can hold ({launch_cap})")
        if run > 0:
            comb = np.zeros(end - start + 1, dtype=np.int64)
            np.cumsum(per[start:end], out=comb[1:])
            comb_d = torch.from_numpy(comb).to(dev)
            # this launch only needs min_freq 1 when the clusters are split over several launches (merged below)
            single = start == 0 and end == len(per)
            ws_bytes = lib.matcha_kmer_workspace_bytes(run, size, n_nodes)
            if ws_bytes == 0:
                raise _lib.MatchaHipError(lib.matcha_last_error().decode())
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            cap = run
            out_k = torch.empty((cap, size), dtype=torch.int64, device=dev)
            out_f = torch.empty(cap, dtype=torch.int64, device=dev)
            n_out = torch.zeros(1, dtype=torch.int64, device=dev)
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.matcha_kmer_generate(_lib.ptr(ids_d), C.c_void_p(off_d.data_ptr() + 8 * start), _lib.ptr(comb_d), end - start, run, size,
                                                n_nodes, min_dis, min_freq_cutoff if single else 1, _lib.ptr(out_k), _lib.ptr(out_f), cap,
                                                _lib.ptr(n_out), _lib.ptr(ws), ws_bytes, st), "matcha_kmer_generate")
            m = int(n_out.item())
            parts.append((out_k[:m].clone(), out_f[:m].clone(), single))     # compact copies; the launch's big buffers are freed
            del ws, out_k, out_f
        start = end
    if len(parts) == 1 and parts[0][2]:
        return parts[0][0].cpu().numpy(), parts[0][1].cpu().numpy()
    # several launches: merge equal rows ON THE DEVICE (each part is sorted and duplicate-free; the same k-mer may occur in
    # several parts), then threshold: lexicographic order by stable sorts from the last column to the first
    rows = torch.cat([p[0] for p in parts])
    freq = torch.cat([p[1] for p in parts])
    del parts
    order = torch.arange(len(rows), device=rows.device)
    for c in range(size - 1, -1, -1):
        order = order[torch.sort(rows[order, c], stable=True).indices]
    rows, freq = rows[order], freq[order]
    uniq, inv = torch.unique_consecutive(rows, dim=0, return_inverse=True)
    tot = torch.zeros(len(uniq), dtype=freq.dtype, device=freq.device).index_add_(0, inv, freq)
    keep = tot >= min_freq_cutoff
    return uniq[keep].cpu().numpy(), tot[keep].cpu().numpy()


def main(argv=None):
    """generate_kmers.py:72-151: every size of config['k-mer_size'] -> temp_dir/all_<k>_counter.npy, all_<k>_freq_counter.npy."""
    import argparse
    ap = argparse.ArgumentParser(description="k-mer generation (generate_kmers.py) on the MI355X path")
    ap.add_argument("--config", default="./config.JSON")
    args = ap.parse_args(argv)
    with open(args.config) as f:
        config = json.load(f)
    temp_dir = config["temp_dir"]
    chrom_range = np.load(os.path.join(temp_dir, "chrom_range.npy"))
    n_nodes = int(np.max(chrom_range))                                  # node ids are 1 .. max(chrom_range) - 1 (generate_kmers.py:80)
    data = np.load(os.path.join(temp_dir, "edge_list.npy"), allow_pickle=True)
    for size in config["k-mer_size"]:
        rows, freq = generate_kmers(data, size, config["min_distance"], config["max_cluster_size"], config["min_freq_cutoff"], n_nodes)
        if len(rows) > 0:                                               # the reference only writes non-empty results (:133-139)
            np.save(os.path.join(temp_dir, "all_%d_counter.npy" % size), rows)
            np.save(os.path.join(temp_dir, "all_%d_freq_counter.npy" % size), freq)
        print("k = %d: %d k-mers" % (size, len(freq)), {">= %d" % c: int(np.sum(freq >= c)) for c in (2, 3, 4, 5, 6, 7, 8)})


if __name__ == "__main__":
    main()
