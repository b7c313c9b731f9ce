"""Throughput of the device k-mer generation against the CPU restatement of generate_kmers.py on a
subsample.  Synthetic clusters: sorted unique node lists of 2..25 bins around random centres of an hg38-1Mb-sized node range."""
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from matcha_amd import kmers as KM
from oracle import kmers as OK

rng = np.random.default_rng(3)
N, n_cl, max_size = 3067, int(sys.argv[1]) if len(sys.argv) > 1 else 200000, 25
cl = []
for _ in range(n_cl):
    n = int(rng.integers(2, max_size + 1))
    centre = int(rng.integers(1, N))
    c = np.unique(np.clip(centre + rng.integers(-40, 41, size=2 * n), 1, N))[:n]
    if len(c) >= 2:
        cl.append(c.astype(np.int64))
print(f"{len(cl)} clusters, {sum(len(c) for c in cl)} nodes in total")
for k in (2, 3, 4, 5):
    total = sum(math.comb(len(c), k) for c in cl if k <= len(c) <= max_size)
    KM.generate_kmers(cl[:1000], k, 0, max_size, 2, n_nodes=N)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows, freq = KM.generate_kmers(cl, k, 0, max_size, 2, n_nodes=N)
    dt = time.perf_counter() - t0
    sub = cl[: max(200, len(cl) // 200)]
    sub_total = sum(math.comb(len(c), k) for c in sub if k <= len(c) <= max_size)
    t1 = time.perf_counter()
    OK.generate_kmers(sub, k, 0, max_size, 2)
    cpu = time.perf_counter() - t1
    print(f"k={k}: {total:>12d} candidate subsets -> {len(rows):>9d} k-mers (freq >= 2) in {dt * 1e3:8.1f} ms end to end "
          f"({total / dt / 1e6:7.1f} M subsets/s incl. host CSR + copies); CPU restatement on {len(sub)} clusters: {sub_total / cpu / 1e6:.3f} M subsets/s")
