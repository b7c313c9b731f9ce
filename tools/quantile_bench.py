"""Throughput of the device quantile transform (csrc/quantile.hip) with the column resident in HBM, beside
scikit-learn's QuantileTransformer (subsample=None: the same arithmetic, every row fitted) on the host cores.
Algorithmic bytes per row: 4 (read for the sort) + 4 (read for the transform) + 4 (write) = 12; the radix sort's own passes
are on top of that."""
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, ".")
from matcha_amd import positives as P

rng = np.random.default_rng(5)
sizes = [int(v) for v in sys.argv[1:]] or [100_000, 1_000_000, 10_000_000, 100_000_000]
for n in sizes:
    col = (np.floor(rng.pareto(1.1, size=n)) + 2).astype("float32")
    x = torch.from_numpy(col).cuda()
    P.quantile_uniform(x)
    torch.cuda.synchronize()
    reps = 20 if n <= 10_000_000 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        w = P.quantile_uniform(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    line = f"n={n:>11d}: {dt * 1e3:9.3f} ms  {n / dt / 1e9:7.3f} G rows/s  {12 * n / dt / 1e9:8.1f} GB/s algorithmic"
    if n <= 10_000_000:
        from sklearn.preprocessing import QuantileTransformer
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t1 = time.perf_counter()
            ref = QuantileTransformer(n_quantiles=1000, output_distribution="uniform", subsample=None).fit_transform(col.reshape(-1, 1)).reshape(-1)
            cpu = time.perf_counter() - t1
        line += f";  scikit-learn on the host: {cpu * 1e3:9.1f} ms ({n / cpu / 1e6:.1f} M rows/s), identical = {bool(np.array_equal(ref, w.cpu().numpy()))}"
    print(line, flush=True)
