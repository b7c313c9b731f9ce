"""Device feature construction (csrc/features.hip) at the sizes of BASELINE.json's 100 kb configuration
(N = 30 344 bins, largest chromosome 2 491 bins), inputs resident in HBM, beside numpy / the reference's row loop on the host
for a bounded sample."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from matcha_amd import features as F, process as PR
from oracle import hypersagnn as O


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


rng = np.random.default_rng(1)
for n in (250, 2491):
    i = np.arange(n)
    a = (rng.gamma(2.0, 1.0, size=(n, n)) / (np.abs(i[:, None] - i[None, :]) + 1.0)).astype(np.float32)
    ad = torch.from_numpy(a).cuda()
    dt = timed(lambda: F.corrcoef_features(ad, [[1, n + 1]]))
    t0 = time.perf_counter()
    ref = O.corrcoef_features(a, np.array([[1, n + 1]]))[0]
    cpu = time.perf_counter() - t0
    err = float(np.abs(F.corrcoef_features(ad, [[1, n + 1]])[0].cpu().numpy() - ref).max())
    print(f"corrcoef n={n:5d}: {dt * 1e3:8.3f} ms  {2.0 * n ** 3 / dt / 1e12:6.2f} TFLOP/s f64 (upper-triangle tiles only: {n ** 3 / dt / 1e12:.2f} executed);"
          f"  numpy on the host {cpu * 1e3:8.1f} ms;  max |diff| {err:.1e}", flush=True)

N = 30344
m = torch.rand((N, N), device="cuda") * (torch.rand((N, N), device="cuda") < 0.5)
work = m.clone()
dt = timed(lambda: F.zscore_rows_(work.copy_(m)), reps=3) - timed(lambda: work.copy_(m), reps=3)
sample = m[:300].cpu().numpy()
t0 = time.perf_counter()
O.zscore_inter(sample)
cpu = (time.perf_counter() - t0) * N / 300
print(f"zscore_rows {N} x {N}: {dt * 1e3:8.2f} ms  {8.0 * N * N / dt / 1e9:7.1f} GB/s algorithmic (4 B read + 4 B written per entry);"
      f"  numpy row loop on the host, extrapolated from 300 rows: {cpu:.1f} s", flush=True)
del m, work

P = 50_000_000
n2c = np.zeros(N + 1, dtype=np.int32)
n2c[1:] = np.minimum(np.arange(N) // 1320, 22)
i2n = torch.arange(1, N + 1, dtype=torch.int32, device="cuda")
b1 = torch.randint(0, N, (P,), device="cuda")
off = (torch.rand(P, device="cuda") ** 4 * (N - 1)).long()            # most pixels near the diagonal, like a contact map
b2 = torch.clamp(b1 + off, max=N - 1)
cnt = torch.rand(P, device="cuda", dtype=torch.float64)
intra = torch.zeros((N, N), dtype=torch.float64, device="cuda")
inter = torch.zeros((N, N), dtype=torch.float64, device="cuda")
n2c_d = n2c
dt = timed(lambda: PR.pixels_to_adj(b1, b2, cnt, i2n, n2c_d, N, out=(intra, inter)), reps=3)
print(f"pixels_to_adj {P} pixels into 2 x {N}^2 float64: {dt * 1e3:8.2f} ms  {P / dt / 1e9:6.2f} G pixels/s  {40.0 * P / dt / 1e9:7.1f} GB/s algorithmic "
      f"(24 B read + 2 x 8 B added per pixel)", flush=True)
