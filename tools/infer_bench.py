#!/usr/bin/env python3
"""Forward-only (inference) throughput of model(x) on one MI355X: the predict_multiway / denoise_contact consumer path."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matcha_amd import synth
import bench
class A: pass
a = A(); a.dim = 64; a.front_end = sys.argv[1] if len(sys.argv) > 1 else "table"
num = synth.LAYOUTS["hg38_1mb"]; N = int(np.sum(num))
clf = bench.make_model(a, num, torch.device("cuda", 0)).eval()
rng = np.random.default_rng(0)
B = 65536
xs = [np.pad(synth.make_edges_fast(rng, N, k, B // 4), ((0, 0), (0, 5 - k))) for k in (2, 3, 4, 5)]
x = torch.from_numpy(np.concatenate(xs)[rng.permutation(B)]).cuda()
with torch.no_grad():
    for _ in range(5): clf(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): clf(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"front_end={a.front_end} fused={'MATCHA_DISABLE_FUSED' not in os.environ} forward: {dt*1e3:.3f} ms per 65536 rows -> {B/dt/1e6:.2f} M rows/s")
