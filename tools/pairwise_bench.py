"""Throughput of the pairwise sweep (denoise_contact.py:147-153) on the fused forward: all intra-chromosome
pairs of one chromosome at k = 2, generated and scored on the device.  Usage: python tools/pairwise_bench.py [layout] [d]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from matcha_amd import predict as PR
from matcha_amd import synth
from tests.test_hip_model import hip_model

layout = sys.argv[1] if len(sys.argv) > 1 else "hg38_100kb"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
num = synth.LAYOUTS[layout]
clf, _ = hip_model(num, d, "table", 1)
cr = np.asarray(synth.chrom_range(num))
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pairs, p = PR.pairwise_probabilities(clf, cr, 0, 2)
    m = PR.proba2matrix(pairs, None, p)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{layout} chromosome 0 ({num[0]} bins), d={d}: {len(pairs)} pairs scored + scattered in {dt * 1e3:.2f} ms -> {len(pairs) / dt / 1e6:.1f} M pairs/s")
