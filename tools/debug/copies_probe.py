"""Which part of a bench step issues the device-to-device copies rocprof shows (__amd_rocclr_copyBuffer)?  PART=assemble|step"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from matcha_amd import synth
from matcha_amd.engine import Trainer
from matcha_amd.sampler import HyperedgeSet, NegativeSampler

part = os.environ.get("PART", "assemble")
dev = torch.device("cuda:0")
num = synth.LAYOUTS["hg38_1mb"]
N = int(np.sum(num))
rng = np.random.default_rng(0)
pool = np.concatenate([np.pad(synth.make_edges_fast(rng, N, k, 20000), ((0, 0), (0, 5 - k))) for k in (2, 3, 4, 5)])
pool = pool[rng.permutation(len(pool))]
shard = torch.from_numpy(pool).to(dev)
shard_w = torch.rand(len(pool), device=dev)
hset = HyperedgeSet(shard)
sampler = NegativeSampler(hset, synth.node2chrom(num), synth.chrom_range(num), neg_num=3, min_dis=0, seed=1)
P, B, L = 96, 384, 5
x = torch.zeros((B, L), dtype=torch.long, device=dev)
y = torch.cat([torch.ones(P, device=dev), torch.zeros(B - P, device=dev)])
w = torch.ones(B, device=dev)
cursor = torch.zeros(1, dtype=torch.long, device=dev)
ar = torch.arange(P, device=dev)
clf = bench.make_model(os.environ.get("FRONT", "table"), 64, num, dev)
clf.train()
tr = Trainer(clf, lr=1e-3, base_seed=9)
M = shard.shape[0]
def assemble():
    idx = (cursor + ar) % M
    cursor.add_(P)
    torch.index_select(shard, 0, idx, out=x[:P])
    torch.index_select(shard_w, 0, idx, out=w[:P])
    sampler.sample_into(x[:P], x[P:])
assemble(); tr.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=0)
torch.cuda.synchronize()
for _ in range(20):
    if part == "assemble":
        assemble()
    else:
        tr.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=0)
torch.cuda.synchronize()
