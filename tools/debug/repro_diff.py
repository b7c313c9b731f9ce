"""Which parameters differ between identical training steps of fresh Trainers? (the check of test_full_size_train_step_is_reproducible,
with names and counts)   usage: python tools/debug/repro_diff.py [deterministic 0|1] [runs]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from matcha_amd import synth
from matcha_amd.engine import Trainer
from tests.test_hip_model import hip_model
from tests.test_hip_properties import _big_batch

det = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
num = synth.LAYOUTS["hg38_1mb"]; N = int(np.sum(num))
rng = np.random.default_rng(1)
x = torch.from_numpy(_big_batch(N, 16384, rng)).cuda()
y = (torch.rand(len(x), device="cuda") < 0.25).float()
w = torch.ones(len(x), device="cuda")
outs = []
for _ in range(runs):
    clf, _ = hip_model(num, 64, "table", 3); clf.train()
    tr = Trainer(clf, base_seed=5, deterministic=det)
    tr.forward_backward(x, y.reshape(-1), w.reshape(-1), 1.0, 0.001, 0)
    torch.cuda.synchronize()
    g = tr.gflat.clone()
    outs.append({n: g[o:o + p.numel()].clone() for (n, p), o in zip(clf.named_parameters(), [0] * 0)} or {"gflat": g})
    outs[-1]["names"] = [(n, p.numel()) for n, p in clf.named_parameters()]
    rt = tr.rt
    outs[-1]["off"] = dict(rt.field_off)
base = outs[0]["gflat"]
offs = sorted(outs[0]["off"].items(), key=lambda kv: kv[1])
for i, o in enumerate(outs[1:], 1):
    d = (o["gflat"] != base)
    print("run", i, "differing gradient elements:", int(d.sum()))
    for (name, lo), (_, hi) in zip(offs, offs[1:] + [("end", base.numel())]):
        c = int(d[lo:hi].sum())
        if c:
            print("   ", name, c, "of", hi - lo, "max abs diff", float((o["gflat"][lo:hi] - base[lo:hi]).abs().max()), "scale", float(base[lo:hi].abs().max()))
