"""Per-step logit error of the golden training runs: the oracle's self-noise (row-permuted batches) next to the HIP path's variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matcha_amd import synth, _lib
from matcha_amd.engine import Trainer
from tests.helpers import gold, logit_err, oracle_self_noise
from tests.test_hip_model import hip_model

def run(name, layout, seed, mode, opts):
    g = gold(f"g3_{name}_phase2.npz")
    for k, v in opts.items():
        _lib.set_option(k, v)
    clf, sd = hip_model(synth.LAYOUTS[layout], 64, mode, seed)
    for m in clf.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clf.train()
    tr = Trainer(clf, lr=1e-3)
    errs = []
    for step in range(10):
        x, y, w = (torch.from_numpy(g[f"{n}{step}"]).cuda() for n in "xyw")
        lg = tr.step(x.contiguous(), y.reshape(-1).contiguous(), w.reshape(-1).contiguous(), 1.0, 0.001, int(g["chroms"][step]))[2]
        errs.append(logit_err(lg.detach().cpu().numpy(), g[f"logits{step}"]))
    for k in opts:
        _lib.set_option(k, 0)
    return errs

for name, layout, seed, mode in (("c23_table_d64", "c23", 44, "table"), ("c1_table_d64", "c1", 43, "table"), ("c23_adj_d64", "c23", 45, "adj")):
    for ps in (0, 1, 2):
        e, pn = oracle_self_noise(name, layout, 64, mode, seed, 1.0, 0.001, "phase2", 10, perm_seed=ps)
        print(f"{name:16s} oracle perm{ps}  ", " ".join(f"{v:.1e}" for v in e), " max param noise %.1e" % max(pn.values()))
    for tag, opts in (("merged", {}), ("4-product", {"disable_merged": 1}), ("layerwise", {"disable_fused": 1})):
        e = run(name, layout, seed, mode, opts)
        print(f"{name:16s} {tag:13s}", " ".join(f"{v:.1e}" for v in e))
