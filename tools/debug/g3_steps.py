"""Per-step logit error of the 10-step golden training run (c23 table d64) for the HIP path variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matcha_amd import synth, _lib
from matcha_amd.engine import Trainer
from tests.helpers import gold, logit_err
from tests.test_hip_model import hip_model

def run(name, layout, seed, opts, deterministic=False):
    g = gold(f"g3_{name}_phase2.npz")
    num = synth.LAYOUTS[layout]
    for k, v in opts.items():
        _lib.set_option(k, v)
    clf, sd = hip_model(num, 64, "table", seed)
    for m in clf.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clf.train()
    tr = Trainer(clf, lr=1e-3, deterministic=deterministic)
    errs = []
    for step in range(10):
        x, y, w = (torch.from_numpy(g[f"{n}{step}"]).cuda() for n in "xyw")
        lg = tr.step(x.contiguous(), y.reshape(-1).contiguous(), w.reshape(-1).contiguous(), 1.0, 0.001, int(g["chroms"][step]))[2]
        errs.append(logit_err(lg.detach().cpu().numpy(), g[f"logits{step}"]))
    for k in opts:
        _lib.set_option(k, 0)
    return errs

for name, layout, seed in (("c23_table_d64", "c23", 44), ("c1_table_d64", "c1", 43), ("hg38_table_d64", "hg38_1mb", 0)):
    try:
        for tag, opts, det in (("fwd32", {}, False), ("fwd32 det", {}, True), ("old fwd", {"disable_fwd32": 1}, False), ("layerwise", {"disable_fused": 1}, False)):
            e = run(name, layout, seed, opts, det)
            print(f"{name:16s} {tag:10s}", " ".join(f"{v:.1e}" for v in e))
    except Exception as ex:
        print(name, "skipped:", ex)

# step-0 gradient error per tensor against the golden gradients (max abs err / max abs ref), c23 vs c1
from tests.test_hip_model import _trainer_grads
for name, layout, seed in (("c23_table_d64", "c23", 44), ("c1_table_d64", "c1", 43)):
    g = gold(f"g3_{name}_phase2.npz")
    clf, sd = hip_model(synth.LAYOUTS[layout], 64, "table", seed)
    for m in clf.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clf.train()
    tr = Trainer(clf, lr=1e-3)
    x, y, w = (torch.from_numpy(g[f"{n}0"]).cuda() for n in "xyw")
    tr.forward_backward(x.contiguous(), y.reshape(-1).contiguous(), w.reshape(-1).contiguous(), 1.0, 0.001, 0)
    torch.cuda.synchronize()
    out = []
    for n, v in _trainer_grads(tr, clf).items():
        if v is None or ("grad0/" + n) not in g.files:
            continue
        ref = g["grad0/" + n]
        d = np.abs(v.cpu().numpy() - ref)
        small = np.abs(ref) < 1e-7
        out.append((float(d.max() / max(np.abs(ref).max(), 1e-30)), n, float(np.abs(ref).max()), float(d.max()), int(small.sum()), ref.size))
    print(name)
    for e, n, sc, dm, ns, sz in sorted(out, reverse=True)[:12]:
        print(f"   {n:45s} rel {e:.1e}  scale {sc:.1e}  max abs err {dm:.1e}   |ref| < 1e-7: {ns}/{sz}")
