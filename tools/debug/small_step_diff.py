"""One Trainer step at a small batch with the small-batch kernels on and off: per-parameter gradient differences (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matcha_amd import synth, _lib
from tests.test_hip_model import hip_model
from matcha_amd.engine import Trainer

mode = sys.argv[1] if len(sys.argv) > 1 else "table"
num = synth.LAYOUTS["c23"]
N = int(np.sum(num))
x, y, w = synth.make_batch(np.random.default_rng(5), N, [2, 3, 5], 60)
xt, yt, wt = (torch.from_numpy(a).cuda().contiguous() for a in (x, y.reshape(-1), w.reshape(-1)))
res = {}
for small in (1, 0, 1):
    _lib.set_option("disable_small_batch", 0 if small else 1)
    clf, _ = hip_model(num, 64, mode, 41)
    clf.train(True)
    tr = Trainer(clf, base_seed=8, deterministic=bool(int(os.environ.get('DET', '0'))))
    lg = tr.forward_backward(xt, yt, wt, 1.0, 0.001, 1).clone()
    torch.cuda.synchronize()
    res.setdefault(small, []).append((lg, tr.gflat.clone(), tr.losses.clone() if hasattr(tr, "losses") else None, tr.rt, clf))
_lib.set_option("disable_small_batch", 0)
a, b = res[1][0], res[0][0]
print("logits diff", float((a[0] - b[0]).abs().max()), "losses", a[2], b[2])
rt, clf = a[3], a[4]
names = {id(p): n for n, p in clf.named_parameters()}
for p_, o in zip(rt.live, rt.seg_off_list[:-1]):
    ga, gb = a[1][o:o + p_.numel()], b[1][o:o + p_.numel()]
    d = float((ga - gb).abs().max())
    if d > 1e-6 * max(float(gb.abs().max()), 1e-6):
        print(f"{names[id(p_)]:50s} small {float(ga.abs().max()):.3e} large {float(gb.abs().max()):.3e} diff {d:.3e}")
c = res[1][1]
print("second small run equal:", torch.equal(a[1], c[1]), "max rel diff", float((a[1] - c[1]).abs().max() / a[1].abs().max()))
