import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from matcha_amd import synth
from matcha_amd.engine import Trainer
from tests.test_hip_model import hip_model
num = synth.LAYOUTS["c23"]; N = int(np.sum(num))
x, y, w = synth.make_batch(np.random.default_rng(5), N, [2, 3], 48)
xt, yt, wt = (torch.from_numpy(a).cuda().contiguous() for a in (x, y.reshape(-1), w.reshape(-1)))
for r in (0, 3, 22):
    gs = []
    for dev in (False, True, True):
        clf, _ = hip_model(num, 64, "adj", 81); clf.train()
        tr = Trainer(clf, lr=1e-3, base_seed=5)
        rc = torch.tensor([r], dtype=torch.int32, device="cuda") if dev else r
        tr.forward_backward(xt, yt, wt, 1.0, 0.001, rc); torch.cuda.synchronize()
        gs.append((tr.gflat.clone(), tr.touched.clone(), tr.losses.clone()))
    rt = tr.rt
    names = {id(p): n for n, p in clf.named_parameters()}
    print("r", r, "losses", gs[0][2].tolist(), gs[1][2].tolist(), "touched equal", torch.equal(gs[0][1], gs[1][1]))
    for p_, o in zip(rt.live, rt.seg_off_list[:-1]):
        a, b, c = (g[0][o:o + p_.numel()] for g in gs)
        d1, d2 = float((a - b).abs().max()), float((b - c).abs().max())
        sc = float(a.abs().max())
        if d1 > 1e-6 * max(sc, 1e-12) or d2 > 1e-6 * max(sc, 1e-12):
            print("   ", names[id(p_)], "scale %.2e  host-vs-dev %.2e  dev-vs-dev %.2e" % (sc, d1, d2))
