"""fused_fwd32 against the four-wave forward (option disable_fwd32): eval logits, training step gradients."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matcha_amd import synth, _lib
from matcha_amd.engine import Trainer
from tests.test_hip_model import hip_model

OPT = os.environ.get("CHECK_OPT", "disable_fwd32")
num = synth.LAYOUTS["c1"]
N = int(np.sum(num))
for ks, per in (([2], 7), ([5], 3), ([2, 3, 4, 5], 40), ([2, 3, 5], 700)):
    x, y, w = synth.make_batch(np.random.default_rng(1), N, ks, per)
    xt = torch.from_numpy(x).cuda()
    outs = {}
    for off in (1, 0):
        with _lib.option(OPT, off):
            clf, _ = hip_model(num, 64, "table", 50)
            clf.eval()
            with torch.no_grad():
                outs[off] = clf(xt).cpu().numpy().ravel()
    d = np.abs(outs[0] - outs[1])
    print("eval ks", ks, "rows", len(x), "max diff", d.max(), "ref scale", np.abs(outs[1]).max(), "first", outs[0][:4], outs[1][:4])
# training step (loss in forward): compare gradients
x, y, w = synth.make_batch(np.random.default_rng(2), N, [2, 3, 4, 5], 300)
xt, yt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(y.reshape(-1)).cuda(), torch.from_numpy(w.reshape(-1)).cuda()
gr = {}
for mode in ("eval", "train"):
    for off in (1, 0):
        with _lib.option(OPT, off):
            clf, _ = hip_model(num, 64, "table", 50)
            clf.train(mode == "train")
            tr = Trainer(clf, base_seed=7)
            lg = tr.forward_backward(xt, yt, wt, 1.0, 0.0, 0)
            torch.cuda.synchronize()
            gr[off] = (tr.gflat.cpu().clone(), lg.cpu().clone(), tr.losses.cpu().clone())
    rt = tr.rt
    print(mode, "logits diff", float((gr[0][1] - gr[1][1]).abs().max()), "loss", gr[0][2].tolist(), gr[1][2].tolist())
    names = list(rt.field_off.keys())
    offs = [rt.field_off[n_] for n_ in names] + [rt.n_flat]
    for n_, a, b in zip(names, offs[:-1], offs[1:]):
        ga, gb = gr[0][0][a:b], gr[1][0][a:b]
        print(f"  {n_:10s} rel diff {float((ga - gb).abs().max() / (gb.abs().max() + 1e-30)):.2e}  scale {float(gb.abs().max()):.2e}")
