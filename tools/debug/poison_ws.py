"""Find reads of never-written workspace bytes: run the same training steps with the workspace pre-filled with zeros, with NaN bit
patterns and with large finite garbage before every step; any difference in the gradients names the tensors that depend on it."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from matcha_amd import synth
from matcha_amd.engine import Trainer
from tests.test_hip_model import hip_model
from tests.test_hip_properties import _big_batch

def run(fill, deterministic, d=64, mode="table", rows=16384, steps=2):
    num = synth.LAYOUTS["hg38_1mb"]; N = int(np.sum(num))
    rng = np.random.default_rng(1)
    x = torch.from_numpy(_big_batch(N, rows, rng)).cuda()
    y = (torch.rand(len(x), device="cuda", generator=torch.Generator("cuda").manual_seed(3)) < 0.25).float()
    w = torch.ones(len(x), device="cuda")
    clf, _ = hip_model(num, d, mode, 3); clf.train()
    tr = Trainer(clf, base_seed=5, deterministic=deterministic)
    gs = []
    for s in range(steps):
        ws, _ = tr._buffers(len(x), x.shape[1]) if hasattr(tr, "_buffers") else (None, None)
        if ws is not None:
            if fill == "zero": ws.zero_()
            elif fill == "nan": ws.fill_(0xFF)
            elif fill == "big": ws.view(torch.float32)[: ws.numel() // 4].fill_(1e30)
        tr.forward_backward(x, y, w, 1.0, 0.0, 0)
        torch.cuda.synchronize()
        gs.append(tr.gflat.clone())
        tr.optimizer_step()
    return tr, gs

for det in (True,):
    base = None
    for fill in ("zero", "zero", "nan", "nan", "none", "none", "big"):
        tr, gs = run(fill, det)
        if base is None: base = gs; print("det", det, "zero-fill |g|max", float(gs[0].abs().max())); continue
        for s, (a, b) in enumerate(zip(base, gs)):
            bad = ~(a == b)
            print("det", det, fill, "step", s, "differing elements", int(bad.sum()), "nan", int(torch.isnan(b).sum()))
            if bad.any():
                rt = tr.rt
                idx = torch.nonzero(bad).flatten()
                offs = list(rt.seg_off_list)
                names = [n for n, _ in tr.clf.named_parameters()] if hasattr(tr, "clf") else None
                import bisect
                segs = sorted(set(bisect.bisect_right(offs, int(i)) - 1 for i in idx[:: max(1, len(idx) // 2000)].tolist()))
                print("   segments", segs, [offs[k] for k in segs])
