"""Which tensors carry the HIP path's later-step deviation?  An fp64 run of the oracle is the exact trajectory; after step 0 every
parameter of (a) the reference's fp32 run (the golden file) and (b) the HIP path is compared with it, tensor by tensor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matcha_amd import synth, _lib
from matcha_amd.engine import Trainer
from oracle import hypersagnn as O
from tests.helpers import gold, logit_err, oracle_state
from tests.test_hip_model import hip_model

name, layout, seed, mode = sys.argv[1:5] if len(sys.argv) > 4 else ("c23_table_d64", "c23", 44, "table")
seed = int(seed)
g = gold(f"g3_{name}_phase2.npz")
num = synth.LAYOUTS[layout]
P, fe, _ = oracle_state(num, 64, mode, seed, requires_grad=True)
P = {k: v.detach().double().requires_grad_(v.requires_grad) for k, v in P.items()}
opt = O.AdamWRef()
x, y, w = (torch.from_numpy(g[f"{n}0"]) for n in "xyw")
loss, bce, recon, logits, grads = O.loss_and_grads(P, fe, x, y.double(), w.double(), 1.0, 0.001, random_chrom=int(g["chroms"][0]))
G64 = {k: (None if v is None else v.clone()) for k, v in grads.items()}
opt.step(P, grads)

clf, sd = hip_model(num, 64, mode, seed)
for m in clf.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
clf.train()
tr = Trainer(clf, lr=1e-3)
tr.forward_backward(x.cuda().contiguous(), y.reshape(-1).cuda().contiguous(), w.reshape(-1).cuda().contiguous(), 1.0, 0.001, int(g["chroms"][0]))
torch.cuda.synchronize()
rt = tr.rt
names = {id(p): n for n, p in clf.named_parameters()}
ghip = {names[id(p)]: tr.gflat[o:o + p.numel()].view(p.shape).cpu().double() for p, o in zip(rt.live, rt.seg_off_list[:-1])}
tr.all_reduce(); tr.optimizer_step()
torch.cuda.synchronize()
params = {n: p.detach().cpu().double() for n, p in clf.named_parameters()}
print(f"{'tensor':48s} {'|g| max':>9s} {'g err hip':>9s} {'g err ref':>9s} | param err after step 0, units of lr: {'hip':>7s} {'ref':>7s}  #elements off by > 0.01 lr (hip / ref)")
for n in sorted(ghip):
    if G64.get(n) is None:
        continue
    g64 = G64[n]
    ge_h = float((ghip[n] - g64).abs().max())
    ge_r = float((torch.from_numpy(g["grad0/" + n]).double() - g64).abs().max()) if ("grad0/" + n) in g.files else float("nan")
    pe_h = (params[n] - P[n].detach()).abs() / 1e-3
    pe_r = (torch.from_numpy(g["param0/" + n]).double() - P[n].detach()).abs() / 1e-3 if ("param0/" + n) in g.files else None
    print(f"{n:48s} {float(g64.abs().max()):9.2e} {ge_h:9.2e} {ge_r:9.2e} | {float(pe_h.max()):7.3f} {float(pe_r.max()) if pe_r is not None else float('nan'):7.3f}   "
          f"{int((pe_h > 0.01).sum())} / {int((pe_r > 0.01).sum()) if pe_r is not None else -1} of {g64.numel()}")
