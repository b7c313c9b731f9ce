"""Ten steps: fp64 oracle (exact) vs the HIP path -- per step the logit error of HIP and of the golden run against exact, and the
tensors whose parameters drift most (units of lr)."""
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
for k in sys.argv[5:]:
    _lib.set_option(k, 1)
g = gold(f"g3_{name}_phase2.npz")
num = synth.LAYOUTS[layout]
P, fe, _ = oracle_state(num, 64, mode, seed, requires_grad=True)
P = {k: v.detach().double().requires_grad_(v.requires_grad) for k, v in P.items()}
opt = O.AdamWRef()
clf, sd = hip_model(num, 64, mode, seed)
for m in clf.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
clf.train()
tr = Trainer(clf, lr=1e-3)
for step in range(10):
    x, y, w = (torch.from_numpy(g[f"{n}{step}"]) for n in "xyw")
    rc = int(g["chroms"][step])
    loss, bce, recon, logits, grads = O.loss_and_grads(P, fe, x, y.double(), w.double(), 1.0, 0.001, random_chrom=rc)
    lg = tr.forward_backward(x.cuda().contiguous(), y.reshape(-1).cuda().contiguous(), w.reshape(-1).cuda().contiguous(), 1.0, 0.001, rc)
    torch.cuda.synchronize()
    rt = tr.rt
    names = {id(p): n for n, p in clf.named_parameters()}
    gerr = []
    for p_, o in zip(rt.live, rt.seg_off_list[:-1]):
        n = names[id(p_)]
        if grads.get(n) is None:
            continue
        gh = tr.gflat[o:o + p_.numel()].view(p_.shape).cpu().double()
        gerr.append((float((gh - grads[n]).abs().max() / max(float(grads[n].abs().max()), 1e-30)), n))
    e_h = logit_err(lg.detach().cpu().numpy(), logits.numpy())
    e_r = logit_err(g[f"logits{step}"], logits.numpy())
    tr.all_reduce(); tr.optimizer_step()
    opt.step(P, grads)
    torch.cuda.synchronize()
    drift = []
    for n, p in clf.named_parameters():
        if n in P and P[n].requires_grad and grads.get(n) is not None:
            d = (p.detach().cpu().double() - P[n].detach()).abs() / 1e-3
            drift.append((float(d.max()), int((d > 0.05).sum()), n))
    drift.sort(reverse=True)
    gerr.sort(reverse=True)
    print(f"step {step}: logits vs exact: hip {e_h:.1e}  golden {e_r:.1e} | worst grad rel err {gerr[0][0]:.1e} {gerr[0][1].split('.')[-2]}.{gerr[0][1].split('.')[-1]} | param drift (lr): " +
          "  ".join(f"{n.replace('encode1.mul_head_attn.', 'mha.')}: {m:.2f} ({c})" for m, c, n in drift[:4]))
