"""Shared test helpers: synthetic model state for the oracle and for the HIP path."""
import os

import numpy as np
import torch

from matcha_amd import synth
from oracle import hypersagnn as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


# Round 6 fixtures g3big_* (tests/golden/make_golden.py::g3_big): name -> (layout, embed_dim, front end, weight seed).  One dropout-free
# training step of the REAL reference on batches large enough for the kernels the library picks at bench sizes.
G3BIG = {
    "hg38_table_d64_k5": ("hg38_1mb", 64, "table", 61), "hg38_adj_d64_k5": ("hg38_1mb", 64, "adj", 62),
    "c23_table_d64_k8": ("c23", 64, "table", 63), "c1_table_d128_k5": ("c1", 128, "table", 64), "c1_adj_d128_k5": ("c1", 128, "adj", 65),
    "c1_table_d128_k8": ("c1", 128, "table", 66), "c1_table_d256_k8": ("c1", 256, "table", 67), "c1_table_d64_k8_small": ("c1", 64, "table", 68),
}


def g3big_batch(g):
    """(x int64 [B, L], y, w float32 [B, 1]) of a g3big fixture (node ids are stored in a narrow integer type)."""
    return torch.from_numpy(g["x0"].astype(np.int64)), torch.from_numpy(g["y0"]), torch.from_numpy(g["w0"])


def g3big_grad_ref(g, name):
    """(stride, reference gradient) of one tensor: stored in full (stride 1) or every stride-th element of the flattened tensor."""
    if ("grad0/" + name) in g.files:
        return 1, g["grad0/" + name].reshape(-1)
    for key in g.files:
        if key.startswith("grad0s") and key.split("/", 1)[1] == name:
            return int(key.split("/", 1)[0][len("grad0s"):]), g[key]
    raise KeyError(name)


def oracle_state(num, d, mode, seed, requires_grad=False):
    """(P, fe, sd_numpy): the same deterministic weights/features make_golden.py loaded into the reference."""
    attr = O.attribute_table(num)
    sd = synth.make_state_dict(np.random.default_rng(seed), num, d, mode, attr)
    P = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.array(v))
        if requires_grad and not k.startswith("attribute_dict"):
            t.requires_grad_(True)
        P[k] = t
    fe = front_end(num, mode, seed)
    return P, fe, sd


def front_end(num, mode, seed):
    if mode == "table":
        return O.FrontEnd(mode="table", bounds=synth.bounds(num))
    intra, inter = synth.make_adjacency(np.random.default_rng(seed + 1000), num)
    feats = [torch.from_numpy(f) for f in O.corrcoef_features(intra, synth.chrom_range(num))]
    return O.FrontEnd(mode="adj", bounds=synth.bounds(num), feats=feats, inter=torch.from_numpy(O.zscore_inter(inter)))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def logit_err(a, b, floor=0.05):
    """ELEMENT-wise relative error of a batch of logits: each |a_i - b_i| against max(|b_i|, floor * max|b|), so a small logit next
    to one large logit is still held to the tolerance (rel_err alone is norm-wise: max|delta| / max|ref|).  The floor is there
    because a logit is a sum of terms of the batch's typical magnitude: below it the absolute error is the meaningful one."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    den = np.maximum(np.abs(b), floor * (np.abs(b).max() + 1e-30))
    return float((np.abs(a - b) / den).max())


def oracle_self_noise(name, layout, d, mode, seed, alpha, beta, tag, n_steps, perm_seed=0):
    """The oracle's own fp32 noise floor on a G3 fixture: the recorded batches are replayed through the oracle with the ROWS of every
    batch permuted -- the same mathematics, another summation order in every reduction over rows -- and compared with the golden
    (reference) values exactly like the HIP path is.  Returns (per-step element-wise logit error, {tensor: max |param - golden| after
    the last step}).  AdamW turns the rounding noise of gradient elements near its eps into visible fractions of lr, so from step 1 on
    this floor -- not the forward tolerance -- is what a correct implementation can be held to (tests/test_hip_model.py)."""
    g = gold(f"g3_{name}_{tag}.npz")
    num = synth.LAYOUTS[layout]
    P, fe, _ = oracle_state(num, d, mode, seed, requires_grad=True)
    opt = O.AdamWRef()
    rng = np.random.default_rng(perm_seed)
    errs = []
    for step in range(n_steps):
        x, y, w = (g[f"{n}{step}"] for n in "xyw")
        perm = rng.permutation(len(x))
        xt, yt, wt = (torch.from_numpy(np.ascontiguousarray(a[perm])) for a in (x, y, w))
        loss, bce, recon, logits, grads = O.loss_and_grads(P, fe, xt, yt, wt, alpha, beta, random_chrom=int(g["chroms"][step]))
        back = np.empty_like(logits.numpy())
        back[perm] = logits.numpy()
        errs.append(logit_err(back, g[f"logits{step}"]))
        opt.step(P, grads)
    pn = {}
    for key in g.files:
        if key.startswith(f"param{n_steps - 1}/"):
            n = key.split("/", 1)[1]
            pn[n] = float(np.abs(P[n].detach().numpy() - g[key]).max())
    return errs, pn
