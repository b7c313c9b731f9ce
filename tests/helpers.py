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


def c3_sampler_case(min_dis):
    """The positives sampler_stats_c3.npz was taken on (make_golden.py::sampler_stats_c3): hg38 1 Mb, 4 000 distinct hyperedges per k in
    {2..5} (filtered to adjacent gaps > min_dis), the first 2 000 of each k as ONE shuffled mixed-k batch zero-padded to L = 5.
    Returns (batch int64 [P, 5], known = {tuple without padding}, all known rows padded [M, 5])."""
    num = synth.LAYOUTS["hg38_1mb"]
    N = int(np.sum(num))
    rng = np.random.default_rng(30 + min_dis)
    pos_by_k = {k: synth.make_edges_fast(rng, N, k, 4000) for k in (2, 3, 4, 5)}
    if min_dis:
        pos_by_k = {k: v[(np.diff(v, axis=1) > min_dis).all(axis=1)] for k, v in pos_by_k.items()}
    rows = [np.pad(r, (0, 5 - k)) for k in (2, 3, 4, 5) for r in pos_by_k[k][:2000]]
    order = np.random.default_rng(77).permutation(len(rows))
    batch = np.stack([rows[o] for o in order]).astype(np.int64)
    known_rows = np.concatenate([np.pad(v, ((0, 0), (0, 5 - k))) for k, v in pos_by_k.items()]).astype(np.int64)
    known = {tuple(int(t) for t in r if t != 0) for r in known_rows}
    return batch, known, known_rows


def c3_sampler_statistics(batch, neg, known, n2c, min_dis):
    """Per k: (histogram of the number of nodes a negative differs from its positive in, histogram of WHICH position of the positive was
    replaced), with the invariants of main.py:383-428 asserted on every negative -- the statistics make_golden.py took on the reference."""
    diff = {k: np.zeros(k + 1, dtype=np.int64) for k in (2, 3, 4, 5)}
    posh = {k: np.zeros(k, dtype=np.int64) for k in (2, 3, 4, 5)}
    for j in range(len(neg)):
        p = batch[j // 3]
        k = int((p != 0).sum())
        r = neg[j]
        assert (r[k:] == 0).all() and (r[:k] != 0).all()
        p, r = p[:k], r[:k]
        assert (np.diff(r) > min_dis).all() and tuple(r.tolist()) not in known
        assert sorted(n2c[r].tolist()) == sorted(n2c[p].tolist())
        rs = set(r.tolist())
        gone = [i for i in range(k) if int(p[i]) not in rs]
        diff[k][len(gone)] += 1
        for i in gone:
            posh[k][i] += 1
    return diff, posh


def c3_chi2_against_reference(diff, posh, min_dis):
    """chi-square distances (per k) between the statistics above and the reference's own, tests/golden/sampler_stats_c3.npz."""
    g = gold("sampler_stats_c3.npz")
    out = {}
    for k in (2, 3, 4, 5):
        rd, rp = g[f"c3_d{min_dis}_diff_k{k}"].astype(np.float64), g[f"c3_d{min_dis}_pos_k{k}"].astype(np.float64)
        assert diff[k].sum() == rd.sum()
        out[k] = (float((((diff[k] - rd) ** 2) / np.maximum(rd + diff[k], 1.0))[1:].sum()), float((((posh[k] - rp) ** 2) / np.maximum(rp + posh[k], 1.0)).sum()))
    return out
