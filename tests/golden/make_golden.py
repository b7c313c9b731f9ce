#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference, which does not exist on the GPU box).
Nothing here is imported by the tests; they read the .npz / model2load files this writes.

What it does
  * puts two dev-only shims in a temp dir on sys.path: a set-backed ``pybloom_live`` stand-in
    (the package is a third-party dependency of the reference that is not installed and not
    installable here -- parity at that boundary is UNPINNED) and nothing else;
  * imports /root/reference/Code/Modules.py unmodified, builds the reference ``Classifier`` in
    'adj' (MultipleEmbedding) and 'table' (Wrap_Embedding) modes;
  * loads the deterministic synthetic weights of matcha_amd/synth.py (so fixtures store outputs
    only), runs eval forwards, dropout-free training steps with torch.optim.AdamW exactly as
    main.py:630 builds it, and the reference's own save_embeddings / get_attributes /
    generate_negative (function bodies exec'd out of main.py's AST, because main.py has no
    __main__ guard and cannot be imported);
  * writes G1..G5 of SURVEY.md §8(c2) + sampler statistics + G6 (inference consumers, §8 f1) + G7 (k-mer generation, §8 f2).

Usage:  python tests/golden/make_golden.py
"""
import ast
import io
import math
import os
import random
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Code"
sys.path.insert(0, ROOT)
from matcha_amd import synth  # noqa: E402

_SHIM = '''
class BloomFilter:
    """exact-set stand-in (dev only): same surface the reference uses (utils.py:83-91, main.py:346)"""
    def __init__(self, capacity, error_rate=1e-3):
        self.capacity = capacity
        self._s = set()
    def add(self, k):
        self._s.add(k)
    def __contains__(self, k):
        return k in self._s
    def __len__(self):
        return len(self._s)
'''


def import_reference():
    shim = tempfile.mkdtemp(prefix="matcha_shim_")
    with open(os.path.join(shim, "pybloom_live.py"), "w") as f:
        f.write(_SHIM)
    sys.path[:0] = [shim, REF]
    with redirect_stdout(io.StringIO()):
        import Modules  # noqa
        import utils  # noqa
    return Modules, utils


def main_functions(names, glb):
    """exec selected top-level function definitions of main.py (it cannot be imported)."""
    src = open(os.path.join(REF, "main.py")).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, os.path.join(REF, "main.py"), "exec"), glb)
    return glb


def build_ref(M, num, d, mode, seed_weights, set_weights=True):
    rng = np.random.default_rng(seed_weights)
    cr = synth.chrom_range(num)
    N = int(np.sum(num))
    glb = dict(num=num, chrom_list=list(range(len(num))), np=np, print=lambda *a, **k: None)
    main_functions({"get_attributes"}, glb)
    attr = glb["get_attributes"]()
    feats = inter_z = None
    with redirect_stdout(io.StringIO()), redirect_stderr_null():
        if mode == "adj":
            arng = np.random.default_rng(seed_weights + 1000)
            intra, inter = synth.make_adjacency(arng, num)
            # main.py:571-577 (script body, restated call-for-call on the reference's numpy)
            feats = []
            for v in cr:
                t = np.corrcoef(intra[v[0] - 1:v[1] - 1, v[0] - 1:v[1] - 1]).astype("float32")
                t[np.isnan(t)] = 0.0
                feats.append(t)
            inter_in = inter.copy()
            ne = M.MultipleEmbedding(feats, d, False, torch.as_tensor(np.cumsum(num)), cr, inter_in)
            inter_z = ne.inter_initial.embedding.numpy().copy()
        else:
            ne = M.Wrap_Embedding(N + 1, d, padding_idx=0)
        clf = M.Classifier(n_head=8, d_model=d, d_k=d, d_v=d, node_embedding=ne, diag_mask=True,
                           bottle_neck=d, attribute_dict=attr)
    sd = synth.make_state_dict(rng, num, d, mode, attr)
    if set_weights:
        missing = clf.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    return clf, attr, feats, inter_z, sd


class redirect_stderr_null:
    def __enter__(self):
        self._old = sys.stderr
        sys.stderr = io.StringIO()

    def __exit__(self, *a):
        sys.stderr = self._old


def set_dropout(model, p=None):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout) and p is not None:
            m.p = p


def eval_logits(clf, x, C, seed=7):
    clf.eval()
    np.random.seed(seed)
    with torch.no_grad():
        out, recon = clf(torch.from_numpy(x), return_recon=True)
    return out.numpy().copy(), recon.numpy().copy()


def predraw_chroms(C, n, seed):
    np.random.seed(seed)
    seq = [int(np.random.choice(np.arange(C), 1)[0]) for _ in range(n)]
    np.random.seed(seed)
    return seq


def g2_eval(M, name, num, d, mode, seed):
    """G2: eval-mode logits, uniform k, mixed k, and the same rows at L=k vs L=5 (headline fact 7)."""
    clf, *_ = build_ref(M, num, d, mode, seed)
    C, N = len(num), int(np.sum(num))
    out = {}
    rng = np.random.default_rng(seed + 1)
    for k in (2, 3, 4, 5):
        e = synth.make_edges(rng, N, k, 12)
        out[f"x_k{k}"] = e
        chrom = predraw_chroms(C, 1, 7)[0]
        lg, rc = eval_logits(clf, e, C)
        out[f"logits_k{k}"], out[f"recon_k{k}"], out[f"chrom_k{k}"] = lg, rc, np.int64(chrom)
        pad = np.pad(e, ((0, 0), (0, 5 - k)))
        lg5, rc5 = eval_logits(clf, pad, C)
        out[f"logits_k{k}_L5"], out[f"recon_k{k}_L5"] = lg5, rc5
    xm, _, _ = synth.make_batch(np.random.default_rng(seed + 2), N, [2, 3, 4, 5], 6)
    out["x_mixed"] = xm
    out["chrom_mixed"] = np.int64(predraw_chroms(C, 1, 7)[0])
    out["logits_mixed"], out["recon_mixed"] = eval_logits(clf, xm, C)
    np.savez_compressed(os.path.join(HERE, f"g2_{name}.npz"), **out)
    print("G2", name, {k: v.shape for k, v in out.items() if k.startswith("logits")})


def g3_train(M, name, num, d, mode, seed, alpha, beta, tag, n_steps=10, full=True, rows=None):
    """G3/G4: dropout-free training steps with the reference model + torch.optim.AdamW (main.py:630).
    ``rows`` = (rows per k of the mixed-k steps, rows of the k = 3 steps); default (6, 24)."""
    clf, attr, feats, inter_z, sd = build_ref(M, num, d, mode, seed)
    C, N = len(num), int(np.sum(num))
    set_dropout(clf, 0.0)
    clf.train()
    opt = torch.optim.AdamW(list(clf.parameters()), lr=1e-3, amsgrad=False)    # main.py:630
    glb = dict(num_list=torch.as_tensor(np.cumsum(num)), batch_size=96, device=torch.device("cpu"),
               np=np, torch=torch, math=math)
    main_functions({"save_embeddings"}, glb)
    out = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="matcha_gold_")
    os.makedirs(os.path.join(tmp, "run"))
    os.chdir(os.path.join(tmp, "run"))
    try:
        np.random.seed(99)
        out["emb_before"] = glb["save_embeddings"](clf, True)                 # G4 (main.py:462-479)
        clf.train()
        chroms = predraw_chroms(C, n_steps, 1234)
        out["chroms"] = np.asarray(chroms, dtype=np.int64)
        brng = np.random.default_rng(seed + 3)
        for step in range(n_steps):
            r_mixed, r_k3 = rows or (6, 24)
            x, y, w = synth.make_batch(brng, N, [2, 3, 4, 5] if step % 2 == 0 else [3], r_mixed if step % 2 == 0 else r_k3)
            out[f"x{step}"], out[f"y{step}"], out[f"w{step}"] = x, y, w
            pred, recon = clf(torch.from_numpy(x), return_recon=True)        # main.py:54
            bce = torch.nn.functional.binary_cross_entropy_with_logits(pred, torch.from_numpy(y), weight=torch.from_numpy(w))
            loss = bce * alpha + recon * beta                                  # main.py:166
            opt.zero_grad()
            loss.backward()
            out[f"bce{step}"], out[f"recon{step}"] = bce.detach().numpy().copy(), recon.detach().numpy().copy()
            out[f"logits{step}"] = pred.detach().numpy().copy()
            if step == 0:
                none = []
                for n_, p in clf.named_parameters():
                    if p.grad is None:
                        none.append(n_)
                    elif full:
                        out["grad0/" + n_] = p.grad.numpy().copy()
                    else:
                        out["gradnorm0/" + n_] = np.float64(p.grad.double().norm().item())
                out["grad_none"] = np.asarray(none)
            opt.step()
            if step in (0, n_steps - 1):
                for n_, p in clf.named_parameters():
                    if p.grad is not None or step == n_steps - 1:
                        changed = not np.array_equal(p.detach().numpy(), np.asarray(sd[n_]))
                        if changed and full:
                            out[f"param{step}/" + n_] = p.detach().numpy().copy()
                        elif changed:
                            out[f"paramnorm{step}/" + n_] = np.float64(p.detach().double().norm().item())
        np.random.seed(99)
        out["emb_after"] = glb["save_embeddings"](clf, True)
        if not full:   # keep the fixture small: every 16th node
            out["emb_before"], out["emb_after"] = out["emb_before"][::16], out["emb_after"][::16]
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, f"g3_{name}_{tag}.npz"), **out)
    print("G3", name, tag, "bce0", out["bce0"], "recon0", out["recon0"], "none", len(out["grad_none"]))


def g3_grads(M, name, num, d, mode, seed, rows_per_k=48):
    """Round 4: ELEMENT-wise step-0 gradients at a layout too large for a full ten-step fixture (hg38 1 Mb: N = 3067, 23 chromosomes):
    every element of every front-end tensor (table / per-chromosome encoders / recon head / attribute_nn / next_w) and of the small
    encoder tensors, every 8th element of the four [8d, d] matrices (their full form is pinned at the c23 layout)."""
    clf, attr, feats, inter_z, sd = build_ref(M, num, d, mode, seed)
    C, N = len(num), int(np.sum(num))
    set_dropout(clf, 0.0)
    clf.train()
    out = {}
    chrom = predraw_chroms(C, 1, 4321)
    out["chroms"] = np.asarray(chrom, dtype=np.int64)
    x, y, w = synth.make_batch(np.random.default_rng(seed + 5), N, [2, 3, 4, 5], rows_per_k)
    out["x0"], out["y0"], out["w0"] = x, y, w
    pred, recon = clf(torch.from_numpy(x), return_recon=True)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(pred, torch.from_numpy(y), weight=torch.from_numpy(w))
    (bce * 1.0 + recon * 0.001).backward()
    out["bce0"], out["recon0"], out["logits0"] = bce.detach().numpy().copy(), recon.detach().numpy().copy(), pred.detach().numpy().copy()
    none = []
    for n_, p in clf.named_parameters():
        if p.grad is None:
            none.append(n_)
            continue
        gnp = p.grad.numpy()
        if n_.endswith(("w_qs.weight", "w_ks.weight", "w_vs.weight", "fc1.weight")) and "encode1" in n_:
            out["grad0s8/" + n_] = gnp.reshape(-1)[::8].copy()
        else:
            out["grad0/" + n_] = gnp.copy()
    out["grad_none"] = np.asarray(none)
    np.savez_compressed(os.path.join(HERE, f"g3g_{name}.npz"), **out)
    print("G3g", name, "bce0", out["bce0"], "recon0", out["recon0"], "tensors", len(out) - 8)


def g3_long(M, name, num, d, mode, seed, n_steps=50):
    """Round 4: a 50-step free-running trajectory (dropout off, torch.optim.AdamW): batches, logits of every step, the final
    embeddings.npy.  No parameters (the ten-step fixtures hold those)."""
    clf, attr, feats, inter_z, sd = build_ref(M, num, d, mode, seed)
    C, N = len(num), int(np.sum(num))
    set_dropout(clf, 0.0)
    clf.train()
    opt = torch.optim.AdamW(list(clf.parameters()), lr=1e-3, amsgrad=False)
    glb = dict(num_list=torch.as_tensor(np.cumsum(num)), batch_size=96, device=torch.device("cpu"), np=np, torch=torch, math=math)
    main_functions({"save_embeddings"}, glb)
    out = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="matcha_gold_")
    os.makedirs(os.path.join(tmp, "run"))
    os.chdir(os.path.join(tmp, "run"))
    try:
        chroms = predraw_chroms(C, n_steps, 1234)
        out["chroms"] = np.asarray(chroms, dtype=np.int64)
        brng = np.random.default_rng(seed + 3)
        for step in range(n_steps):
            x, y, w = synth.make_batch(brng, N, [2, 3, 4, 5] if step % 2 == 0 else [3], 24 if step % 2 == 0 else 96)
            out[f"x{step}"], out[f"y{step}"], out[f"w{step}"] = x.astype(np.int16), y, w
            pred, recon = clf(torch.from_numpy(x), return_recon=True)
            bce = torch.nn.functional.binary_cross_entropy_with_logits(pred, torch.from_numpy(y), weight=torch.from_numpy(w))
            loss = bce * 1.0 + recon * 0.001
            opt.zero_grad()
            loss.backward()
            out[f"logits{step}"] = pred.detach().numpy().copy()
            opt.step()
        np.random.seed(99)
        out["emb_after"] = glb["save_embeddings"](clf, True)
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, f"g3long_{name}.npz"), **out)
    print("G3long", name, "steps", n_steps)


def g1_g5(M):
    """G1 reference-initialised state_dict (tiny); G5 preprocessing pins; reference-pickled model2load."""
    num = synth.LAYOUTS["tiny"]
    for mode in ("adj", "table"):
        torch.manual_seed(0)
        clf, attr, feats, inter_z, _ = build_ref(M, num, 16, mode, 11, set_weights=False)
        sd = {k: v.numpy().copy() for k, v in clf.state_dict().items()}
        np.savez_compressed(os.path.join(HERE, f"g1_tiny_{mode}_refinit.npz"), **sd)
        x = synth.make_batch(np.random.default_rng(5), int(np.sum(num)), [2, 3, 4], 5)[0]
        chrom = predraw_chroms(len(num), 1, 7)[0]
        lg, rc = eval_logits(clf, x, len(num))
        np.savez_compressed(os.path.join(HERE, f"g1_tiny_{mode}_refinit_out.npz"), x=x, logits=lg, recon=rc, chrom=np.int64(chrom))
        # what main.py:322/:685 writes: the whole pickled module (class refs to `Modules.*` + tensors)
        torch.save(clf, os.path.join(HERE, f"ref_model2load_tiny_{mode}"))
        torch.save({"model_link": clf.state_dict(), "epoch": 0}, os.path.join(HERE, f"ref_model_chkpt_tiny_{mode}"))
        if mode == "adj":
            g5 = {"attr": attr, "inter_z": inter_z}
            for i, f in enumerate(feats):
                g5[f"feat{i}"] = f
            np.savez_compressed(os.path.join(HERE, "g5_tiny_preproc.npz"), **g5)
        print("G1", mode, len(sd), "keys")


def sampler_stats(M, U):
    """Run the reference's own generate_negative (main.py:361-459) on synthetic k=3 / mixed data with the
    exact-set stand-in and record the invariants of SURVEY.md §8(c3) + the differing-node histogram."""
    num = synth.LAYOUTS["hg38_1mb"]
    N = int(np.sum(num))
    cr = synth.chrom_range(num)
    n2c = synth.node2chrom(num)
    rng = np.random.default_rng(3)
    out = {}
    for k in (2, 3, 5):
        pos = synth.make_edges_fast(rng, N, k, 3000)
        from pybloom_live import BloomFilter
        dicts = [BloomFilter(10) for _ in range(k + 1)]
        for r in pos:
            dicts[k].add(tuple(int(v) for v in r))
        glb = dict(train_dict=dicts, test_dict=dicts, max_size=k, min_size=k, node2chrom={i: int(n2c[i]) for i in range(1, N + 1)},
                   chrom_range=cr, min_dis=0, task_mode="class", device=torch.device("cpu"), np=np, torch=torch,
                   math=math, random=random, np2tensor_hyper=U.np2tensor_hyper,
                   pad_sequence=torch.nn.utils.rnn.pad_sequence)
        main_functions({"generate_negative", "neighbor_check"}, glb)
        np.random.seed(5)
        random.seed(5)
        batch = pos[:1500]
        x, y, w, s = glb["generate_negative"](batch, "train_dict", np.ones(len(batch), dtype=np.float32), neg_num=3)
        x = x.numpy()
        neg = x[len(batch):]
        assert len(neg) == 3 * len(batch)
        posset = {tuple(r) for r in pos.tolist()}
        diff_hist = np.zeros(k + 1, dtype=np.int64)
        for j, r in enumerate(neg):
            p = batch[j // 3]
            assert (np.diff(r) > 0).all() and tuple(r.tolist()) not in posset
            assert sorted(n2c[r].tolist()) == sorted(n2c[p].tolist())
            diff_hist[len(set(r.tolist()) - set(p.tolist()))] += 1
        out[f"diff_hist_k{k}"] = diff_hist
        print("sampler k", k, "diff histogram", diff_hist)
    np.savez_compressed(os.path.join(HERE, "sampler_stats.npz"), **out)


def ref_functions(fname, names, glb):
    """exec selected top-level function definitions of a reference script that cannot be imported (no __main__ guard)."""
    tree = ast.parse(open(os.path.join(REF, fname)).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    exec(compile(ast.Module(body=body, type_ignores=[]), os.path.join(REF, fname), "exec"), glb)
    return glb


def g6_inference(M, U):
    """G6: the inference consumers on the reference-pickled tiny models -- predict_multiway.py's
    parse_file + predict (+ sigmoid) on a small text file, and denoise_contact.py's generate_pair_wise + predict +
    proba2matrix for two chromosomes."""
    from torch.nn.utils.rnn import pad_sequence
    num = synth.LAYOUTS["tiny"]
    cr = np.asarray(synth.chrom_range(num))
    res = 1000000
    names = [f"chr{i + 1}" for i in range(len(num))]
    bin2node, node = {}, 1
    for c, n in enumerate(num):
        for b in range(n):
            bin2node[f"{names[c]}:{b * res}"] = node
            node += 1
    rng = np.random.default_rng(66)
    lines = []
    for _ in range(40):
        k = int(rng.integers(1, 7))
        items = []
        for _ in range(k):
            c = int(rng.integers(0, len(num) + 1))                     # one value past the list: a chromosome to be skipped
            cname = names[c] if c < len(num) else "chrX"
            nb = num[c] if c < len(num) else 16
            items.append(f"{cname}:{int(rng.integers(0, nb * res))}")   # unaligned position -> floor to the bin
        if rng.random() < 0.3 and items:
            items.append(items[0])                                     # duplicate bin inside a line
        lines.append("\t".join(items))
    text = "\n".join(lines) + "\n"
    tmp = tempfile.mkdtemp(prefix="matcha_g6_")
    np.save(os.path.join(tmp, "bin2node.npy"), bin2node)
    with open(os.path.join(tmp, "in.txt"), "w") as f:
        f.write(text)
    common = dict(np=np, os=os, sys=sys, math=math, torch=torch, print=lambda *a, **k: None, trange=range,
                  np2tensor_hyper=U.np2tensor_hyper, pad_sequence=pad_sequence, device=torch.device("cpu"))
    pm = ref_functions("predict_multiway.py", {"parse_file", "predict"}, dict(common, temp_dir=tmp, chrom_list=names, res=res))
    dc = ref_functions("denoise_contact.py", {"generate_pair_wise", "proba2matrix", "predict"}, dict(common, chrom_range=cr, min_dis=2))
    parsed = pm["parse_file"](os.path.join(tmp, "in.txt"))
    samples = np.empty(len(parsed), dtype=object)
    for i, row in enumerate(parsed):
        samples[i] = list(row)
    out = {"text": np.array(text), "res": np.int64(res), "names": np.array(names),
           "bin_keys": np.array(list(bin2node.keys())), "bin_vals": np.array(list(bin2node.values()), dtype=np.int64),
           "n_samples": np.int64(len(parsed)), "sample_len": np.array([len(r) for r in parsed], dtype=np.int64),
           "samples_pad": np.array([list(r) + [0] * (8 - len(r)) for r in parsed], dtype=np.int64)}
    for mode in ("adj", "table"):
        with redirect_stdout(io.StringIO()):
            clf = torch.load(os.path.join(HERE, f"ref_model2load_tiny_{mode}"), map_location="cpu", weights_only=False)
        # predict_multiway.py:104-112 (script body): predict in chunks of 1e4 rows, padded per chunk, then sigmoid
        logits = pm["predict"](clf, samples)
        out[f"multiway_proba_{mode}"] = torch.sigmoid(torch.from_numpy(logits)).numpy()
        for cid in (0, 2):
            pw = dc["generate_pair_wise"](cid)
            lg = dc["predict"](clf, pw).reshape(-1)
            proba = torch.sigmoid(torch.from_numpy(lg)).numpy()                       # denoise_contact.py:151-153
            out[f"pairs_c{cid}"] = pw.copy()
            out[f"pair_proba_{mode}_c{cid}"] = proba
            out[f"pair_matrix_{mode}_c{cid}"] = dc["proba2matrix"](pw.copy(), None, proba)   # (mutates its first argument)
    np.savez_compressed(os.path.join(HERE, "g6_inference_tiny.npz"), **out)
    print("G6", len(parsed), "multiway samples;", {k: v.shape for k, v in out.items() if k.startswith("pair_matrix")})


def g7_kmers(M, U):
    """G7: generate_kmers.py's build_dict on a synthetic cluster file -- rows sorted lexicographically
    (the script's own order depends on worker scheduling).  Clusters: sorted unique node lists as process.py:66-77 writes."""
    from collections import Counter
    from itertools import combinations
    rng = np.random.default_rng(77)
    n_nodes, max_size = 60, 9
    clusters = []
    for _ in range(400):
        n = int(rng.integers(2, 13))                                    # some exceed max_size and must be skipped (:88)
        # draw from a few "hot" neighbourhoods so that k-mers repeat across clusters
        centre = int(rng.choice([8, 20, 33, 47]))
        pool = np.clip(centre + rng.integers(-7, 8, size=3 * n), 1, n_nodes)
        c = np.unique(pool)[:n]
        if len(c) >= 2:
            clusters.append(np.sort(c).astype(np.int64))
    out = {"n_clusters": np.int64(len(clusters)), "cl_len": np.array([len(c) for c in clusters], dtype=np.int64),
           "cl_flat": np.concatenate(clusters), "n_nodes": np.int64(n_nodes), "max_size": np.int64(max_size)}
    for min_dis, cutoff in ((0, 2), (2, 1), (1, 3)):
        for size in (2, 3, 4, 5):
            new_data = [np.array(d) for d in clusters if (len(d) >= size) & (len(d) <= max_size)]      # generate_kmers.py:86-90
            node2usefulindex = [[] for _ in range(n_nodes + 1)]
            for i, datum in enumerate(new_data):                                                       # :92-95
                for n in datum:
                    node2usefulindex[n].append(i)
            glb = dict(np=np, Counter=Counter, combinations=combinations, tqdm=lambda x: x, node2usefulindex=node2usefulindex,
                       new_data=new_data, min_dis=min_dis, min_freq_cutoff=cutoff)
            ref_functions("generate_kmers.py", {"build_dict"}, glb)
            _, rows, freq = glb["build_dict"](size, list(range(n_nodes + 1)))
            rows = np.asarray(rows, dtype=np.int64).reshape(-1, size)
            freq = np.asarray(freq, dtype=np.int64).reshape(-1)
            order = np.lexsort(rows.T[::-1]) if len(rows) else np.zeros(0, dtype=np.int64)
            out[f"kmers_d{min_dis}_c{cutoff}_k{size}"] = rows[order]
            out[f"freq_d{min_dis}_c{cutoff}_k{size}"] = freq[order]
    np.savez_compressed(os.path.join(HERE, "g7_kmers.npz"), **out)
    print("G7", {k: v.shape for k, v in out.items() if k.startswith("kmers_d0")})


def g8_positives(M, U):
    """G8: main.py:551-566 / :594-597 on synthetic frequency columns -- scikit-learn's own
    QuantileTransformer(n_quantiles=1000, output_distribution='uniform').fit_transform as the reference constructs it
    (deterministic up to 10 000 rows), and with subsample=None above that (every row fitted; the default would draw a random
    subsample).  Inputs are stored too: they are small and heavy-tailed counts are awkward to regenerate bit for bit."""
    import warnings
    import sklearn
    from sklearn.preprocessing import QuantileTransformer
    rng = np.random.default_rng(88)
    cols = {
        "counts_u": rng.integers(2, 50, size=5000).astype("float32"),                  # synth k-mer frequencies
        "counts_heavy": (np.floor(rng.pareto(1.2, size=9999)) + 2).astype("float32"),  # heavy tail, most rows tied at 2
        "short": (np.floor(rng.pareto(1.0, size=700)) + 1).astype("float32"),          # fewer rows than quantiles
        "real": rng.gamma(2.0, 1.0, size=10000).astype("float32"),                     # no ties
        "constant": np.full(37, 3.0, dtype="float32"),
        "single": np.array([5.0], dtype="float32"),
        "pair": np.array([1.0, 5.0], dtype="float32"),
        "big_counts": (np.floor(rng.pareto(1.2, size=60000)) + 2).astype("float32"),   # > 10 000 rows: subsample=None
        "big_real": rng.gamma(2.0, 1.0, size=40000).astype("float32"),
    }
    out = {"sklearn_version": np.array(sklearn.__version__), "numpy_version": np.array(np.__version__)}
    for name, col in cols.items():
        kw = {"subsample": None} if len(col) > 10000 else {}
        qt = QuantileTransformer(n_quantiles=1000, output_distribution="uniform", **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                                             # "n_quantiles is greater than ..." for the short columns
            w = qt.fit_transform(col.copy().reshape((-1, 1))).reshape((-1))            # main.py:555
        out[f"{name}_freq"] = col
        out[f"{name}_weight"] = w.astype("float32")
        out[f"{name}_quantiles"] = qt.quantiles_[:, 0].astype("float64")
    # the selection + normalisation of main.py:551-566, :594-597 for two sizes (script-level statements of main.py, which
    # cannot be exec'd in isolation: restated here statement by statement around scikit-learn's transform)
    neg_num = 3
    for size, name in ((2, "counts_u"), (3, "counts_heavy")):
        out[f"sel_data_k{size}"] = np.sort(rng.integers(1, 200, size=(len(cols[name]), size)), axis=1).astype("int")
    for cutoff in (0.6, 0.4):
        data_list, weight_list = [], []
        for size, name in ((2, "counts_u"), (3, "counts_heavy")):
            data = out[f"sel_data_k{size}"]
            weight = QuantileTransformer(n_quantiles=1000, output_distribution="uniform").fit_transform(cols[name].copy().reshape((-1, 1))).reshape((-1))
            mask = weight > cutoff
            data, weight = data[mask], weight[mask]
            data_list.append(np.pad(data, ((0, 0), (0, 3 - size))))
            weight_list.append(weight)
        weight = np.concatenate(weight_list, axis=0)
        out[f"sel_rows_c{cutoff}"] = np.concatenate(data_list, axis=0)
        out[f"sel_weight_c{cutoff}"] = weight.copy()
        weight /= np.mean(weight)
        weight *= neg_num
        out[f"sel_norm_c{cutoff}"] = weight
    np.savez_compressed(os.path.join(HERE, "g8_positives.npz"), **out)
    print("G8", {k: v.shape for k, v in out.items() if k.endswith("_weight")})


class _NpProxy:
    """numpy for the exec'd process.py functions: np.save of a ragged python list (process.py:87) needs an explicit object
    array on numpy >= 1.24 (shim 2); np.load of the pickled dicts needs allow_pickle."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def save(path, obj):
        try:
            arr = np.asanyarray(obj)
        except ValueError:
            arr = np.empty(len(obj), dtype=object)
            for i, o in enumerate(obj):
                arr[i] = o
        np.save(path, arr, allow_pickle=True)

    @staticmethod
    def load(path, **kw):
        kw["allow_pickle"] = True
        return np.load(path, **kw)


def g9_process(M, U):
    """G9: process.py's build_node_dict, parse_file and parse_cool_contact exec'd from the reference on a
    synthetic genome (3 listed chromosomes + 1 unlisted), a synthetic cluster file and a synthetic cooler -- h5py is not
    installed, so ``h5py.File`` is a dict of the arrays a cooler holds (dev-only stand-in, like the pybloom_live one) --
    plus main.py:571-575 (np.corrcoef per chromosome) and Modules.py:146-152 (scipy zscore of the positive inter entries) on
    the resulting matrices."""
    import math
    import shutil
    import types
    import pandas as pd
    import scipy.stats
    rng = np.random.default_rng(99)
    tmp = tempfile.mkdtemp(prefix="matcha_g9_")
    res = 1000000
    chrom_list = ["chr1", "chr2", "chrX"]
    size_lines = [("chr1", 9000000), ("chr1", 12300000), ("chr2", 8000000), ("chrY", 3000000), ("chrX", 5500001)]
    size_text = "".join("%s\t%d\n" % kv for kv in size_lines)
    with open(os.path.join(tmp, "sizes.txt"), "w") as f:
        f.write(size_text)
    sizes = {"chr1": 12300000, "chr2": 8000000, "chrX": 5500001, "chrY": 3000000}
    lines = []
    for c in range(300):
        n = int(rng.integers(1, 12))
        items = []
        for _ in range(n):
            chrom = str(rng.choice(["chr1", "chr1", "chr2", "chr2", "chrX", "chrY"]))
            items.append("%s:%d" % (chrom, int(rng.integers(0, sizes[chrom]))))
        if c % 17 == 0:
            items = items + items[:2]                                   # repeated items collapse (process.py:67)
        lines.append("\t".join(["cluster%d" % c] + items))
    lines.append("\t".join(["huge"] + ["chr1:%d" % int(rng.integers(0, sizes["chr1"])) for _ in range(6 * 50 + 1)]))   # > 50 * max: skipped unparsed
    lines.append("single\tchr1:5")
    lines.append("")
    cluster_text = "\n".join(lines) + "\n"
    with open(os.path.join(tmp, "x.cluster"), "w") as f:
        f.write(cluster_text)
    # a cooler: bins of every chromosome of the file (chrY included, unlisted), upper-triangle pixels, some NaN weights
    names = ["chr1", "chr2", "chrY", "chrX"]
    b_chrom, b_start = [], []
    for ci, c in enumerate(names):
        for b in range(math.ceil(sizes[c] / res)):
            b_chrom.append(ci)
            b_start.append(b * res)
    nb = len(b_chrom)
    iu, ju = np.triu_indices(nb)
    pick = rng.random(len(iu)) < 0.55
    bin1, bin2 = iu[pick].astype(np.int64), ju[pick].astype(np.int64)
    balanced = rng.gamma(2.0, 1.0, size=len(bin1)) / (np.abs(bin1 - bin2) + 1.0)
    balanced[rng.random(len(bin1)) < 0.05] = np.nan
    counts = rng.integers(1, 40, size=len(bin1)).astype(np.int32)
    out = {"sizes_text": np.array(size_text), "cluster_text": np.array(cluster_text), "bins_chrom": np.array(b_chrom, dtype=np.int32),
           "bins_start": np.array(b_start, dtype=np.int64), "chrom_names": np.array(names), "bin1": bin1, "bin2": bin2,
           "balanced": balanced, "count": counts, "max_cluster_size": np.int64(6)}
    for key, pixels in (("balanced", {"bin1_id": bin1, "bin2_id": bin2, "balanced": balanced, "count": counts}),
                        ("count", {"bin1_id": bin1, "bin2_id": bin2, "count": counts})):
        cooler = {"resolutions": {str(res): {"bins": {"chrom": np.array(b_chrom), "start": np.array(b_start)},
                                             "chroms": {"name": np.array([n.encode() for n in names])}, "pixels": pixels}}}
        glb = dict(np=_NpProxy(), pd=pd, math=math, os=os, sys=sys, tqdm=lambda x: x, trange=range, print=lambda *a, **k: None,
                   h5py=types.SimpleNamespace(File=lambda path, mode: cooler), chrom_size=os.path.join(tmp, "sizes.txt"),
                   chrom_list=chrom_list, res=res, temp_dir=tmp, cluster_path=os.path.join(tmp, "x.cluster"), mcool_path="x.mcool",
                   max_cluster_size=6)
        ref_functions("process.py", {"build_node_dict", "parse_file", "parse_cool_contact"}, glb)
        with redirect_stdout(io.StringIO()):
            glb["build_node_dict"]()
            glb["parse_file"]()
            glb["parse_cool_contact"]()
        out[f"intra_{key}"] = np.load(os.path.join(tmp, "intra_adj.npy"))
        out[f"inter_{key}"] = np.load(os.path.join(tmp, "inter_adj.npy"))
    chrom_range = np.load(os.path.join(tmp, "chrom_range.npy"))
    bin2node = np.load(os.path.join(tmp, "bin2node.npy"), allow_pickle=True).item()
    node2chrom = np.load(os.path.join(tmp, "node2chrom.npy"), allow_pickle=True).item()
    node2bin = np.load(os.path.join(tmp, "node2bin.npy"), allow_pickle=True).item()
    edge_list = np.load(os.path.join(tmp, "edge_list.npy"), allow_pickle=True)
    N = int(np.max(chrom_range)) - 1
    out["chrom_range"] = chrom_range
    out["bin2node_keys"] = np.array(list(bin2node.keys()))
    out["bin2node_vals"] = np.array(list(bin2node.values()), dtype=np.int64)
    out["node2chrom"] = np.array([node2chrom[i] for i in range(1, N + 1)], dtype=np.int64)
    out["node2bin"] = np.array([node2bin[i] for i in range(1, N + 1)])
    out["edge_len"] = np.array([len(e) for e in edge_list], dtype=np.int64)
    out["edge_flat"] = np.concatenate([np.asarray(e, dtype=np.int64) for e in edge_list])
    # main.py:568-575 and Modules.py:146-152 on the balanced matrices
    inter_initial = out["inter_balanced"].astype("float32")
    adj = out["intra_balanced"].astype("float32")
    for ci, v in enumerate(chrom_range):
        temp = adj[v[0] - 1:v[1] - 1, v[0] - 1:v[1] - 1]
        with np.errstate(invalid="ignore", divide="ignore"):
            temp = np.corrcoef(temp).astype("float32")
        temp[np.isnan(temp)] = 0.0
        out[f"corr_{ci}"] = temp
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(len(inter_initial)):
            temp = inter_initial[i, :]
            inter_initial[i, temp > 0] = scipy.stats.mstats.zscore(temp[temp > 0]).astype("float32")
    inter_initial[np.isnan(inter_initial)] = 0.0
    out["inter_zscore"] = inter_initial
    np.savez_compressed(os.path.join(HERE, "g9_process.npz"), **out)
    shutil.rmtree(tmp)
    print("G9", "N =", N, "clusters kept", len(edge_list), "pixels", len(bin1), {k: out[k].shape for k in ("intra_balanced", "corr_0")})


def round4(M):
    g3_grads(M, "hg38_table_d64", synth.LAYOUTS["hg38_1mb"], 64, "table", 46)
    g3_grads(M, "hg38_adj_d64", synth.LAYOUTS["hg38_1mb"], 64, "adj", 47)
    g3_long(M, "c23_table_d64", synth.LAYOUTS["c23"], 64, "table", 48)
    g3_long(M, "c23_adj_d64", synth.LAYOUTS["c23"], 64, "adj", 49)


def g3_big(M, name, num, d, mode, seed, ks, rows_per_k, n_eval=64):
    """Round 6 fixture g3big_*: ONE dropout-free training step of the real reference on a batch large enough for the kernels the
    library picks at bench sizes (d = 64: more than 512 half tiles -> fused_fwd32_kernel + tail_bwd64_kernel + fused_bwdh_kernel on a
    full grid; d = 128: enc128 at thousands of rows; d = 256 / k up to 8: the wide layer-wise kernels and the ML = 8 instances).
    Stored: the batch (node ids in the narrowest integer type), logits, losses, every gradient tensor -- in full up to 32 768
    elements, every s-th element above (key grad0s<s>/name) --, which tensors have grad None, and eval-mode logits of the first
    `n_eval` rows evaluated as their own batch at the same width L (the forward-only kernels; Modules.py:278-318)."""
    clf, attr, feats, inter_z, sd = build_ref(M, num, d, mode, seed)
    C, N = len(num), int(np.sum(num))
    set_dropout(clf, 0.0)
    clf.train()
    out = {}
    chrom = predraw_chroms(C, 1, 4321)
    out["chroms"] = np.asarray(chrom, dtype=np.int64)
    x, y, w = synth.make_batch(np.random.default_rng(seed + 5), N, list(ks), rows_per_k)
    out["x0"] = x.astype(np.int16 if N < 32767 else np.int32)
    out["y0"], out["w0"] = y, w
    pred, recon = clf(torch.from_numpy(x), return_recon=True)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(pred, torch.from_numpy(y), weight=torch.from_numpy(w))
    (bce * 1.0 + recon * 0.001).backward()
    out["bce0"], out["recon0"], out["logits0"] = bce.detach().numpy().copy(), recon.detach().numpy().copy(), pred.detach().numpy().copy()
    none = []
    for n_, p in clf.named_parameters():
        if p.grad is None:
            none.append(n_)
            continue
        gnp = p.grad.numpy().reshape(-1)
        stride = -(-gnp.size // 32768)
        if stride == 1:
            out["grad0/" + n_] = p.grad.numpy().copy()
        else:
            out[f"grad0s{stride}/" + n_] = gnp[::stride].copy()
    out["grad_none"] = np.asarray(none)
    clf.eval()
    np.random.seed(7)
    with torch.no_grad():
        lg, rc = clf(torch.from_numpy(x[:n_eval]), return_recon=True)
    out["logits_eval"], out["recon_eval"], out["chrom_eval"] = lg.numpy().copy(), rc.numpy().copy(), np.int64(predraw_chroms(C, 1, 7)[0])
    np.savez_compressed(os.path.join(HERE, f"g3big_{name}.npz"), **out)
    print("G3big", name, "rows", len(x), "L", x.shape[1], "bce0", out["bce0"], "recon0", out["recon0"], "tensors", sum(k.startswith("grad0") for k in out))


def g10_host_streams(M, U):
    """Round 6 fixture g10: the reference's INTEGER host work under fixed seeds -- DataGenerator.__init__ / next_iter
    (Modules.py:620-681) on uniform-k inputs (np.random.seed drives its permutations) and sync_shuffle (utils.py:142-149; torch's
    global generator drives randperm).  Inputs are regenerated by the test from the same default_rng seeds; outputs are stored."""
    out = {}
    rng = np.random.default_rng(1010)
    for k, m in ((2, 700), (3, 1300)):
        edges = synth.make_edges(rng, 300, k, m)
        weight = rng.uniform(0.1, 3.0, size=m).astype(np.float32)
        out[f"edges_k{k}"], out[f"weight_k{k}"] = edges, weight
        for bs, nb in ((96, 10), (250, 3), (96, 2)):
            np.random.seed(77 + k)
            with redirect_stdout(io.StringIO()):
                dg = M.DataGenerator(edges.copy(), weight.copy(), bs, nb, min_size=k, max_size=k, flag=(bs == 250))
            tag = f"k{k}_b{bs}_n{nb}"
            out[f"dg_len_{tag}"] = np.int64(len(dg.edges[k]))
            for it in range(5):
                with redirect_stdout(io.StringIO()):
                    e, wv = dg.next_iter()
                out[f"dg_e_{tag}_{it}"] = np.asarray(e)
                out[f"dg_w_{tag}_{it}"] = np.asarray(wv)
    for n in (1, 7, 384, 1000):
        a = np.arange(n, dtype=np.int64) * 3 + 1
        b = rng.random(n).astype(np.float32)
        torch.manual_seed(5 + n)
        sa, sb = U.sync_shuffle([torch.from_numpy(a), torch.from_numpy(b)])
        out[f"ss_in_b_{n}"] = b
        out[f"ss_a_{n}"], out[f"ss_b_{n}"] = sa.numpy().copy(), sb.numpy().copy()
        torch.manual_seed(5 + n)
        sa2, = U.sync_shuffle([torch.from_numpy(a)], min(n, 10))
        out[f"ss_a10_{n}"] = sa2.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g10_host_streams.npz"), **out)
    print("G10", len(out), "arrays")


class _RaggedNp:
    """numpy for the reference's utils.np2tensor_hyper on mixed-k input (shim 2): np.asarray / np.array of a ragged list
    falls back to a 1-D object array, which is what numpy < 1.24 returned there."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def _ragged(obj, *a, **kw):
        try:
            return np.asarray(obj, *a, **kw)
        except ValueError:
            arr = np.empty(len(obj), dtype=object)
            for i, o in enumerate(obj):
                arr[i] = o
            return arr

    asarray = _ragged
    array = _ragged


def sampler_stats_c3(M, U):
    """Round 6: statistics of the reference's own generate_negative (main.py:361-459) at the layout of BASELINE configs[2] -- hg38 1 Mb (23
    chromosomes), ONE mixed-k batch with k in {2..5} (ragged rows -> pad_sequence, size_list), min_dis 0 and 2, 2 000 positives per k,
    neg_num 3: per k the histogram of the number of nodes a negative differs in, the histogram of WHICH position of the positive was
    replaced, and the invariants of SURVEY.md 8 c3 asserted on the reference's output while the statistics are taken."""
    num = synth.LAYOUTS["hg38_1mb"]
    N = int(np.sum(num))
    cr = synth.chrom_range(num)
    n2c = synth.node2chrom(num)
    from pybloom_live import BloomFilter
    out = {}
    for min_dis in (0, 2):
        rng = np.random.default_rng(30 + min_dis)
        pos_by_k = {k: synth.make_edges_fast(rng, N, k, 4000) for k in (2, 3, 4, 5)}
        if min_dis:
            pos_by_k = {k: v[(np.diff(v, axis=1) > min_dis).all(axis=1)] for k, v in pos_by_k.items()}
        dicts = [BloomFilter(10) for _ in range(6)]
        for k, v in pos_by_k.items():
            for r in v:
                dicts[k].add(tuple(int(t) for t in r))
        glb = dict(train_dict=dicts, test_dict=dicts, max_size=5, min_size=2, node2chrom={i: int(n2c[i]) for i in range(1, N + 1)},
                   chrom_range=cr, min_dis=min_dis, task_mode="class", device=torch.device("cpu"), np=np, torch=torch,
                   math=math, random=random, np2tensor_hyper=U.np2tensor_hyper, pad_sequence=torch.nn.utils.rnn.pad_sequence)
        main_functions({"generate_negative", "neighbor_check"}, glb)
        rows = [r for k in (2, 3, 4, 5) for r in pos_by_k[k][:2000]]
        order = np.random.default_rng(77).permutation(len(rows))
        batch = np.empty(len(rows), dtype=object)
        for i, o in enumerate(order):
            batch[i] = rows[o]
        np.random.seed(6 + min_dis)
        random.seed(6 + min_dis)
        U.np = _RaggedNp()
        try:
            x, y, w, sizes = glb["generate_negative"](batch, "train_dict", np.ones(len(batch), dtype=np.float32), neg_num=3)
        finally:
            U.np = np
        x = x.numpy()
        P = len(batch)
        assert x.shape == (4 * P, 5) and sizes.shape[0] == 4 * P
        assert y[:P].min() == 1 and y[P:].max() == 0 and float(w[P:].min()) == 1.0
        known = {k: {tuple(r) for r in v.tolist()} for k, v in pos_by_k.items()}
        diff = {k: np.zeros(k + 1, dtype=np.int64) for k in (2, 3, 4, 5)}
        posh = {k: np.zeros(k, dtype=np.int64) for k in (2, 3, 4, 5)}
        for j in range(3 * P):
            p = batch[j // 3]
            k = len(p)
            r = x[P + j]
            assert (r[k:] == 0).all() and int(sizes[P + j]) == k and int(sizes[j // 3]) == k
            r = r[:k]
            assert (np.diff(r) > min_dis).all() and tuple(r.tolist()) not in known[k]
            assert sorted(n2c[r].tolist()) == sorted(n2c[p].tolist())
            gone = [i for i in range(k) if p[i] not in r]
            diff[k][len(gone)] += 1
            for i in gone:
                posh[k][i] += 1
        for k in (2, 3, 4, 5):
            out[f"c3_d{min_dis}_diff_k{k}"] = diff[k]
            out[f"c3_d{min_dis}_pos_k{k}"] = posh[k]
            print("sampler c3 min_dis", min_dis, "k", k, "diff", diff[k], "positions", posh[k])
    np.savez_compressed(os.path.join(HERE, "sampler_stats_c3.npz"), **out)


def round6(M, U):
    K5, K8 = (2, 3, 4, 5), (2, 3, 4, 5, 6, 7, 8)
    g3_big(M, "hg38_table_d64_k5", synth.LAYOUTS["hg38_1mb"], 64, "table", 61, K5, 2304)     # 9 216 rows: ~1 100 half tiles
    g3_big(M, "hg38_adj_d64_k5", synth.LAYOUTS["hg38_1mb"], 64, "adj", 62, K5, 2304)
    g3_big(M, "c23_table_d64_k8", synth.LAYOUTS["c23"], 64, "table", 63, K8, 1320)           # 9 240 rows, L = 8 (n_attr = 24: the fused front end)
    g3_big(M, "c1_table_d128_k5", synth.LAYOUTS["c1"], 128, "table", 64, K5, 1024)           # 4 096 rows (enc128)
    g3_big(M, "c1_adj_d128_k5", synth.LAYOUTS["c1"], 128, "adj", 65, K5, 1024)
    g3_big(M, "c1_table_d128_k8", synth.LAYOUTS["c1"], 128, "table", 66, K8, 600)            # 4 200 rows, L = 8
    g3_big(M, "c1_table_d256_k8", synth.LAYOUTS["c1"], 256, "table", 67, K8, 300)            # 2 100 rows, L = 8 (configs[4]'s shape)
    g3_big(M, "c1_table_d64_k8_small", synth.LAYOUTS["c1"], 64, "table", 68, K8, 16)         # 112 rows: the small-batch kernels at ML = 8
    g10_host_streams(M, U)
    sampler_stats_c3(M, U)


def main():
    torch.set_num_threads(4)
    M, U = import_reference()
    if "--round4" in sys.argv:       # only the fixtures round 4 added (the others are unchanged)
        round4(M)
        return
    if "--round6" in sys.argv:       # only the fixtures round 6 added
        round6(M, U)
        return
    g1_g5(M)
    g2_eval(M, "tiny_adj", synth.LAYOUTS["tiny"], 16, "adj", 21)
    g2_eval(M, "tiny_table", synth.LAYOUTS["tiny"], 16, "table", 22)
    g2_eval(M, "c1_adj", synth.LAYOUTS["c1"], 16, "adj", 23)
    g2_eval(M, "hg38_table_d64", synth.LAYOUTS["hg38_1mb"], 64, "table", 24)
    g2_eval(M, "hg38_adj_d64", synth.LAYOUTS["hg38_1mb"], 64, "adj", 25)
    for mode in ("adj", "table"):
        g3_train(M, f"tiny_{mode}", synth.LAYOUTS["tiny"], 16, mode, 31, 0.0, 1.0, "phase1")    # main.py:637-638
        g3_train(M, f"tiny_{mode}", synth.LAYOUTS["tiny"], 16, mode, 31, 1.0, 0.001, "phase2")  # main.py:672-673
    g3_train(M, "hg38_table_d64", synth.LAYOUTS["hg38_1mb"], 64, "table", 41, 1.0, 0.001, "phase2", n_steps=3, full=False)
    g3_train(M, "hg38_adj_d64", synth.LAYOUTS["hg38_1mb"], 64, "adj", 42, 1.0, 0.001, "phase2", n_steps=3, full=False)
    # FULL d = 64 fixtures (every gradient, parameters after 1 and 10 AdamW steps) on layouts small enough to commit: the shapes
    # the fused d = 64 kernels run -- C1's 512 bins (n_attr = 5) and a 23-chromosome layout (n_attr = 24: the K = 32 attribute GEMM)
    g3_train(M, "c1_table_d64", synth.LAYOUTS["c1"], 64, "table", 43, 1.0, 0.001, "phase2", n_steps=10, full=True, rows=(48, 160))
    g3_train(M, "c23_table_d64", synth.LAYOUTS["c23"], 64, "table", 44, 1.0, 0.001, "phase2", n_steps=10, full=True, rows=(48, 160))
    g3_train(M, "c23_adj_d64", synth.LAYOUTS["c23"], 64, "adj", 45, 1.0, 0.001, "phase2", n_steps=10, full=True, rows=(48, 160))
    sampler_stats(M, U)
    g6_inference(M, U)
    g7_kmers(M, U)
    g8_positives(M, U)
    g9_process(M, U)
    round4(M)
    round6(M, U)


if __name__ == "__main__":
    main()
