"""The `*_cpu` twins of SURVEY.md section 8 b2 (oracle/c/ops_cpu.c, test infrastructure) against numpy / torch on the host: they are the
plain-C statement of what the byte / index / optimizer entry points of include/matcha_hip.h must compute; the `-m gpu` tests compare the
HIP kernels with them (tests/test_hip_kernels.py::test_hip_ops_equal_cpu_twins)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from matcha_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def twins():
    path = os.path.join(ROOT, "oracle", "_build", "libmatcha_oracle.so")
    assert os.path.exists(path), "build it with __graft_entry__.build() (make -C oracle/c)"
    return C.CDLL(path)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_gather_and_scatter_twins_vs_numpy():
    lib = twins()
    rng = np.random.default_rng(0)
    N, d, T = 300, 24, 5000
    table = rng.standard_normal((N + 1, d)).astype(np.float32)
    ids = rng.integers(0, N + 1, size=T).astype(np.int64)
    ids[7], ids[11] = N + 3, -1                                    # out of range: flagged, read as row 0
    rows = np.zeros((T, d), np.float32)
    status = np.zeros(4, np.int32)
    assert lib.matcha_gather_rows_cpu(p(ids), C.c_int64(T), C.c_int32(d), p(table), C.c_int64(N), p(rows), p(status)) == 0
    ref = table[np.where((ids < 0) | (ids > N), 0, ids)]
    assert np.array_equal(rows, ref) and status[0] == 1
    x = rng.integers(0, N + 1, size=T).astype(np.int64)
    dx0 = rng.standard_normal((T, d)).astype(np.float32)
    dtab = np.zeros((N + 1, d), np.float32)
    assert lib.matcha_embed_scatter_bwd_cpu(p(x), C.c_int64(T), C.c_int32(d), p(dx0), p(dtab)) == 0
    want = torch.zeros(N + 1, d).index_add_(0, torch.from_numpy(x), torch.from_numpy(dx0)).numpy()
    want[0] = 0
    assert np.abs(dtab - want).max() <= 1e-5 and float(np.abs(dtab[0]).max()) == 0.0


def test_adamw_twin_vs_torch_adamw():
    """Three tensors as segments of one flat buffer; the second one's group is untouched in step 2 ("grad is None": skipped, its step
    count does not advance) -- torch.optim.AdamW on the same tensors, main.py:630."""
    lib = twins()
    g = torch.Generator().manual_seed(1)
    sizes = [40, 12, 100]
    params = [torch.randn(n, generator=g, requires_grad=True) for n in sizes]
    opt = torch.optim.AdamW(params, lr=1e-3)
    flat = np.concatenate([q.detach().numpy() for q in params]).astype(np.float32)
    m, v = np.zeros_like(flat), np.zeros_like(flat)
    seg_off = np.array([0, 40, 52, 152], np.int64)
    seg_group = np.array([0, 1, 0], np.int32)
    seg_step = np.zeros(3, np.int32)
    for step in range(4):
        grads = [torch.randn(n, generator=g) * 0.1 for n in sizes]
        skip = step == 2
        for q, gr, i in zip(params, grads, range(3)):
            q.grad = None if (skip and i == 1) else gr.clone()
        opt.step()
        gflat = np.concatenate([gr.numpy() for gr in grads]).astype(np.float32)
        touched = np.array([1, 0 if skip else 1], np.int32)
        assert lib.matcha_adamw_step_cpu(p(flat), p(gflat), p(m), p(v), C.c_int64(len(flat)), p(seg_off), C.c_int32(3), p(seg_group), p(touched),
                                         p(seg_step), C.c_double(1e-3), C.c_double(0.9), C.c_double(0.999), C.c_double(1e-8), C.c_double(1e-2),
                                         C.c_double(1.0)) == 0
        ref = np.concatenate([q.detach().numpy() for q in params])
        assert np.abs(flat - ref).max() <= 2e-7, step
        if skip:
            assert np.all(gflat[40:52] != 0) and np.all(gflat[:40] == 0)       # only the updated segments had their gradients zeroed
    assert seg_step.tolist() == [4, 3, 4]


def test_hashset_twin():
    lib = twins()
    edges = np.array([[1, 5, 0, 0], [2, 3, 9, 0], [4, 6, 7, 8]], np.int64)
    rows = np.array([[2, 3, 9], [2, 3, 0], [1, 5, 0], [4, 6, 7]], np.int64)
    out = np.zeros(4, np.uint8)
    assert lib.matcha_hashset_contains_cpu(p(edges), C.c_int64(3), C.c_int32(4), p(rows), C.c_int64(4), C.c_int32(3), p(out)) == 0
    assert out.tolist() == [1, 0, 1, 0]                            # a prefix of a wider hyperedge is another hyperedge


@pytest.mark.parametrize("min_dis", [0, 2])
def test_oracle_sampler_distribution_vs_reference_at_the_c3_layout(min_dis):
    """Round 6 fixture sampler_stats_c3.npz: the reference's own generate_negative (main.py:361-459) on ONE mixed-k batch (k in {2..5},
    hg38 1 Mb, 23 chromosomes, neg_num 3, min_dis 0 / 2).  The oracle sampler draws from another random stream, so the comparison is
    distributional: per k, the histogram of how many nodes a negative differs in and of which positions were replaced -- two-sample
    chi-square, sum (a - b)^2 / (a + b), against 6 000 negatives per k (df <= 4: 30 is far in the tail) -- with every invariant of
    main.py:383-428 asserted on the way.  """
    from oracle import sampler as OS
    from tests.helpers import c3_sampler_case, c3_sampler_statistics, gold
    num = synth.LAYOUTS["hg38_1mb"]
    n2c, cr = synth.node2chrom(num), synth.chrom_range(num)
    batch, known, _ = c3_sampler_case(min_dis)
    neg = OS.sample_negatives(batch, known, n2c, cr, 3, min_dis, seed=11)
    diff, posh = c3_sampler_statistics(batch, neg, known, n2c, min_dis)
    g = gold("sampler_stats_c3.npz")
    for k in (2, 3, 4, 5):
        rd, rp = g[f"c3_d{min_dis}_diff_k{k}"].astype(np.float64), g[f"c3_d{min_dis}_pos_k{k}"].astype(np.float64)
        assert diff[k].sum() == rd.sum() == 6000 and diff[k][0] == 0
        c_d = float((((diff[k] - rd) ** 2) / np.maximum(rd + diff[k], 1.0))[1:].sum())
        c_p = float((((posh[k] - rp) ** 2) / np.maximum(rp + posh[k], 1.0)).sum())
        assert c_d < 30.0 and c_p < 30.0, (k, diff[k], rd, posh[k], rp)


def _neg_sample_cpu(pos, known_rows, n2c, cr, neg_num, min_dis, seed):
    lib = twins()
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    known_rows = np.ascontiguousarray(known_rows, dtype=np.int64)
    P, L = pos.shape
    neg = np.zeros((P * neg_num, L), dtype=np.int64)
    status = np.zeros(4, dtype=np.int32)
    n2c32 = np.ascontiguousarray(n2c, dtype=np.int32)
    cr32 = np.ascontiguousarray(cr, dtype=np.int32)
    sd = np.array([seed], dtype=np.uint64)
    lib.matcha_neg_sample_cpu.restype = C.c_int
    rc = lib.matcha_neg_sample_cpu(None, p(known_rows), C.c_int64(len(known_rows)), C.c_int32(known_rows.shape[1] if len(known_rows) else L), p(pos),
                                   C.c_int64(P), C.c_int32(L), C.c_int32(neg_num), C.c_int32(min_dis), p(n2c32), C.c_int32(len(n2c32) - 1), p(cr32),
                                   C.c_int32(len(cr32)), p(sd), p(neg), p(status))
    assert rc == 0
    return neg, status


@pytest.mark.parametrize("layout,ks,min_dis", [("tiny", [2, 3], 0), ("c1", [2, 3, 4, 5], 0), ("c1", [3], 2), ("c1", [2, 5, 8], 1)])
def test_neg_sample_cpu_twin_equals_the_python_restatement(layout, ks, min_dis):
    """oracle/c/sampler_cpu.c (the C twin of matcha_neg_sample) against oracle/sampler.py, bit for bit: two independent
    statements of main.py:361-459 on the shared counter RNG; plus the phase-1 quirk (empty set: negatives == positives, main.py:589)."""
    from oracle import sampler as OS
    num = synth.LAYOUTS[layout]
    N = int(np.sum(num))
    rng = np.random.default_rng(3)
    L = max(ks)
    pool = np.concatenate([np.pad(synth.make_edges_fast(rng, N, k, 150 if layout == "tiny" else 400), ((0, 0), (0, L - k))) for k in ks])
    known = {tuple(int(v) for v in r if v) for r in pool}
    n2c, cr = synth.node2chrom(num), synth.chrom_range(num)
    pos = pool[np.random.default_rng(2).permutation(len(pool))[:200]]
    fresh = np.pad(synth.make_edges_fast(np.random.default_rng(9), N, ks[0], 20), ((0, 0), (0, L - ks[0])))      # mostly non-members: returned unchanged
    pos = np.concatenate([pos, fresh])
    neg, status = _neg_sample_cpu(pos, pool, n2c, cr, 3, min_dis, 42)
    assert np.array_equal(neg, OS.sample_negatives(pos, known, n2c, cr, 3, min_dis, seed=42))
    # rows whose 65 536 trials were exhausted (two unchanged nodes of the positive closer than min_dis: every candidate is rejected) come back
    # equal to their positive and are counted; the reference would loop forever there (main.py:392)
    exhausted = sum(1 for n in range(len(neg)) if np.array_equal(neg[n], pos[n // 3]) and tuple(int(v) for v in pos[n // 3] if v) in known)
    assert status[0] == 0 and status[1] == exhausted and (min_dis > 0 or exhausted == 0)
    neg0, _ = _neg_sample_cpu(pos, pool[:0], n2c, cr, 3, min_dis, 42)
    assert np.array_equal(neg0, np.repeat(pos, 3, axis=0))


def _forward_cpu(sd, num, d, x):
    """matcha_forward_cpu (oracle/c/head_cpu.c) on a numpy state_dict of the table front end: the ABI structs of matcha_amd/_lib.py filled
    with HOST pointers."""
    from matcha_amd import _lib
    lib = twins()
    N = int(np.sum(num))
    keep = []

    def a(name, squeeze=False):
        arr = np.ascontiguousarray(np.asarray(sd[name], dtype=np.float32))
        if squeeze:
            arr = np.ascontiguousarray(arr.reshape(arr.shape[0], -1))
        keep.append(arr)
        return arr.ctypes.data

    t = _lib.Tensors()
    pre, pp = "encode1.mul_head_attn.", "encode1.pff_n1."
    for field, name, sq in (("table", "node_embedding.weight", False), ("attr_w", "attribute_nn.weight", False), ("attr_b", "attribute_nn.bias", False),
                            ("next_w", "next_w.FF_Linear0.weight", False), ("next_b", "next_w.FF_Linear0.bias", False),
                            ("ln_q_g", pre + "layer_norm1.weight", False), ("ln_q_b", pre + "layer_norm1.bias", False),
                            ("ln_k_g", pre + "layer_norm2.weight", False), ("ln_k_b", pre + "layer_norm2.bias", False),
                            ("ln_v_g", pre + "layer_norm3.weight", False), ("ln_v_b", pre + "layer_norm3.bias", False),
                            ("w_q", pre + "w_qs.weight", False), ("w_k", pre + "w_ks.weight", False), ("w_v", pre + "w_vs.weight", False),
                            ("fc1_w", pre + "fc1.weight", False), ("fc1_b", pre + "fc1.bias", False),
                            ("pff0_w", pp + "PWF_Conv0.weight", True), ("pff0_b", pp + "PWF_Conv0.bias", False),
                            ("pff1_w", pp + "PWF_Conv1.weight", True), ("pff1_b", pp + "PWF_Conv1.bias", False),
                            ("pff_ln_g", pp + "layer_norm.weight", False), ("pff_ln_b", pp + "layer_norm.bias", False),
                            ("ln1_g", "layer_norm1.weight", False), ("ln1_b", "layer_norm1.bias", False),
                            ("ln2_g", "layer_norm2.weight", False), ("ln2_b", "layer_norm2.bias", False),
                            ("cls_w", "pff_classifier.PWF_Conv0.weight", True), ("cls_b", "pff_classifier.PWF_Conv0.bias", False)):
        setattr(t, field, a(name, sq))
    fr = _lib.Frozen()
    fr.attr_table = a("attribute_dict_embedding.weight")
    shape = _lib.Shape(d, len(num) + 1, N, len(num), 0, max(num))
    x = np.ascontiguousarray(x, dtype=np.int64)
    out = np.zeros(len(x), dtype=np.float32)
    lib.matcha_forward_cpu.restype = C.c_int
    rc = lib.matcha_forward_cpu(C.byref(shape), C.byref(t), C.byref(fr), p(x), C.c_int64(len(x)), C.c_int32(x.shape[1]), p(out))
    assert rc == 0
    return out


@pytest.mark.parametrize("name,layout,d,seed", [("tiny_table", "tiny", 16, 22), ("hg38_table_d64", "hg38_1mb", 64, 24)])
def test_forward_cpu_twin_vs_reference_goldens(name, layout, d, seed):
    """oracle/c/head_cpu.c (the plain-C twin of matcha_forward, table front end, eval mode) against the REAL reference's
    eval logits (G2): uniform k = 2..5 at L = k and zero-padded to L = 5 (padding slots are attended: the two differ), and a mixed batch."""
    from oracle import hypersagnn as O
    from tests.helpers import gold, logit_err
    g = gold(f"g2_{name}.npz")
    num = synth.LAYOUTS[layout]
    sd = synth.make_state_dict(np.random.default_rng(seed), num, d, "table", O.attribute_table(num))
    for k in (2, 3, 4, 5):
        xk = g[f"x_k{k}"]
        assert logit_err(_forward_cpu(sd, num, d, xk), g[f"logits_k{k}"]) < 2e-5, k
        assert logit_err(_forward_cpu(sd, num, d, np.pad(xk, ((0, 0), (0, 5 - k)))), g[f"logits_k{k}_L5"]) < 2e-5, k
    assert logit_err(_forward_cpu(sd, num, d, g["x_mixed"]), g["logits_mixed"]) < 2e-5


def test_forward_cpu_twin_vs_reference_at_embed_dim_256_and_k8():
    """... and at configs[4]'s shape (embed_dim 256, k up to 8, L = 8): the eval logits of the g3big fixture."""
    from oracle import hypersagnn as O
    from tests.helpers import gold, logit_err, G3BIG
    layout, d, mode, seed = G3BIG["c1_table_d256_k8"]
    g = gold("g3big_c1_table_d256_k8.npz")
    num = synth.LAYOUTS[layout]
    sd = synth.make_state_dict(np.random.default_rng(seed), num, d, mode, O.attribute_table(num))
    n = len(g["logits_eval"])
    assert logit_err(_forward_cpu(sd, num, d, g["x0"][:n].astype(np.int64)), g["logits_eval"]) < 2e-5
