"""Deterministic embedding backward (csrc/table_grad.hip; nn.Embedding's weight gradient, Modules.py:29-34): the sort +
segmented sum against torch.index_add_ and against the float-atomic scatter it replaces, bitwise reproducibility, the C-ABI
entry point matcha_scatter_rows (reduce side of the row-sparse data-parallel exchange), and the device
status word for node ids outside the tables.  GPU only."""
import ctypes as C

import numpy as np
import pytest
import torch

from matcha_amd import _lib, synth
from tests.test_hip_model import hip_model

pytestmark = pytest.mark.gpu


def _scatter(ids, rows, n_nodes, d):
    lib = _lib.load()
    n = ids.numel()
    out = torch.zeros(n_nodes + 1, d, device="cuda")
    ws = torch.empty(lib.matcha_scatter_rows_workspace_bytes(n, d, n_nodes), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.matcha_scatter_rows(_lib.ptr(ids), _lib.ptr(rows), n, d, n_nodes, _lib.ptr(out), _lib.ptr(ws), ws.numel(), st), "matcha_scatter_rows")
    return out


@pytest.mark.parametrize("n_nodes,d,n", [(3067, 64, 229_709), (30, 16, 5000), (1_000_000, 256, 131_073), (30344, 128, 70_001), (7, 64, 3)])
def test_scatter_rows_matches_index_add_and_is_reproducible(n_nodes, d, n):
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    ids = torch.randint(0, n_nodes + 1, (n,), generator=g, device="cuda", dtype=torch.int32)          # id 0 = unused entry
    ids[torch.rand(n, generator=g, device="cuda") < 0.2] = 0
    rows = torch.randn(n, d, generator=g, device="cuda")
    got = _scatter(ids, rows, n_nodes, d)
    ref = torch.zeros(n_nodes + 1, d, dtype=torch.float64, device="cuda")
    ref.index_add_(0, ids.long(), rows.double())
    ref[0] = 0
    assert float(got[0].abs().max()) == 0.0                                     # the padding row never receives a gradient
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= 2e-6 * scale * max(1.0, (n / n_nodes) ** 0.5)
    assert torch.equal(got, _scatter(ids, rows, n_nodes, d))                    # one writer per row, fixed order: bitwise
    # accumulates into what is there (matcha_backward accumulates into grads)
    lib = _lib.load()
    ws = torch.empty(lib.matcha_scatter_rows_workspace_bytes(n, d, n_nodes), dtype=torch.uint8, device="cuda")
    twice = got.clone()
    _lib.check(lib.matcha_scatter_rows(_lib.ptr(ids), _lib.ptr(rows), n, d, n_nodes, _lib.ptr(twice), _lib.ptr(ws), ws.numel(),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.allclose(twice, 2 * got, rtol=1e-6, atol=1e-6 * scale)


@pytest.mark.parametrize("d,layout", [(64, "hg38_1mb"), (128, "c1")])
def test_sorted_table_gradient_equals_atomic_scatter_and_is_bitwise_reproducible(d, layout):
    """The Trainer's table gradient with the sort + segmented sum (Trainer(deterministic=True)) against the float-atomic scatter
    (the default), fused (d = 64) and layer-wise (d = 128) paths; two runs of the deterministic path are bitwise equal INCLUDING
    the table (round 1 had to exempt it)."""
    from matcha_amd.engine import Trainer
    num = synth.LAYOUTS[layout]
    N = int(np.sum(num))
    rng = np.random.default_rng(3)
    xs = [np.pad(synth.make_edges_fast(rng, N, k, 2048), ((0, 0), (0, 5 - k))) for k in (2, 3, 4, 5)]
    x = torch.from_numpy(np.concatenate(xs)[rng.permutation(8192)]).cuda()
    y = (torch.rand(len(x), device="cuda") < 0.25).float()
    w = torch.ones(len(x), device="cuda")
    grads = []
    for sorted_ in (True, True, False):
        clf, _ = hip_model(num, d, "table", 3)
        clf.train()
        tr = Trainer(clf, base_seed=5, deterministic=sorted_)
        tr.forward_backward(x, y, w, 1.0, 0.001, 0)
        torch.cuda.synchronize()
        grads.append(tr.gflat.clone())
    assert torch.equal(grads[0], grads[1])                                      # bitwise, table included
    nt = (N + 1) * d
    a, b = grads[0][:nt], grads[2][:nt]
    assert float(a.abs().max()) > 0
    assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())            # same sums up to the atomics' order
    if d == 64:
        # the fused path's default also sums the heads' d x_hat with float atomics (fused_bwdm_kernel): next_w and attribute_nn see that order
        c, e = grads[0][nt:], grads[2][nt:]
        assert float((c - e).abs().max()) <= 2e-5 * float(c.abs().max())
        assert int((c != e).sum()) <= 2 * (64 * 64 + 64) + 64 * 8 + 64         # next_w / attribute_nn and their biases at most
    else:
        # d = 128: the deterministic Trainer stays on the layer-by-layer kernels, the default runs the fused attention block (enc128.hip:
        # d x_hat through float atomics, bf16-plane products) -- equal to rounding; with the fused block switched off, the rest is bitwise
        # untouched by the deterministic switch
        c, e = grads[0][nt:], grads[2][nt:]
        assert float((c - e).abs().max()) <= 1e-4 * float(c.abs().max())
        _lib.set_option("disable_fused", 1)
        try:
            clf, _ = hip_model(num, d, "table", 3)
            clf.train()
            tr = Trainer(clf, base_seed=5, deterministic=False)
            tr.forward_backward(x, y, w, 1.0, 0.001, 0)
            torch.cuda.synchronize()
        finally:
            _lib.set_option("disable_fused", 0)
        assert torch.equal(grads[0][nt:], tr.gflat[nt:])


def test_out_of_range_ids_are_flagged_not_dereferenced():
    """ids outside [0, N] used to index the tables directly.  Now they are read as the padding id, the
    status word records it, and the python surface raises the reference's IndexError."""
    from matcha_amd.engine import Trainer
    num = synth.LAYOUTS["tiny"]
    N = int(np.sum(num))
    for mode in ("table", "adj"):
        clf, _ = hip_model(num, 64, mode, 3)
        clf.eval()
        good = torch.tensor([[1, 5, 9], [2, 20, 0]]).cuda()
        with torch.no_grad():
            ref = clf(good)
        for bad_id in (N + 1, 10 ** 12, -3):
            bad = good.clone()
            bad[1, 1] = bad_id
            with pytest.raises(IndexError), torch.no_grad():
                clf(bad)
            with pytest.raises(IndexError), torch.no_grad():
                clf.get_node_embeddings(bad)
            with torch.no_grad():
                assert torch.equal(clf(good), ref)                               # the flag was cleared; good input works again
        clf.train()
        tr = Trainer(clf)
        bad = good.clone()
        bad[0, 0] = N + 7
        tr.step(bad, torch.ones(2, device="cuda"), torch.ones(2, device="cuda"))
        with pytest.raises(IndexError):
            tr.check_status()
        tr.step(good, torch.ones(2, device="cuda"), torch.ones(2, device="cuda"))
        tr.check_status()
        for n, p in clf.named_parameters():
            assert bool(torch.isfinite(p).all()), n


def test_sampler_flags_nodes_without_chromosome_and_counts_exhausted_rows():
    from matcha_amd.sampler import HyperedgeSet, NegativeSampler
    num = synth.LAYOUTS["tiny"]
    n2c, cr = synth.node2chrom(num), synth.chrom_range(num)
    pos = torch.tensor([[1, 5, 9], [2, 20, 40]]).cuda()
    hs = HyperedgeSet(pos)
    broken = n2c.copy()
    broken[5] = -1                                            # what train.run's default fill leaves for ids missing from node2chrom.npy
    smp = NegativeSampler(hs, broken, cr, neg_num=8, seed=1)
    smp.sample(pos)
    with pytest.raises(KeyError):
        smp.check_status()
    # a chromosome too small for the min-distance rule: every trial fails, rows come back equal to the positive and are counted
    smp = NegativeSampler(hs, n2c, cr, neg_num=2, min_dis=40, seed=1)
    neg = smp.sample(pos)
    assert smp.check_status() == 4 and torch.equal(neg, pos.repeat_interleave(2, dim=0))
    assert smp.check_status() == 0


@pytest.mark.parametrize("mode,d", [("table", 64), ("adj", 64), ("table", 32)])
def test_get_embedding_matches_oracle_intermediates(mode, d):
    """Classifier.get_embedding (reference Modules.py:261-276): dynamic, static and the attention probabilities of real query
    slots against the oracle's intermediates, mixed k with padding."""
    from oracle import hypersagnn as O
    from tests.helpers import oracle_state
    num = synth.LAYOUTS["c23" if d == 64 else "tiny"]
    clf, _ = hip_model(num, d, mode, 9)
    P, fe, _ = oracle_state(num, d, mode, 9)
    clf.eval()
    x, _, _ = synth.make_batch(np.random.default_rng(4), int(np.sum(num)), [2, 3, 5], 30)
    xt = torch.from_numpy(x)
    np.random.seed(3)
    chrom = int(np.random.choice(np.arange(len(num)), 1)[0])
    np.random.seed(3)
    with torch.no_grad():
        dyn, sta, attn, recon = clf.get_embedding(xt, None, None, return_recon=True)
        _, ref_recon, im = O.classifier_forward(P, fe, xt, random_chrom=chrom, return_intermediates=True)
    B, L = x.shape
    assert dyn.shape == (B, L, d) and sta.shape == (B, L, d) and attn.shape == (8 * B, L, L)
    assert float((dyn.cpu() - im["dynamic"]).abs().max()) <= 1e-4 * float(im["dynamic"].abs().max())
    assert float((sta.cpu() - im["X"]).abs().max()) <= 1e-5
    real = torch.from_numpy(x != 0)
    qmask = real.repeat(8, 1).unsqueeze(-1).float()                              # rows of padding queries are zero here
    assert float((attn.cpu() - im["attn"] * qmask).abs().max()) <= 1e-5
    assert abs(float(recon.cpu()[0]) - float(ref_recon[0])) <= 1e-4 * max(1.0, abs(float(ref_recon[0])))
