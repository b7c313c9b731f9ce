"""adj front end (MultipleEmbedding, what the reference's main.py actually runs) on the HIP path vs the golden
fixtures of the real reference and vs the oracle.  GPU only (-m gpu)."""
import io
import os

import numpy as np
import pytest
import torch

from matcha_amd import synth
from oracle import hypersagnn as O
from oracle import rng as R
from tests.helpers import GOLD, gold, oracle_state, logit_err, rel_err
from tests.test_hip_model import GAUGE, TOL, hip_model

pytestmark = pytest.mark.gpu

ADJ_CASES = [("tiny_adj", "tiny", 16, 21), ("c1_adj", "c1", 16, 23), ("hg38_adj_d64", "hg38_1mb", 64, 25)]


@pytest.mark.parametrize("name,layout,d,seed", ADJ_CASES)
def test_g2_eval_logits_adj(name, layout, d, seed):
    g = gold(f"g2_{name}.npz")
    clf, _ = hip_model(synth.LAYOUTS[layout], d, "adj", seed)
    clf.eval()
    with torch.no_grad():
        for k in (2, 3, 4, 5):
            x = torch.from_numpy(g[f"x_k{k}"])
            np.random.seed(7)                           # the generator seeded numpy with 7 before each forward (Modules.py:192)
            lg, rc = clf(x, return_recon=True)
            assert logit_err(lg.cpu().numpy(), g[f"logits_k{k}"]) < TOL, k
            assert rel_err(rc.cpu().numpy(), g[f"recon_k{k}"]) < TOL, k
            np.random.seed(7)
            lg5, rc5 = clf(torch.nn.functional.pad(x, (0, 5 - k)), return_recon=True)
            assert logit_err(lg5.cpu().numpy(), g[f"logits_k{k}_L5"]) < TOL, k
            assert rel_err(rc5.cpu().numpy(), g[f"recon_k{k}_L5"]) < TOL, k
        np.random.seed(7)
        lg, rc = clf(torch.from_numpy(g["x_mixed"]), return_recon=True)
        assert logit_err(lg.cpu().numpy(), g["logits_mixed"]) < TOL
        assert rel_err(rc.cpu().numpy(), g["recon_mixed"]) < TOL


def test_reference_pickle_loads_and_runs_adj():
    import Modules  # noqa: F401
    out = gold("g1_tiny_adj_refinit_out.npz")
    clf = torch.load(os.path.join(GOLD, "ref_model2load_tiny_adj"), map_location="cuda", weights_only=False)
    clf.eval()
    np.random.seed(7)
    with torch.no_grad():
        lg, rc = clf(torch.from_numpy(out["x"]), return_recon=True)
    assert logit_err(lg.cpu().numpy(), out["logits"]) < TOL
    assert rel_err(rc.cpu().numpy(), out["recon"]) < TOL
    buf = io.BytesIO()
    torch.save(clf, buf)
    buf.seek(0)
    clf2 = torch.load(buf, map_location="cuda", weights_only=False)
    np.random.seed(7)
    with torch.no_grad():
        lg2 = clf2(torch.from_numpy(out["x"]))
    assert torch.allclose(lg.cpu(), lg2.cpu(), atol=1e-6)


def _train_g3_adj(name, layout, d, seed, alpha, beta, tag, n_steps, full, use_fused):
    """G3 on the adj front end: the shared driver of tests/test_hip_model.py (gradients of step 0 read from the Trainer's flat
    buffer before AdamW, parameters after the first and the last step, embeddings before / after)."""
    from tests.test_hip_model import _train_g3
    _train_g3(name, layout, d, seed, alpha, beta, tag, n_steps, full, use_fused, mode="adj")


@pytest.mark.parametrize("use_fused", [False, True])
@pytest.mark.parametrize("tag,alpha,beta", [("phase1", 0.0, 1.0), ("phase2", 1.0, 0.001)])
def test_g3_training_tiny_adj(tag, alpha, beta, use_fused):
    _train_g3_adj("tiny_adj", "tiny", 16, 31, alpha, beta, tag, 10, True, use_fused)


@pytest.mark.parametrize("use_fused", [False, True])
def test_g3_training_hg38_adj_d64(use_fused):
    _train_g3_adj("hg38_adj_d64", "hg38_1mb", 64, 42, 1.0, 0.001, "phase2", 3, False, use_fused)


def test_adj_training_dropout_masks_match_oracle_rng():
    """Train mode incl. the dropout(0.2) on the gathered feature rows: kernel masks == oracle/rng.py masks."""
    num, d = synth.LAYOUTS["tiny"], 16
    clf, _ = hip_model(num, d, "adj", 5)
    P, fe, _ = oracle_state(num, d, "adj", 5)
    clf.train()
    x, _, _ = synth.make_batch(np.random.default_rng(4), int(np.sum(num)), [2, 4, 5], 16)
    xt = torch.from_numpy(x)
    rt = clf._runtime()
    np.random.seed(3)
    chrom = int(np.random.choice(np.arange(len(num)), 1)[0])
    np.random.seed(3)
    with torch.no_grad():
        lg, rc = clf(xt, return_recon=True)
    seed = (int(torch.initial_seed()) * 1000003 + rt.seed_counter) & 0x7FFFFFFFFFFFFFFF
    T = x.size
    masks = {"adj": torch.from_numpy(R.dropout_mask(seed, R.STREAM_DROP_ADJ, O.P_DROP_ADJ, T, max(num))),
             "fc1": torch.from_numpy(R.dropout_mask(seed, R.STREAM_DROP_FC1, O.P_DROP_FC1, T, d)),
             "pff": torch.from_numpy(R.dropout_mask(seed, R.STREAM_DROP_PFF, O.P_DROP_PFF, T, d))}
    with torch.no_grad():
        ref, rref = O.classifier_forward(P, fe, xt, masks=masks, random_chrom=chrom)
    assert logit_err(lg.cpu().numpy(), ref.numpy()) < TOL
    assert rel_err(rc.cpu().numpy(), rref.numpy()) < TOL


def test_adj_backward_with_dropout_matches_oracle():
    """Gradients in train mode (all three dropouts on) against oracle autograd with the same injected masks."""
    num, d = synth.LAYOUTS["tiny"], 16
    clf, _ = hip_model(num, d, "adj", 8)
    P, fe, _ = oracle_state(num, d, "adj", 8, requires_grad=True)
    clf.train()
    x, y, w = synth.make_batch(np.random.default_rng(6), int(np.sum(num)), [2, 3, 5], 24)
    xt, yt, wt = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(w)
    rt = clf._runtime()
    np.random.seed(11)
    chrom = int(np.random.choice(np.arange(len(num)), 1)[0])
    np.random.seed(11)
    lg, rc = clf(xt, return_recon=True)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(lg, yt.cuda(), weight=wt.cuda()) + 0.01 * rc
    loss.backward()
    seed = (int(torch.initial_seed()) * 1000003 + rt.seed_counter) & 0x7FFFFFFFFFFFFFFF
    T = x.size
    masks = {"adj": torch.from_numpy(R.dropout_mask(seed, R.STREAM_DROP_ADJ, O.P_DROP_ADJ, T, max(num))),
             "fc1": torch.from_numpy(R.dropout_mask(seed, R.STREAM_DROP_FC1, O.P_DROP_FC1, T, d)),
             "pff": torch.from_numpy(R.dropout_mask(seed, R.STREAM_DROP_PFF, O.P_DROP_PFF, T, d))}
    _, _, _, logits, grads = O.loss_and_grads(P, fe, xt, yt, wt, 1.0, 0.01, random_chrom=chrom, masks=masks)
    assert logit_err(lg.detach().cpu().numpy(), logits.numpy()) < TOL
    for n, p in clf.named_parameters():
        if grads.get(n) is None or n == GAUGE:
            continue
        ref = grads[n].numpy()
        assert p.grad is not None, n
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= TOL * max(np.abs(ref).max(), 1e-3), n


def test_recon_row_count_and_explicit_recon_gradient():
    """matcha_forward reports the number of rows the recon mean ran over (losses[2]); feeding beta through the device-side
    `drecon` pointer (the data-parallel weighting path, matcha_amd/parallel.py::recon_grad_weight) gives the same
    gradients as the scalar beta."""
    import ctypes as C
    from matcha_amd import _lib
    from matcha_amd.engine import Trainer
    num = synth.LAYOUTS["tiny"]
    N = int(np.sum(num))
    x, y, w = synth.make_batch(np.random.default_rng(3), N, [2, 3, 5], 20)
    xt, yt, wt = (torch.from_numpy(a).cuda() for a in (x, y, w))
    n2c = synth.node2chrom(num)[x]
    grads = []
    for explicit in (False, True):
        clf, _ = hip_model(num, 16, "adj", 31)
        clf.eval()
        tr = Trainer(clf)
        rt = tr.rt
        B, L = xt.shape
        ws, logits = tr._buffers(B, L)
        opts = tr._opts(1.0, 0.25, 2)
        yv, wv = yt.reshape(-1).contiguous(), wt.reshape(-1).contiguous()
        _lib.check(tr.lib.matcha_forward(C.byref(rt.shape), C.byref(rt.params), C.byref(rt.frozen), C.byref(opts), _lib.ptr(xt), B, L,
                                         _lib.ptr(yv), _lib.ptr(wv), _lib.ptr(logits), _lib.ptr(tr.losses), _lib.ptr(ws), ws.numel(),
                                         rt.stream()), "matcha_forward")
        assert float(tr.losses[2]) == float(((n2c >= 0) & (n2c != 2)).sum())
        drecon = torch.full((1,), 0.25, device="cuda") if explicit else None
        _lib.check(tr.lib.matcha_backward(C.byref(rt.shape), C.byref(rt.params), C.byref(rt.frozen), C.byref(opts), _lib.ptr(xt), B, L,
                                          _lib.ptr(yv), _lib.ptr(wv), None, _lib.ptr(drecon) if explicit else None, C.byref(tr.grads),
                                          _lib.ptr(tr.touched), _lib.ptr(ws), ws.numel(), rt.stream()), "matcha_backward")
        torch.cuda.synchronize()
        grads.append(tr.gflat.clone())
    assert float(grads[0].abs().max()) > 0
    assert torch.equal(grads[0], grads[1])


def test_unfused_adj_rows_through_the_fused_front_end_kernel():
    """A caller may hand the adj front end feature rows padded to another unit than 64 floats (matcha_frozen.feat_row_pad, include/matcha_hip.h):
    the fused adj kernels are then not eligible, the per-chromosome encoders run as separate kernels and their output rows enter
    front_fwd3_kernel through its `dense` source (node row = row t of a [T, 64] buffer instead of table[id]) -- the one path of that kernel the
    other tests do not reach.  One training step at embed_dim 64 against the oracle, kernel set asserted."""
    from matcha_amd import _lib
    from matcha_amd.engine import Trainer
    from tests.test_hip_model import _trainer_grads
    old = _lib.FEAT_ROW_PAD
    _lib.FEAT_ROW_PAD = 128
    try:
        num = synth.LAYOUTS["c23"]
        clf, _ = hip_model(num, 64, "adj", 27)
        P, fe, _ = oracle_state(num, 64, "adj", 27, requires_grad=True)
        for m in clf.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        clf.train()
        x, y, w = synth.make_batch(np.random.default_rng(8), int(np.sum(num)), [2, 3, 4, 5], 300)
        xt, yt, wt = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(w)
        tr = Trainer(clf, lr=1e-3)
        with _lib.launch_log() as log:
            logits = tr.forward_backward(xt.cuda(), yt.cuda().reshape(-1), wt.cuda().reshape(-1), 1.0, 0.05, 2)
            torch.cuda.synchronize()
        ran = {k for k, n in log.counts.items() if n > 0}
        assert {"front_fwd3_kernel", "adj_encode_fwd_kernel"} <= ran and "adj_fused_fwd_kernel" not in ran, sorted(ran)
        _, bce, recon, lg_ref, g_ref = O.loss_and_grads(P, fe, xt, yt, wt, 1.0, 0.05, random_chrom=2)
        assert logit_err(logits.cpu().numpy(), lg_ref.numpy()) < TOL
        assert abs(float(tr.losses[1]) - float(recon[0])) <= TOL * max(1.0, abs(float(recon[0])))
        mine = _trainer_grads(tr, clf)
        checked = 0
        for n, gref in g_ref.items():
            if gref is None or n == GAUGE:
                continue
            assert mine[n] is not None, n
            r = gref.numpy()
            assert np.abs(mine[n].cpu().numpy() - r).max() <= TOL * max(np.abs(r).max(), 1e-3), n
            checked += 1
        assert checked >= 30
    finally:
        _lib.FEAT_ROW_PAD = old
