"""The attribute path's three row sources (csrc/attr_src.hpp; reference Modules.py:243-249, :263-264, main.py:497-512):
attr_mode 1 -- the table has get_attributes' structure and a token's row is rebuilt from its node id (one random row per token),
attr_mode 0 with rows padded to one 128-byte fetch unit, and attr_mode 0 on the plain [N+1, C+1] table.  The three must agree
bit for bit on a get_attributes table; a table WITHOUT that structure must take the gathering path and still match the oracle."""
import numpy as np
import pytest
import torch

from matcha_amd import synth
from oracle import hypersagnn as O
from tests.helpers import logit_err, oracle_state

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(num, d, mode, seed, attr=None):
    import Modules as M
    attr = O.attribute_table(num) if attr is None else attr
    sd = synth.make_state_dict(np.random.default_rng(seed), num, d, mode, O.attribute_table(num))
    for key in list(sd):
        if key.startswith("attribute_dict"):                     # one frozen parameter under two names (Modules.py:245-249)
            sd[key] = attr
    N = int(np.sum(num))
    if mode == "table":
        ne = M.Wrap_Embedding(N + 1, d, padding_idx=0)
    else:
        intra, inter = synth.make_adjacency(np.random.default_rng(seed + 1000), num)
        feats = O.corrcoef_features(intra, synth.chrom_range(num))
        ne = M.MultipleEmbedding(feats, d, False, torch.as_tensor(np.cumsum(num)), synth.chrom_range(num), inter.copy())
    clf = M.Classifier(n_head=8, d_model=d, d_k=d, d_v=d, node_embedding=ne, diag_mask=True, bottle_neck=d, attribute_dict=attr)
    clf.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    for m in clf.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return clf.to("cuda"), sd


def _step(clf, x, y, w, deterministic=True):
    from matcha_amd.engine import Trainer
    clf.train()
    tr = Trainer(clf, lr=1e-3, deterministic=deterministic)
    logits = tr.forward_backward(x, y, w, 1.0, 0.001, 1).clone()
    torch.cuda.synchronize()
    return tr, logits, tr.gflat.clone()


@pytest.mark.parametrize("mode,d", [("table", 64), ("table", 128), ("table", 32), ("adj", 64)])
def test_computed_padded_and_plain_attribute_rows_agree_bitwise(mode, d):
    """c23: n_attr = 24 (the K = 32 attribute GEMM of the fused front end; embed_fwd_kernel at the other dims)."""
    import matcha_amd.Modules as MM
    num = synth.LAYOUTS["c23"]
    x, y, w = synth.make_batch(np.random.default_rng(5), int(np.sum(num)), [2, 3, 5], 60)
    xt, yt, wt = (torch.from_numpy(a).cuda().contiguous() for a in (x, y.reshape(-1), w.reshape(-1)))
    outs = {}
    try:
        for name, compute, padrows in (("computed", True, True), ("padded", False, True), ("plain", False, False)):
            MM.ATTR_COMPUTE, MM.ATTR_PAD = compute, padrows
            clf, _ = _model(num, d, mode, 71)
            tr, logits, grads = _step(clf, xt, yt, wt)
            assert tr.rt.attr_mode == (1 if compute else 0)
            assert tr.rt.frozen.attr_ld == (32 if padrows else 0)      # the padded table stays available under attr_mode 1 (front_fused.hip reads it)
            with torch.no_grad():
                clf.eval()
                ev = clf(xt).clone()
            outs[name] = (logits, grads, ev)
    finally:
        MM.ATTR_COMPUTE, MM.ATTR_PAD = True, True
    ref = outs["plain"]
    for name in ("computed", "padded"):
        got = outs[name]
        if name == "computed" and mode == "table" and d == 64:
            # round 6: under attr_mode 1 the fused front end (front_fwd2_kernel) does not run the K = 32 attribute product at all -- a row
            # of get_attributes' table is one-hot || coordinate, so attribute_nn(row) = Wa[:, chrom] + coord Wa[:, C] + ba, two fused
            # multiply-adds -- and next_w runs as bf16 plane products: the same numbers to rounding, not to the bit
            for a, b, what in ((got[0], ref[0], "training logits"), (got[2], ref[2], "eval logits"), (got[1], ref[1], "gradients")):
                assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), (name, what, float((a - b).abs().max()))
            continue
        assert torch.equal(got[0], ref[0]), (name, "training logits")
        assert torch.equal(got[2], ref[2]), (name, "eval logits")
        if mode == "table":
            assert torch.equal(got[1], ref[1]), (name, "gradients")      # deterministic table gradient: every sum in a fixed order
        else:
            # adj front end: the per-chromosome weight gradients are float atomics
            assert float((got[1] - ref[1]).abs().max()) <= 1e-6 * float(ref[1].abs().max()), name


@pytest.mark.parametrize("mode,d", [("table", 64), ("table", 128), ("adj", 64)])
def test_attribute_table_without_the_structure_takes_the_gather_and_matches_the_oracle(mode, d):
    """A dense random attribute_dict (nothing one-hot about it): the runtime must not pick attr_mode 1, and the gathered rows
    (padded to 32 floats) must reproduce the oracle's logits and gradients."""
    num = synth.LAYOUTS["c23"]
    N = int(np.sum(num))
    rng = np.random.default_rng(17)
    attr = rng.normal(size=(N + 1, 24)).astype(np.float32)
    attr[0] = 0.0                                                         # row 0 is the padding row (main.py:508)
    clf, sd = _model(num, d, mode, 73, attr=attr)
    x, y, w = synth.make_batch(np.random.default_rng(6), N, [2, 3, 4, 5], 40)
    xt, yt, wt = (torch.from_numpy(a).cuda().contiguous() for a in (x, y.reshape(-1), w.reshape(-1)))
    tr, logits, grads = _step(clf, xt, yt, wt)
    assert tr.rt.attr_mode == 0 and tr.rt.frozen.attr_ld == 32
    P, fe, _ = oracle_state(num, d, mode, 73, requires_grad=True)
    P["attribute_dict_embedding.weight"] = torch.from_numpy(attr)
    loss, bce, recon, lg_ref, g_ref = O.loss_and_grads(P, fe, torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(w), 1.0, 0.001,
                                                       random_chrom=1)
    assert logit_err(logits.cpu().numpy(), lg_ref.detach().numpy()) < TOL
    rt = tr.rt
    names = {id(p): n for n, p in clf.named_parameters()}
    for p_, o in zip(rt.live, rt.seg_off_list[:-1]):
        n = names[id(p_)]
        ref = g_ref.get(n)
        if ref is None or n == "encode1.mul_head_attn.layer_norm2.bias":
            continue
        got = grads[o:o + p_.numel()].view(p_.shape).cpu()
        assert float((got - ref).abs().max()) <= TOL * max(float(ref.abs().max()), 1e-3), n


def test_one_hot_table_with_another_coordinate_is_not_computed():
    """One-hot chromosomes but a coordinate that is not index / num[0]: the bit-for-bit check must refuse attr_mode 1."""
    num = synth.LAYOUTS["c23"]
    attr = O.attribute_table(num).copy()
    attr[7, -1] = np.nextafter(attr[7, -1], np.float32(1.0))              # one ulp off in one row
    clf, _ = _model(num, 64, "table", 75, attr=attr)
    assert clf._runtime().attr_mode == 0
