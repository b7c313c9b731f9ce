"""Model-level parity of the HIP path: Modules.Classifier (C ABI underneath) against the golden fixtures of the
real reference and against the oracle on the same seeded inputs.  GPU only (-m gpu)."""
import ctypes as C
import io
import os
import sys

import numpy as np
import pytest
import torch

from matcha_amd import synth, _lib
from oracle import hypersagnn as O
from oracle import rng as R
from tests.helpers import GOLD, gold, oracle_state, logit_err, rel_err, front_end, G3BIG, g3big_batch, g3big_grad_ref

pytestmark = pytest.mark.gpu
TOL = 1e-4     # north_star: <= 1e-4 rel fp32
GAUGE = "encode1.mul_head_attn.layer_norm2.bias"    # see tests/test_oracle_golden.py


def hip_model(num, d, mode, seed):
    """Our Modules.Classifier on cuda:0 with the deterministic synthetic weights."""
    import Modules as M
    attr = O.attribute_table(num)
    sd = synth.make_state_dict(np.random.default_rng(seed), num, d, mode, attr)
    N = int(np.sum(num))
    if mode == "table":
        ne = M.Wrap_Embedding(N + 1, d, padding_idx=0)
    else:
        intra, inter = synth.make_adjacency(np.random.default_rng(seed + 1000), num)
        feats = O.corrcoef_features(intra, synth.chrom_range(num))
        ne = M.MultipleEmbedding(feats, d, False, torch.as_tensor(np.cumsum(num)), synth.chrom_range(num), inter.copy())
    clf = M.Classifier(n_head=8, d_model=d, d_k=d, d_v=d, node_embedding=ne, diag_mask=True, bottle_neck=d, attribute_dict=attr)
    res = clf.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return clf.to("cuda"), sd


def test_state_dict_keys_match_reference():
    """Every key (and shape) of the reference's Classifier.state_dict(), G1."""
    for mode in ("adj", "table"):
        ref = gold(f"g1_tiny_{mode}_refinit.npz")
        clf, _ = hip_model(synth.LAYOUTS["tiny"], 16, mode, 1)
        mine = clf.state_dict()
        assert list(mine.keys()) == list(ref.files)
        for k in ref.files:
            assert tuple(mine[k].shape) == ref[k].shape, k


TABLE_CASES = [("tiny_table", "tiny", 16, 22), ("hg38_table_d64", "hg38_1mb", 64, 24)]


@pytest.mark.parametrize("name,layout,d,seed", TABLE_CASES)
def test_g2_eval_logits_table(name, layout, d, seed):
    g = gold(f"g2_{name}.npz")
    clf, _ = hip_model(synth.LAYOUTS[layout], d, "table", seed)
    clf.eval()
    with torch.no_grad():
        for k in (2, 3, 4, 5):
            x = torch.from_numpy(g[f"x_k{k}"])
            lg = clf(x).cpu().numpy()
            assert lg.shape == (len(x), 1)
            assert logit_err(lg, g[f"logits_k{k}"]) < TOL, k
            lg5 = clf(torch.nn.functional.pad(x, (0, 5 - k))).cpu().numpy()
            assert logit_err(lg5, g[f"logits_k{k}_L5"]) < TOL, k     # pad slots are attended (fact 7)
        lg, rc = clf(torch.from_numpy(g["x_mixed"]), return_recon=True)
        assert logit_err(lg.cpu().numpy(), g["logits_mixed"]) < TOL
        assert float(rc.cpu()[0]) == 0.0


def test_reference_pickle_loads_and_runs_table():
    """torch.load of a model2load written by the REFERENCE (main.py:322) -> our classes -> HIP forward."""
    import Modules  # noqa: F401  (the pickle's GLOBALs are Modules.*)
    out = gold("g1_tiny_table_refinit_out.npz")
    clf = torch.load(os.path.join(GOLD, "ref_model2load_tiny_table"), map_location="cuda", weights_only=False)
    assert type(clf).__module__ == "Modules" and type(clf).__name__ == "Classifier"
    clf.eval()
    with torch.no_grad():
        lg = clf(torch.from_numpy(out["x"]))
    assert logit_err(lg.cpu().numpy(), out["logits"]) < TOL
    assert clf.layer_norm1.weight.device.type == "cuda"       # denoise_contact.py:101 reads this
    # round trip through our own torch.save / torch.load
    buf = io.BytesIO()
    torch.save(clf, buf)
    buf.seek(0)
    clf2 = torch.load(buf, map_location="cuda", weights_only=False)
    with torch.no_grad():
        lg2 = clf2(torch.from_numpy(out["x"]))
    assert torch.equal(lg.cpu(), lg2.cpu())
    # checkpoint file of the reference (main.py:316-321) loads by key
    ck = torch.load(os.path.join(GOLD, "ref_model_chkpt_tiny_table"), map_location="cpu", weights_only=False)
    clf2.load_state_dict(ck["model_link"])


def _check_grads(g, grads, none_ref, full):
    """Gradients of step 0 against the reference's (G3): which tensors have grad None, then every element (full fixtures)
    or the norm of every tensor (the hg38-sized fixtures store norms only)."""
    assert {n for n, v in grads.items() if v is None} - {"attribute_dict_embedding.weight"} == none_ref
    checked = 0
    for n, v in grads.items():
        if v is None or n == GAUGE:
            continue
        if full:
            ref = g["grad0/" + n]
            assert np.abs(v.cpu().numpy() - ref).max() <= TOL * max(np.abs(ref).max(), 1e-3), n
        else:
            gn = float(g["gradnorm0/" + n])
            assert abs(float(v.double().norm()) - gn) <= TOL * max(gn, 1e-3), n
        checked += 1
    assert checked >= 20


def _trainer_grads(tr, clf):
    """name -> gradient view into the Trainer's flat buffer (None where the step did not touch the tensor's group: the
    tensors whose .grad is None in the reference)."""
    rt = tr.rt
    names = {id(p): n for n, p in clf.named_parameters()}
    touched = tr.touched.cpu().tolist()
    out = {n: None for n, _ in clf.named_parameters()}
    for p, o, grp in zip(rt.live, rt.seg_off_list[:-1], rt.seg_group_list):
        if touched[grp]:
            out[names[id(p)]] = tr.gflat[o:o + p.numel()].view(p.shape).clone()
    return out


def _train_g3(name, layout, d, seed, alpha, beta, tag, n_steps, full, use_fused, mode="table"):
    g = gold(f"g3_{name}_{tag}.npz")
    num = synth.LAYOUTS[layout]
    clf, sd = hip_model(num, d, mode, seed)
    for m in clf.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clf.eval()
    N = int(np.sum(num))
    np.random.seed(99)
    with torch.no_grad():
        emb0 = clf.get_node_embeddings(torch.arange(1, N + 1).view(-1, 1))[:, 0, :].cpu().numpy()
    np.testing.assert_allclose(emb0 if full else emb0[::16], g["emb_before"], rtol=0, atol=1e-6 if mode == "table" else 2e-6)
    clf.train()
    if use_fused:
        from matcha_amd.engine import Trainer
        tr = Trainer(clf, lr=1e-3)
    else:
        opt = torch.optim.AdamW(list(clf.parameters()), lr=1e-3, amsgrad=False)     # exactly main.py:630
    none_ref = set(g["grad_none"].tolist()) - {"attribute_dict_embedding.weight"}
    chroms = g["chroms"].tolist()
    np.random.seed(1234)                  # adj mode, autograd path: Classifier.forward draws random_chrom like the reference (= chroms)
    n_param_checked = 0
    for step in range(n_steps):
        x, y, w = (torch.from_numpy(g[f"{n}{step}"]).cuda() for n in "xyw")
        if use_fused:
            # Trainer.step taken apart so that the gradient buffer can be read before AdamW zeroes it
            logits = tr.forward_backward(x.contiguous(), y.reshape(-1).contiguous(), w.reshape(-1).contiguous(), alpha, beta, chroms[step])
            if step == 0:
                _check_grads(g, _trainer_grads(tr, clf), none_ref, full)
            tr.all_reduce()
            tr.optimizer_step()
            bce, recon = tr.losses[0], tr.losses[1:2]
            logits = logits.view(-1, 1)
        else:
            logits, recon = clf(x, return_recon=True)
            bce = torch.nn.functional.binary_cross_entropy_with_logits(logits, y, weight=w)
            loss = bce * alpha + recon * beta
            opt.zero_grad()
            loss.backward()
            if step == 0:
                _check_grads(g, {n: p.grad for n, p in clf.named_parameters()}, none_ref, full)
            opt.step()
        # Step 0 is pure forward parity: TOL element-wise on the logits.  Later steps sit behind AdamW updates, and AdamW's first step
        # is lr * g / (|g| + eps): a weight whose true gradient is ~1e-9 moves by whatever fraction of lr the implementation's rounding
        # noise there dictates.  Measured against an fp64 run of the oracle (the exact trajectory, tools/debug/g3_f64_steps.py): on the
        # n_attr = 24 table fixture ONE element of fc1.weight lands 0.11 lr away from exact in the HIP path (0.009 lr in the reference's
        # own fp32 run), every other parameter of every tensor stays within 0.005 lr for all ten steps, and that single weight moves the
        # later logits by 0.6 - 1.8e-4 -- with the merged heads, the four-product heads and the layer-by-layer kernels alike
        # (tools/debug/g3_noise.py: 1.8 / 1.0 / 1.2e-4; with libm's tanh instead of the one-v_exp one 0.7 / 1.3 / 1.2e-4: it is the
        # luck of one near-zero element, not a property of a formulation).  What this trajectory test can therefore hold the later
        # steps to is 2 TOL on the logits and TOL on the north-star quantity, the output PROBABILITY; the kernels themselves are held
        # to TOL at EVERY step, with the evolved weights, by test_teacher_forced_steps_match_oracle_at_every_step below.
        lg_np, lg_ref = logits.detach().cpu().numpy(), g[f"logits{step}"]
        assert logit_err(lg_np, lg_ref) < (TOL if step == 0 else 2 * TOL), step
        pr, pr_ref = 1.0 / (1.0 + np.exp(-lg_np.astype(np.float64))), 1.0 / (1.0 + np.exp(-lg_ref.astype(np.float64)))
        assert np.abs(pr - pr_ref).max() <= TOL * pr_ref.max(), step
        assert abs(float(bce) - float(g[f"bce{step}"])) < TOL * max(1.0, abs(float(g[f"bce{step}"])))
        assert abs(float(recon.reshape(-1)[0]) - float(g[f"recon{step}"][0])) < TOL * max(1.0, abs(float(g[f"recon{step}"][0]))), step
        if step in (0, n_steps - 1):
            params = dict(clf.named_parameters())
            for key in g.files:
                if key.startswith(f"param{step}/") or key.startswith(f"paramnorm{step}/"):
                    n = key.split("/", 1)[1]
                    if n == GAUGE:
                        continue
                    ref = g[key]
                    got = params[n].detach().cpu().numpy()
                    if key.startswith("paramnorm"):
                        assert abs(float(np.linalg.norm(got.astype(np.float64))) - float(ref)) <= 2 * TOL * max(float(ref), 1e-3), (step, n)
                    else:
                        # AdamW's update is lr * m / (sqrt(v) + eps): where |g| is within a few orders of eps = 1e-8 the rounding
                        # noise of the gradient itself (the reference's too) moves the step by a visible fraction of lr.  Step 0:
                        # elements whose golden gradient is above 1e-6 must agree to 2e-4 of the tensor's scale, the others to
                        # the one lr they can move.  Last step: all but 0.1 % of the elements agree, none is further off than a
                        # fifth of the lr * steps it may have moved.
                        diff = np.abs(got - ref)
                        scale = max(np.abs(ref).max(), 1e-3)
                        lr = 1e-3
                        if step == 0 and ("grad0/" + n) in g.files:
                            solid = np.abs(g["grad0/" + n]) >= 1e-6
                            assert diff[solid].max(initial=0.0) <= 2 * TOL * scale, (step, n)
                            assert diff.max() <= 1.05 * lr, (step, n)
                        else:
                            assert np.quantile(diff, 0.999) <= 2 * TOL * scale, (step, n)
                            assert diff.max() <= 0.2 * lr * (step + 1) + 2 * TOL * scale, (step, n)
                    n_param_checked += 1
    assert n_param_checked >= 40, n_param_checked        # two snapshots of every live tensor
    clf.eval()
    np.random.seed(99)
    with torch.no_grad():
        emb1 = clf.get_node_embeddings(torch.arange(1, N + 1).view(-1, 1))[:, 0, :].cpu().numpy()
    ref = g["emb_after"]
    mine = emb1 if full else emb1[::16]
    assert np.abs(mine - ref).max() <= TOL * max(1.0, np.abs(ref).max())          # final embeddings (embeddings.npy)


@pytest.mark.parametrize("use_fused", [False, True])
@pytest.mark.parametrize("tag,alpha,beta", [("phase1", 0.0, 1.0), ("phase2", 1.0, 0.001)])
def test_g3_training_tiny_table(tag, alpha, beta, use_fused):
    _train_g3("tiny_table", "tiny", 16, 31, alpha, beta, tag, 10, True, use_fused)


@pytest.mark.parametrize("use_fused", [False, True])
def test_g3_training_hg38_table_d64(use_fused):
    _train_g3("hg38_table_d64", "hg38_1mb", 64, 41, 1.0, 0.001, "phase2", 3, False, use_fused)


D64_FULL = [("c1_table_d64", "c1", 43, "table"), ("c23_table_d64", "c23", 44, "table"), ("c23_adj_d64", "c23", 45, "adj")]


@pytest.mark.parametrize("variant", ["merged", "four_product"])
@pytest.mark.parametrize("name,layout,seed,mode", D64_FULL)
def test_teacher_forced_steps_match_oracle_at_every_step(name, layout, seed, mode, variant):
    """Ten AdamW steps on the golden batches with the HIP path as the teacher: BEFORE every step the oracle takes over the HIP
    path's current parameters (fp32, bit for bit), so both compute the same function of the same weights, and logits, losses and
    every gradient element must agree to TOL at every step -- kernels on evolved weights, free of the eps-regime noise AdamW adds
    to a free-running trajectory (see _train_g3).  Both head formulations are held to the same bar."""
    from matcha_amd.engine import Trainer
    g = gold(f"g3_{name}_phase2.npz")
    num = synth.LAYOUTS[layout]
    if variant == "four_product":
        _lib.set_option("disable_merged", 1)
    try:
        clf, _ = hip_model(num, 64, mode, seed)
        for m in clf.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        clf.train()
        tr = Trainer(clf, lr=1e-3)
        P, fe, _ = oracle_state(num, 64, mode, seed, requires_grad=True)
        worst_lg, worst_g = 0.0, 0.0
        for step in range(10):
            x, y, w = (torch.from_numpy(g[f"{n}{step}"]) for n in "xyw")
            rc = int(g["chroms"][step])
            with torch.no_grad():
                for n, p in clf.named_parameters():
                    if n in P and P[n].requires_grad:
                        P[n].copy_(p.detach().cpu())
            loss, bce, recon, lg_ref, g_ref = O.loss_and_grads(P, fe, x, y, w, 1.0, 0.001, random_chrom=rc)
            lg = tr.forward_backward(x.cuda().contiguous(), y.reshape(-1).cuda().contiguous(), w.reshape(-1).cuda().contiguous(), 1.0, 0.001, rc)
            torch.cuda.synchronize()
            e = logit_err(lg.cpu().numpy(), lg_ref.numpy())
            worst_lg = max(worst_lg, e)
            assert e < TOL, (step, e)
            assert abs(float(tr.losses[0]) - float(bce)) < TOL * max(1.0, abs(float(bce))), step
            assert abs(float(tr.losses[1]) - float(recon[0])) < TOL * max(1.0, abs(float(recon[0]))), step
            for n, v in _trainer_grads(tr, clf).items():
                ref = g_ref.get(n)
                if n == GAUGE or n.startswith("attribute_dict"):
                    continue
                assert (v is None) == (ref is None), (step, n)
                if v is None:
                    continue
                ge = float((v.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-3)
                worst_g = max(worst_g, ge)
                assert ge <= TOL, (step, n, ge)
            tr.all_reduce()
            tr.optimizer_step()
        print(f"teacher-forced {name} {variant}: worst logit err {worst_lg:.1e}, worst gradient err {worst_g:.1e}")
    finally:
        _lib.set_option("disable_merged", 0)


@pytest.mark.parametrize("variant", ["merged", "four_product"])
@pytest.mark.parametrize("name,layout,seed,mode", [("c23_table_d64", "c23", 44, "table")])
def test_g3_training_d64_full_both_head_formulations(name, layout, seed, mode, variant):
    """The free-running golden trajectory on the fixture with the largest later-step deviation, with the merged heads and with the
    reference's four products per head: the same gates hold for both (the 2 TOL gate of the later steps must not be
    what lets one formulation pass)."""
    if variant == "four_product":
        _lib.set_option("disable_merged", 1)
    try:
        _train_g3(name, layout, 64, seed, 1.0, 0.001, "phase2", 10, True, True, mode=mode)
    finally:
        _lib.set_option("disable_merged", 0)


@pytest.mark.parametrize("use_fused", [False, True])
@pytest.mark.parametrize("name,layout,seed,mode", D64_FULL)
def test_g3_training_d64_full(name, layout, seed, mode, use_fused):
    """The bench's own kernel configuration (embed_dim 64: fused forward with the loss inside, saved-Q/K/V fused backward, fused
    front end, fused AdamW) against FULL reference fixtures: every gradient element of step 0 and every parameter after 1 and
    10 torch.optim.AdamW steps, n_attr = 5 (C1 bins) and n_attr = 24 (the K = 32 attribute GEMM), 192-row mixed-k and 160-row
    k = 3 batches (several tiles per launch)."""
    _train_g3(name, layout, 64, seed, 1.0, 0.001, "phase2", 10, True, use_fused, mode=mode)


@pytest.mark.parametrize("d,layout", [(32, "tiny"), (128, "c1")])
def test_backward_matches_oracle_other_dims(d, layout):
    """Gradients for embed dims the golden files do not cover (oracle autograd on the same inputs)."""
    num = synth.LAYOUTS[layout]
    clf, _ = hip_model(num, d, "table", 77)
    P, fe, _ = oracle_state(num, d, "table", 77, requires_grad=True)
    clf.eval()
    x, y, w = synth.make_batch(np.random.default_rng(3), int(np.sum(num)), [2, 3, 5], 20)
    xt, yt, wt = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(w)
    loss, bce, recon, logits, grads = O.loss_and_grads(P, fe, xt, yt, wt, 1.0, 0.0)
    lg = clf(xt)
    l2 = torch.nn.functional.binary_cross_entropy_with_logits(lg, yt.cuda(), weight=wt.cuda())
    l2.backward()
    assert logit_err(lg.detach().cpu().numpy(), logits.numpy()) < TOL
    for n, p in clf.named_parameters():
        if grads.get(n) is None or n == GAUGE:
            assert p.grad is None or n == GAUGE or not p.requires_grad
            continue
        ref = grads[n].numpy()
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= TOL * max(np.abs(ref).max(), 1e-3), n


@pytest.mark.parametrize("B,ks", [(1, [5]), (2, [5]), (3, [5]), (1, [8]), (4, [2]), (7, [2])])
def test_tiny_batches_at_embed_dim_128_run_and_match_oracle(B, ks):
    """a differentiated forward on one to three rows at embed_dim 128 failed with MATCHA_ENOMEM (the fused attention
    block's records are sized per half tile and did not fit where the layer-wise path keeps Q / K / V / P / O); such batches now run on
    the layer-by-layer kernels.  model(x) in train mode + backward against the oracle; the launch log says which path ran."""
    num = synth.LAYOUTS["c1"]
    clf, _ = hip_model(num, 128, "table", 78)
    P, fe, _ = oracle_state(num, 128, "table", 78, requires_grad=True)
    for m in clf.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clf.train()
    x, y, w = synth.make_batch(np.random.default_rng(9), int(np.sum(num)), ks, B)
    xt, yt, wt = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(w)
    loss, bce, recon, logits, grads = O.loss_and_grads(P, fe, xt, yt, wt, 1.0, 0.0)
    with _lib.launch_log() as log:
        lg = clf(xt)
        torch.nn.functional.binary_cross_entropy_with_logits(lg, yt.cuda(), weight=wt.cuda()).backward()
        torch.cuda.synchronize()
    ran = {k for k, n in log.counts.items() if n > 0}
    assert ("enc128_fwd_kernel" in ran) == ("enc128_bwd_kernel" in ran), sorted(ran)        # forward and backward agree on the path
    assert logit_err(lg.detach().cpu().numpy(), logits.numpy()) < TOL
    for n, p in clf.named_parameters():
        if grads.get(n) is None or n == GAUGE:
            continue
        ref = grads[n].numpy()
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= TOL * max(np.abs(ref).max(), 1e-3), n


def test_training_dropout_masks_match_oracle_rng():
    """Train-mode forward: the kernels' dropout masks are the counter-RNG of oracle/rng.py, bit for bit, so the
    oracle with the same (injected) masks reproduces the logits."""
    num = synth.LAYOUTS["tiny"]
    d = 16
    clf, _ = hip_model(num, d, "table", 5)
    P, fe, _ = oracle_state(num, d, "table", 5)
    clf.train()
    x, _, _ = synth.make_batch(np.random.default_rng(4), int(np.sum(num)), [2, 4, 5], 16)
    xt = torch.from_numpy(x)
    rt = clf._runtime()
    with torch.no_grad():
        lg = clf(xt).cpu().numpy()
    seed = (int(torch.initial_seed()) * 1000003 + rt.seed_counter) & 0x7FFFFFFFFFFFFFFF
    T = x.size
    masks = {"fc1": torch.from_numpy(R.dropout_mask(seed, R.STREAM_DROP_FC1, O.P_DROP_FC1, T, d)),
             "pff": torch.from_numpy(R.dropout_mask(seed, R.STREAM_DROP_PFF, O.P_DROP_PFF, T, d))}
    with torch.no_grad():
        ref, _ = O.classifier_forward(P, fe, xt, masks=masks)
    assert logit_err(lg, ref.numpy()) < TOL
    # and the masks have the right rate
    assert abs(float((masks["fc1"] == 0).float().mean()) - 0.3) < 0.05
    assert abs(float((masks["pff"] == 0).float().mean()) - 0.4) < 0.05


def test_adamw_fused_vs_torch():
    """matcha_adamw_step against torch.optim.AdamW (what main.py:630 builds), with a segment that is skipped on
    some steps (grad None -> no decay, no step increment)."""
    lib = _lib.load()
    sizes = [64, 7, 1000, 13, 4096]
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + (s + 3) // 4 * 4)
    n = offs[-1]
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    params = [p0[offs[i]:offs[i] + sizes[i]].clone().requires_grad_(True) for i in range(len(sizes))]
    opt = torch.optim.AdamW(params, lr=1e-3)
    flat = p0.clone().cuda()
    grad = torch.zeros(n, device="cuda")
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    seg_off = torch.tensor(offs, dtype=torch.int64, device="cuda")
    seg_group = torch.tensor([0, 0, 1, 0, 2], dtype=torch.int32, device="cuda")
    seg_step = torch.zeros(len(sizes), dtype=torch.int32, device="cuda")
    coef = torch.zeros(3 * len(sizes), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for step in range(25):
        touched = [1, 1 if step % 3 != 1 else 0, 1 if step % 2 == 0 else 0]
        tt = torch.tensor(touched, dtype=torch.int32, device="cuda")
        grad.zero_()
        for i, p in enumerate(params):
            if touched[int(seg_group[i])]:
                gi = torch.randn(sizes[i], generator=g) * (10.0 ** (i - 2))
                p.grad = gi.clone()
                grad[offs[i]:offs[i] + sizes[i]] = gi.cuda()
            else:
                p.grad = None
        opt.step()
        _lib.check(lib.matcha_adamw_step(_lib.ptr(flat), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), n, _lib.ptr(seg_off), len(sizes),
                                         _lib.ptr(seg_group), _lib.ptr(tt), _lib.ptr(seg_step), _lib.ptr(coef), 1e-3, 0.9, 0.999, 1e-8,
                                         1e-2, 1.0, st))
        torch.cuda.synchronize()
        assert float(grad.abs().max()) == 0.0          # zero_grad fused
    got = flat.cpu()
    for i, p in enumerate(params):
        ref = p.detach()
        assert (got[offs[i]:offs[i] + sizes[i]] - ref).abs().max() <= 2e-6 * max(1.0, float(ref.abs().max())), i


def test_second_backward_through_one_forward_is_refused():
    """matcha_backward overwrites the saved activations with their gradients (include/matcha_hip.h): like torch without
    retain_graph, a second backward through the same forward must fail loudly, not return garbage."""
    num = synth.LAYOUTS["tiny"]
    clf, _ = hip_model(num, 128, "table", 4)
    clf.train()
    x, y, _ = synth.make_batch(np.random.default_rng(0), int(np.sum(num)), [2, 3], 8)
    lg = clf(torch.from_numpy(x))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(lg, torch.from_numpy(y).cuda())
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        loss.backward()


@pytest.mark.parametrize("name,mode,seed", [("hg38_table_d64", "table", 46), ("hg38_adj_d64", "adj", 47)])
def test_g3g_elementwise_gradients_at_the_hg38_layout(name, mode, seed):
    """Round 4 fixture g3g_*: every gradient element of the front-end tensors (3068 x 64 table; 23 per-chromosome encoders, the
    recon head of the drawn chromosome; attribute_nn with n_attr = 24; next_w) and of the small encoder tensors, every 8th element
    of the four [8d, d] matrices, at the true hg38 1 Mb layout -- the fused Trainer path (the bench's kernels) against the reference."""
    from matcha_amd.engine import Trainer
    g = gold(f"g3g_{name}.npz")
    clf, _ = hip_model(synth.LAYOUTS["hg38_1mb"], 64, mode, seed)
    for m in clf.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clf.train()
    tr = Trainer(clf, lr=1e-3)
    x, y, w = (torch.from_numpy(g[f"{n}0"]).cuda() for n in "xyw")
    logits = tr.forward_backward(x.contiguous(), y.reshape(-1).contiguous(), w.reshape(-1).contiguous(), 1.0, 0.001, int(g["chroms"][0]))
    torch.cuda.synchronize()
    assert logit_err(logits.cpu().numpy(), g["logits0"]) < TOL
    assert abs(float(tr.losses[0]) - float(g["bce0"])) < TOL * max(1.0, abs(float(g["bce0"])))
    assert abs(float(tr.losses[1]) - float(g["recon0"][0])) < TOL * max(1.0, abs(float(g["recon0"][0])))
    grads = _trainer_grads(tr, clf)
    assert {n for n, v in grads.items() if v is None} - {"attribute_dict_embedding.weight"} == set(g["grad_none"].tolist()) - {"attribute_dict_embedding.weight"}
    checked = 0
    for n, v in grads.items():
        if v is None or n == GAUGE:
            continue
        if ("grad0/" + n) in g.files:
            ref, got = g["grad0/" + n], v.cpu().numpy()
        else:
            ref, got = g["grad0s8/" + n], v.cpu().numpy().reshape(-1)[::8]
        assert np.abs(got - ref).max() <= TOL * max(np.abs(ref).max(), 1e-3), n
        checked += 1
    assert checked >= 25


# kernels a g3big case MUST run / must NOT run (matcha_launch_log): the library picks kernels by batch size and embed_dim, and these
# tests are about the ones it picks at bench sizes -- if a size rule moves a case onto another kernel, the case fails instead of
# silently testing something else (round-5 review: the "bench configuration" test had moved onto the small-batch forward).
_BIG64 = ({"fused_fwd32_kernel", "tail_bwd64_kernel", "fused_bwdh_kernel", "fbm_reduce_kernel", "fbm_chain_kernel"}, {"fused_fwd32h_kernel", "plan_small_kernel"})
G3BIG_KERNELS = {
    "hg38_table_d64_k5": (_BIG64[0] | {"front_fwd3_kernel", "front_bwd_kernel"}, _BIG64[1] | {"embed_fwd_kernel", "front_fwd_kernel"}),
    "hg38_adj_d64_k5": (_BIG64[0] | {"adj_fused_fwd_kernel", "adj_recon_kernel", "adj_fused_bwd_kernel"}, _BIG64[1] | {"adj_encode_fwd_kernel"}),
    "c23_table_d64_k8": (_BIG64[0] | {"front_fwd3_kernel", "front_bwd_kernel"}, _BIG64[1]),
    "c1_table_d64_k8_small": ({"fused_fwd32h_kernel", "plan_small_kernel", "fused_bwdh_kernel", "embed_fwd_kernel"}, {"fused_fwd32_kernel", "tail_bwd64_kernel"}),    # (n_attr = 5: the front end as separate kernels)
    "c1_table_d128_k5": ({"enc128_fwd_kernel", "enc128_bwd_kernel", "enc128_unfold_kernel"}, {"attn_fwd_wide_kernel", "attn_bwd_wide_kernel"}),
    "c1_adj_d128_k5": ({"enc128_fwd_kernel", "enc128_bwd_kernel"}, {"attn_fwd_wide_kernel", "attn_bwd_wide_kernel"}),
    "c1_table_d128_k8": ({"enc128_fwd_kernel", "enc128_bwd_kernel"}, {"attn_fwd_wide_kernel", "attn_bwd_wide_kernel"}),
    "c1_table_d256_k8": ({"gemm_wide_kernel", "gemm_tn_wide_kernel", "attn_fwd_wide_kernel", "attn_bwd_wide_kernel"}, {"enc128_fwd_kernel", "fused_fwd32_kernel"}),
}


@pytest.mark.parametrize("name", sorted(G3BIG))
def test_g3big_bench_sized_step_vs_reference(name):
    """Round 6 fixtures g3big_*: one dropout-free training step of the REAL reference on 2 100 - 9 240 rows against the Trainer path
    (the bench's kernels), element by element at the north-star tolerance: logits, losses, which tensors have grad None, every stored
    gradient element; then eval-mode logits of the first rows through model(x) (the forward-only kernels).  The launch log asserts
    WHICH kernels ran: embed_dim 64 at > 512 half tiles = fused_fwd32_kernel + tail_bwd64_kernel + fused_bwdh_kernel on a full grid
    (the headline's kernel set, 31 % + 50 % of the bench step), embed_dim 128 = the fused attention block at thousands of rows,
    embed_dim 256 / k <= 8 = the wide layer-wise kernels (main.py:164-183, Modules.py:278-318)."""
    from matcha_amd.engine import Trainer
    layout, d, mode, seed = G3BIG[name]
    g = gold(f"g3big_{name}.npz")
    clf, _ = hip_model(synth.LAYOUTS[layout], d, mode, seed)
    for m in clf.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clf.train()
    tr = Trainer(clf, lr=1e-3)
    x, y, w = (t.cuda() for t in g3big_batch(g))
    with _lib.launch_log() as log:
        logits = tr.forward_backward(x.contiguous(), y.reshape(-1).contiguous(), w.reshape(-1).contiguous(), 1.0, 0.001, int(g["chroms"][0]))
        torch.cuda.synchronize()
    must, must_not = G3BIG_KERNELS[name]
    ran = {k for k, n in log.counts.items() if n > 0}
    assert must <= ran, (name, sorted(must - ran), sorted(ran))
    assert not (must_not & ran), (name, sorted(must_not & ran))
    assert logit_err(logits.cpu().numpy(), g["logits0"]) < TOL
    assert abs(float(tr.losses[0]) - float(g["bce0"])) < TOL * max(1.0, abs(float(g["bce0"])))
    assert abs(float(tr.losses[1]) - float(g["recon0"][0])) < TOL * max(1.0, abs(float(g["recon0"][0])))
    grads = _trainer_grads(tr, clf)
    assert {n for n, v in grads.items() if v is None} - {"attribute_dict_embedding.weight"} == set(g["grad_none"].tolist()) - {"attribute_dict_embedding.weight"}
    checked, worst = 0, 0.0
    for n, v in grads.items():
        if v is None or n == GAUGE:
            continue
        stride, ref = g3big_grad_ref(g, n)
        got = v.cpu().numpy().reshape(-1)[::stride]
        e = float(np.abs(got - ref).max()) / max(float(np.abs(ref).max()), 1e-3)
        worst = max(worst, e)
        assert e <= TOL, (n, e)
        checked += 1
    assert checked >= 25
    clf.eval()
    n_eval = len(g["logits_eval"])
    np.random.seed(7)                      # adj front end: Classifier.forward draws random_chrom like the reference (= chrom_eval)
    with torch.no_grad():
        lg = clf(x[:n_eval].contiguous())
    e_eval = logit_err(lg.cpu().numpy(), g["logits_eval"])
    assert e_eval < TOL
    print(f"g3big {name}: logits {logit_err(logits.cpu().numpy(), g['logits0']):.1e}, worst gradient {worst:.1e}, eval logits {e_eval:.1e}; kernels {sorted(ran)}")


@pytest.mark.parametrize("name,mode,seed", [("c23_table_d64", "table", 48), ("c23_adj_d64", "adj", 49)])
def test_g3long_fifty_step_trajectory(name, mode, seed):
    """Round 4 fixture g3long_*: fifty free-running AdamW steps (dropout off) against the reference's: the output PROBABILITY of every
    step within TOL (north_star's quantity), the final embeddings.npy within TOL; the logits themselves carry the eps-regime noise of
    AdamW discussed in _train_g3 and are held to 2 TOL over the fifty steps."""
    from matcha_amd.engine import Trainer
    g = gold(f"g3long_{name}.npz")
    num = synth.LAYOUTS["c23"]
    clf, _ = hip_model(num, 64, mode, seed)
    for m in clf.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clf.train()
    tr = Trainer(clf, lr=1e-3)
    worst_p, worst_l = 0.0, 0.0
    for step in range(50):
        x = torch.from_numpy(g[f"x{step}"].astype(np.int64)).cuda()
        y, w = (torch.from_numpy(g[f"{n}{step}"]).cuda() for n in "yw")
        logits = tr.step(x.contiguous(), y.reshape(-1).contiguous(), w.reshape(-1).contiguous(), 1.0, 0.001, int(g["chroms"][step]))[2]
        lg_np, lg_ref = logits.detach().cpu().numpy().reshape(-1, 1), g[f"logits{step}"]
        pr, pr_ref = 1.0 / (1.0 + np.exp(-lg_np.astype(np.float64))), 1.0 / (1.0 + np.exp(-lg_ref.astype(np.float64)))
        worst_p = max(worst_p, float(np.abs(pr - pr_ref).max() / pr_ref.max()))
        worst_l = max(worst_l, logit_err(lg_np, lg_ref))
        assert np.abs(pr - pr_ref).max() <= TOL * pr_ref.max(), step
        assert logit_err(lg_np, lg_ref) < 2 * TOL, step
    clf.eval()
    np.random.seed(99)
    N = int(np.sum(num))
    with torch.no_grad():
        emb = clf.get_node_embeddings(torch.arange(1, N + 1).view(-1, 1))[:, 0, :].cpu().numpy()
    ref = g["emb_after"]
    print(f"fifty steps {name}: worst probability err {worst_p:.1e}, worst logit err {worst_l:.1e}, embeddings {np.abs(emb - ref).max() / max(1.0, np.abs(ref).max()):.1e}")
    assert np.abs(emb - ref).max() <= TOL * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("layout,d,ks,rows", [("hg38_1mb", 64, (2, 3, 4, 5), 1024), ("c1", 128, (2, 3, 4, 5, 6, 7, 8), 150), ("c1", 256, (2, 5, 8), 100),
                                              # front_fwd3_kernel at the ends of the attribute widths it takes (n_attr = 4 and 32: 3 and 31 chromosomes)
                                              ([40, 33, 27], 64, (2, 3, 5), 400), ([10] * 16 + [7] * 15, 64, (2, 4, 5), 400)])
def test_eval_logits_vs_the_plain_c_forward_twin(layout, d, ks, rows):
    """model(x) in eval mode against oracle/c/head_cpu.c (matcha_forward_cpu: the plain-C statement of Classifier.forward for the table front
    end, pinned to the reference's G2 / g3big logits by tests/test_cpu_twins.py) on a few thousand mixed-k rows at embed_dim 64 / 128 / 256."""
    from tests.test_cpu_twins import _forward_cpu
    num = synth.LAYOUTS[layout] if isinstance(layout, str) else layout
    N = int(np.sum(num))
    clf, sd = hip_model(num, d, "table", 91)
    clf.eval()
    x, _, _ = synth.make_batch(np.random.default_rng(4), N, list(ks), rows)
    with _lib.launch_log() as log, torch.no_grad():
        lg = clf(torch.from_numpy(x)).cpu().numpy().reshape(-1)
    if d == 64 and (len(num) + 1) % 4 == 0:
        assert log.counts.get("front_fwd3_kernel", 0) == 1, sorted(log.counts)
    ref = _forward_cpu(sd, num, d, x)
    assert logit_err(lg, ref) < TOL
