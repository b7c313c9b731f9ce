"""Training driver (matcha_amd/train.py == the flow of the reference's main.py): host logic on CPU, a short end-to-end
run on the GPU writing the reference's output files."""
import json
import os

import numpy as np
import pytest
import torch

from matcha_amd import synth, utils as U
from tests.helpers import gold


def _write_temp_dir(tmp, layout="tiny", ks=(2, 3), m=400, d=16, seed=0):
    num = synth.LAYOUTS[layout]
    N = int(np.sum(num))
    rng = np.random.default_rng(seed)
    temp = os.path.join(tmp, "Temp")
    os.makedirs(temp)
    cr = synth.chrom_range(num)
    np.save(os.path.join(temp, "chrom_range.npy"), cr)
    n2c = synth.node2chrom(num)
    np.save(os.path.join(temp, "node2chrom.npy"), {int(i): int(n2c[i]) for i in range(1, N + 1)}, allow_pickle=True)
    for k in ks:
        np.save(os.path.join(temp, "all_%d_counter.npy" % k), synth.make_edges(rng, N, k, m))
        np.save(os.path.join(temp, "all_%d_freq_counter.npy" % k), synth.make_freq(rng, m))
    intra, inter = synth.make_adjacency(rng, num)
    np.save(os.path.join(temp, "intra_adj.npy"), intra)
    np.save(os.path.join(temp, "inter_adj.npy"), inter)
    cfg = {"cluster_path": "x", "mcool_path": "x", "resolution": 1000000, "chrom_list": ["chr%d" % (i + 1) for i in range(len(num))],
           "chrom_size": "x", "temp_dir": temp, "max_cluster_size": 25, "min_distance": 0, "k-mer_size": list(ks), "min_freq_cutoff": 2,
           "quantile_cutoff_for_positive": 0.6, "quantile_cutoff_for_unlabel": 0.4, "embed_dim": d}
    with open(os.path.join(tmp, "config.JSON"), "w") as f:
        json.dump(cfg, f)
    return cfg, num


def test_host_helpers(tmp_path):
    from matcha_amd import train as T
    g = gold("g5_tiny_preproc.npz")
    assert np.array_equal(T.get_attributes(synth.LAYOUTS["tiny"]), g["attr"])           # main.py:497-512 via the reference
    cfg, num = _write_temp_dir(str(tmp_path))
    assert U.get_config(os.path.join(tmp_path, "config.JSON"))["embed_dim"] == 16
    # utils surface
    assert isinstance(U.np2tensor_hyper([[1, 2], [3, 4]]), torch.Tensor)
    ragged = U.np2tensor_hyper([[1, 2], [3, 4, 5]])
    assert isinstance(ragged, list) and U.pad_rows([[1, 2], [3, 4, 5]]).tolist() == [[1, 2, 0], [3, 4, 5]]
    a, b = U.sync_shuffle([torch.arange(10), torch.arange(10) * 2], 4)
    assert len(a) == 4 and torch.equal(a * 2, b)
    y = torch.tensor([1., 0., 1., 0.])
    p = torch.tensor([.9, .2, .6, .4])
    s = torch.tensor([2, 2, 3, 3])
    assert U.accuracy(p, y, s) == "2 1.000 3 1.000 "
    auc, aupr = U.roc_auc_cuda(y, p, s, 3)
    assert auc.startswith("all 1.000") and aupr.split(" ")[-2] == "3"                    # the label main.py:313 parses


@pytest.mark.gpu
def test_load_kmers_on_device(tmp_path):
    """main.py:551-566 through the device quantile transform (tests/test_positives.py holds its parity tests)."""
    from matcha_amd import train as T
    cfg, num = _write_temp_dir(str(tmp_path))
    e, w = T.load_kmers(cfg["temp_dir"], [2, 3], 0.6)
    assert e.shape[1] == 3 and len(e) == len(w) and (w > 0.6).all()
    assert 0.3 * 800 < len(e) < 0.5 * 800                                                # ~40 % of rows pass the 0.6 quantile
    assert ((e != 0).sum(1) >= 2).all() and (np.diff(np.where(e == 0, 10 ** 9, e), axis=1) > 0).all()
    feats, inter = T.build_features(cfg["temp_dir"], synth.chrom_range(num))
    assert [tuple(f.shape) for f in feats] == [(n, n) for n in num] and not any(torch.isnan(f).any() for f in feats)
    assert inter.is_cuda and inter.shape == (sum(num), sum(num))


@pytest.mark.gpu
@pytest.mark.parametrize("front_end", ["table", "adj"])
def test_end_to_end_run_writes_reference_outputs(tmp_path, front_end):
    from matcha_amd import train as T
    import Modules  # noqa: F401
    cfg, num = _write_temp_dir(str(tmp_path), m=600)
    N = int(np.sum(num))
    np.random.seed(0)
    torch.manual_seed(0)
    emb_path = os.path.join(tmp_path, "embeddings.npy")
    logs = []
    model = T.run(cfg, front_end=front_end, epochs1=1, epochs2=2, batches_per_epoch=4, emb_path=emb_path, log=logs.append)
    temp = cfg["temp_dir"]
    emb = np.load(emb_path)
    assert emb.shape == (N, 16) and emb.dtype == np.float32 and np.isfinite(emb).all()
    ck = torch.load(os.path.join(temp, "model.chkpt"), map_location="cpu", weights_only=False)
    assert set(ck.keys()) == {"model_link", "epoch"} and ck["epoch"] == 1
    loaded = torch.load(os.path.join(temp, "model2load"), map_location="cuda", weights_only=False)
    x = torch.tensor([[1, 5, 9], [2, 20, 0]])
    with torch.no_grad():
        model.eval()
        loaded.eval()
        assert torch.allclose(model(x), loaded(x), atol=1e-6)
    assert any("Training" in l for l in logs) and any("Validation" in l for l in logs)
    # phase 2 learns something on separable synthetic data: bce of the last epoch is finite and below chance-level log(2)*E[w]
    last = [l for l in logs if "Training" in l][-1]
    bce = float(last.split("bce:")[1].split(",")[0])
    assert np.isfinite(bce)
    # predict surface (main.py:482-494) with ragged input
    out = T.predict(model, [[1, 5], [2, 20, 40]])
    assert out.shape == (2, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("front_end,d,layout", [("table", 64, "c23"), ("adj", 64, "c23"), ("table", 16, "tiny")])
def test_graph_replayed_epoch_equals_the_step_by_step_epoch(tmp_path, front_end, d, layout):
    """train_epoch on one GPU replays ONE captured step (positives picked by a device counter, sampler, assembly, forward, backward,
    AdamW, running sums: Session.graph_epoch).  Same kernels, same seeds, same order as the call-by-call loop: with the deterministic
    table gradient every parameter, every prediction and both loss sums come out bitwise equal; with the adj front end (float atomics
    in its weight gradients, the reconstruction chromosome read from device memory per replay) to rounding."""
    from matcha_amd import train as T
    from tests.test_hip_model import hip_model
    num = synth.LAYOUTS[layout]
    N = int(np.sum(num))
    rng = np.random.default_rng(3)
    edges = np.concatenate([np.pad(synth.make_edges(rng, N, k, 400), ((0, 0), (0, 3 - k))) for k in (2, 3)])
    weights = rng.uniform(0.6, 1.0, size=len(edges)).astype(np.float32)
    res = {}
    for graph in (False, True):
        T.GRAPH_EPOCHS = graph
        try:
            np.random.seed(5)
            torch.manual_seed(5)
            clf, _ = hip_model(num, d, front_end, 81)
            clf.train()
            sess = T.Session(clf, synth.node2chrom(num), synth.chrom_range(num).astype(np.int32), 2, 3, 0, seed=11, deterministic=True)
            sess.set_known(edges)
            assert sess.graph_ok(0.001) == graph
            out = []
            for ep in range(2):                                   # the second epoch replays the graph captured in the first
                out.append(T.train_epoch(sess, edges, weights, 1.0, 0.001, batch_size=24))
            torch.cuda.synchronize()
            res[graph] = (out, {n: p.detach().cpu().clone() for n, p in clf.named_parameters()})
        finally:
            T.GRAPH_EPOCHS = True
    (o0, p0), (o1, p1) = res[False], res[True]
    for a, b in zip(o0, o1):
        if front_end == "table":
            assert a == b
        else:
            assert abs(a[0] - b[0]) < 1e-5 and abs(a[1] - b[1]) < 1e-3 * max(1.0, abs(a[1])) and a[2:] == b[2:]
    for n in p0:
        if front_end == "table":
            assert torch.equal(p0[n], p1[n]), n
        else:
            # 32 AdamW steps on float-atomic weight gradients: the recon heads' gradients (x beta = 1e-3) sit in AdamW's eps regime,
            # where the order of the atomics decides the sign of a step -- a few lr for single elements, nothing for the bulk
            diff = (p0[n] - p1[n]).abs().reshape(-1)
            assert float(diff.max()) <= 1e-2, n
            assert float(torch.quantile(diff, 0.9)) <= 2e-4, n


@pytest.mark.gpu
def test_step_select_and_record_equal_the_torch_bookkeeping():
    """matcha_step_select / matcha_step_record (csrc/epoch_step.hip; main.py:58, :160-161, :185-188, :449-451) against the torch ops they
    replace in the captured step: the step's rows, weights, chromosome cell and seeds; sigmoid, sizes, loss sums, counter -- bit for bit,
    over several steps including a counter past the epoch's end (clamped: never out of bounds)."""
    import ctypes as C
    from matcha_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    P, L, n_steps, B = 96, 5, 7, 384
    pos = torch.randint(0, 3000, (n_steps * P, L), generator=g).to(dev)
    pos[:, 3:] *= (torch.rand(n_steps * P, 2, generator=g) < 0.5).to(dev)                    # ragged: zero padding on the right
    w = torch.rand(n_steps * P, generator=g).to(dev)
    chroms = torch.randint(0, 23, (n_steps,), generator=g, dtype=torch.int32).to(dev)
    it = torch.zeros(1, dtype=torch.long, device=dev)
    x = torch.zeros((B, L), dtype=torch.long, device=dev)
    ww = torch.ones(B, device=dev)
    cell = torch.zeros(1, dtype=torch.int32, device=dev)
    s0 = torch.tensor([11], dtype=torch.int64, device=dev)
    s1 = torch.tensor([2 ** 40 + 5], dtype=torch.int64, device=dev)
    sums = torch.zeros(2, device=dev)
    preds = torch.zeros((n_steps, B), device=dev)
    sizes = torch.zeros((n_steps, B), dtype=torch.long, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ref_sums = torch.zeros(2, device=dev)
    for step in range(n_steps + 2):                                                          # two replays past the end
        _lib.check(lib.matcha_step_select(_lib.ptr(pos), _lib.ptr(w), n_steps * P, L, _lib.ptr(it), P, _lib.ptr(x), _lib.ptr(ww), _lib.ptr(chroms),
                                          n_steps, _lib.ptr(cell), _lib.ptr(s0), _lib.ptr(s1), st), "matcha_step_select")
        if step < n_steps:
            assert torch.equal(x[:P], pos[step * P:(step + 1) * P]) and torch.equal(ww[:P], w[step * P:(step + 1) * P])
            assert int(cell) == int(chroms[step])
        assert int(s0) == 11 + step + 1 and int(s1) == 2 ** 40 + 5 + step + 1
        x[P:] = torch.randint(0, 3000, (B - P, L), generator=g).to(dev)                      # "negatives"
        logits = torch.randn(B, generator=g).to(dev) * 4
        losses = torch.rand(3, generator=g).to(dev)
        _lib.check(lib.matcha_step_record(_lib.ptr(logits), _lib.ptr(losses), _lib.ptr(x), B, L, _lib.ptr(it), n_steps, _lib.ptr(sums),
                                          _lib.ptr(preds), _lib.ptr(sizes), st), "matcha_step_record")
        row = min(step, n_steps - 1)
        assert torch.equal(preds[row], torch.sigmoid(logits)) and torch.equal(sizes[row], (x != 0).sum(1))
        ref_sums[0] += losses[0]
        ref_sums[1] += losses[1]
        assert torch.equal(sums, ref_sums) and int(it) == step + 1


@pytest.mark.gpu
@pytest.mark.parametrize("front_end,d,layout", [("table", 64, "c23"), ("adj", 64, "c23"), ("table", 16, "tiny")])
def test_graph_replayed_eval_epoch_equals_the_call_by_call_eval(front_end, d, layout):
    """eval_epoch (main.py:200-258) replays ONE captured forward-only step per batch -- device-side batch selection, sampler,
    matcha_forward(forward_only) with the loss inside, matcha_step_record (train._graph_eval) -- and must report what the eager loop over
    model(x, return_recon=True) + torch's BCE reports: same rows, same negatives (same sampler seed sequence), same metrics."""
    from matcha_amd import train as T
    from tests.test_hip_model import hip_model
    num = synth.LAYOUTS[layout]
    N = int(np.sum(num))
    rng = np.random.default_rng(4)
    edges = np.concatenate([np.pad(synth.make_edges(rng, N, k, 300), ((0, 0), (0, 3 - k))) for k in (2, 3)])
    weights = rng.uniform(0.6, 1.0, size=len(edges)).astype(np.float32)
    res = {}
    for graph in (False, True):
        T.GRAPH_EPOCHS = graph
        try:
            np.random.seed(6)
            torch.manual_seed(6)
            clf, _ = hip_model(num, d, front_end, 82)
            sess = T.Session(clf, synth.node2chrom(num), synth.chrom_range(num).astype(np.int32), 2, 3, 0, seed=12)
            sess.set_known(edges)
            out = [T.eval_epoch(sess, edges, weights, batch_size=24) for _ in range(2)]      # the second call replays the captured graph
            res[graph] = out
        finally:
            T.GRAPH_EPOCHS = True
    for a, b in zip(res[False], res[True]):
        assert abs(a[0] - b[0]) < 2e-5 * max(1.0, abs(a[0])), (a, b)                          # mean bce over the batches
        assert abs(a[1] - b[1]) < 1e-4 * max(1.0, abs(a[1])), (a, b)                          # mean reconstruction loss
        assert a[2:] == b[2:], (a, b)                                                         # accuracy / AUROC / AUPR strings
