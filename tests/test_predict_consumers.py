"""the inference consumers (predict_multiway.py, denoise_contact.py's pairwise sweep) on the HIP forward,
against G6 -- outputs of the reference's own parse_file / predict / generate_pair_wise / proba2matrix on the
reference-pickled tiny models (tests/golden/make_golden.py::g6_inference)."""
import os

import numpy as np
import pytest
import torch

from matcha_amd import predict as PR
from matcha_amd import synth
from tests.helpers import GOLD, gold, rel_err

TOL = 1e-4


def _g6():
    g = gold("g6_inference_tiny.npz")
    bin2node = {str(k): int(v) for k, v in zip(g["bin_keys"], g["bin_vals"])}
    samples = [list(map(int, row[:n])) for row, n in zip(g["samples_pad"], g["sample_len"])]
    return g, bin2node, samples


def test_parse_file_matches_reference(tmp_path):
    g, bin2node, samples = _g6()
    path = os.path.join(tmp_path, "in.txt")
    with open(path, "w") as f:
        f.write(str(g["text"]))
    got = PR.parse_file(path, bin2node, [str(n) for n in g["names"]], int(g["res"]))
    assert got == samples and len(got) == int(g["n_samples"])
    with open(path, "w") as f:
        f.write("chr1-12\n")
    with pytest.raises(EOFError):
        PR.parse_file(path, bin2node, ["chr1"], 1000000)


def test_pairs_and_matrix_match_reference():
    g, _, _ = _g6()
    cr = np.asarray(synth.chrom_range(synth.LAYOUTS["tiny"]))
    for cid in (0, 2):
        pairs = PR.generate_pair_wise(cr, cid, 2)
        assert np.array_equal(pairs.numpy(), g[f"pairs_c{cid}"])
        for mode in ("adj", "table"):
            src = g[f"pairs_c{cid}"].copy()
            m = PR.proba2matrix(src, None, g[f"pair_proba_{mode}_c{cid}"])
            assert np.array_equal(src, g[f"pairs_c{cid}"])                       # caller's array untouched
            assert np.array_equal(m, g[f"pair_matrix_{mode}_c{cid}"])
    # weights and the inter-chromosomal layout (denoise_contact.py:41-45, :49-60), against the same arithmetic in numpy
    s = np.array([[3, 7], [4, 9], [3, 9]])
    p, w = np.array([0.2, 0.5, 0.9], dtype=np.float32), np.array([2.0, 0.5, 1.0], dtype=np.float32)
    m = PR.proba2matrix(s, w, p, intra=False)
    ref = np.zeros((2, 3), dtype=np.float32)
    ref[s[:, 0] - 3, s[:, 1] - 7] += np.maximum(p * w, p)
    assert np.array_equal(m, ref)
    assert PR.generate_pair_wise(cr, 1, 100).shape == (0, 2)                     # min_dis larger than the chromosome


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["adj", "table"])
def test_consumers_on_reference_pickle(mode, tmp_path):
    import Modules  # noqa: F401  (the pickle's GLOBALs are Modules.*)
    g, bin2node, samples = _g6()
    clf = torch.load(os.path.join(GOLD, f"ref_model2load_tiny_{mode}"), map_location="cuda", weights_only=False)
    # predict_multiway: file -> probabilities (and the savetxt output)
    path, out = os.path.join(tmp_path, "in.txt"), os.path.join(tmp_path, "out.txt")
    with open(path, "w") as f:
        f.write(str(g["text"]))
    got_samples, proba = PR.predict_multiway(clf, path, bin2node, [str(n) for n in g["names"]], int(g["res"]), out)
    assert got_samples == samples
    assert proba.shape == g[f"multiway_proba_{mode}"].shape
    assert rel_err(proba, g[f"multiway_proba_{mode}"]) < TOL
    assert rel_err(np.loadtxt(out).reshape(-1, 1), g[f"multiway_proba_{mode}"]) < TOL
    # chunking is part of the result: a chunk is padded to ITS widest row (fact 7) -- tiny chunks change short rows' logits
    lg_full = PR.predict(clf, samples)
    lg_small = PR.predict(clf, samples, batch_size=3)
    assert lg_full.shape == lg_small.shape and float(np.abs(lg_full - lg_small).max()) > 1e-6
    # denoise_contact's sweep on the device
    cr = np.asarray(synth.chrom_range(synth.LAYOUTS["tiny"]))
    for cid in (0, 2):
        pairs, p = PR.pairwise_probabilities(clf, cr, cid, 2, batch_rows=37)
        assert pairs.is_cuda and np.array_equal(pairs.cpu().numpy(), g[f"pairs_c{cid}"])
        assert rel_err(p.cpu().numpy(), g[f"pair_proba_{mode}_c{cid}"]) < TOL
        m = PR.proba2matrix(pairs, None, p)
        assert m.is_cuda and rel_err(m.cpu().numpy(), g[f"pair_matrix_{mode}_c{cid}"]) < TOL


@pytest.mark.gpu
def test_pairwise_sweep_full_chromosome_properties():
    """hg38 1 Mb, chr1 (249 bins, 30 628 pairs at min_dis 2), d = 64: symmetric matrix, zero band |i-j| < min_dis,
    probabilities in (0, 1), and independent of how the sweep is batched."""
    from tests.test_hip_model import hip_model
    num = synth.LAYOUTS["hg38_1mb"]
    clf, _ = hip_model(num, 64, "table", 12)
    cr = np.asarray(synth.chrom_range(num))
    pairs, p = PR.pairwise_probabilities(clf, cr, 0, 2)
    n = num[0]
    assert len(pairs) == (n - 2) * (n - 1) // 2
    assert float(p.min()) > 0.0 and float(p.max()) < 1.0
    _, p2 = PR.pairwise_probabilities(clf, cr, 0, 2, batch_rows=4099)
    assert torch.allclose(p, p2, rtol=0, atol=1e-6)
    m = PR.proba2matrix(pairs, None, p)
    assert m.shape == (n, n) and torch.equal(m, m.T)
    band = torch.triu(torch.ones(n, n, device=m.device), 0) * torch.tril(torch.ones(n, n, device=m.device), 1)
    assert float((m * band).abs().max()) == 0.0


@pytest.mark.gpu
def test_predict_cli_multiway_and_pairwise(tmp_path, monkeypatch):
    """python -m matcha_amd.predict {multiway, pairwise}: config.JSON + temp_dir/{model2load, bin2node.npy, chrom_range.npy},
    the files the reference's scripts read (predict_multiway.py:93-112, denoise_contact.py:91-99)."""
    import json
    import shutil
    import Modules  # noqa: F401
    g, bin2node, samples = _g6()
    temp = os.path.join(tmp_path, "Temp")
    os.makedirs(temp)
    shutil.copy(os.path.join(GOLD, "ref_model2load_tiny_table"), os.path.join(temp, "model2load"))
    np.save(os.path.join(temp, "bin2node.npy"), bin2node, allow_pickle=True)
    np.save(os.path.join(temp, "chrom_range.npy"), np.asarray(synth.chrom_range(synth.LAYOUTS["tiny"])))
    cfg = {"temp_dir": temp, "resolution": int(g["res"]), "chrom_list": [str(n) for n in g["names"]], "min_distance": 2}
    cpath = os.path.join(tmp_path, "config.JSON")
    with open(cpath, "w") as f:
        json.dump(cfg, f)
    inp, out = os.path.join(tmp_path, "in.txt"), os.path.join(tmp_path, "out.txt")
    with open(inp, "w") as f:
        f.write(str(g["text"]))
    PR.main(["multiway", "-i", inp, "-o", out, "--config", cpath])
    assert rel_err(np.loadtxt(out).reshape(-1, 1), g["multiway_proba_table"]) < TOL
    mout = os.path.join(tmp_path, "chr1.npy")
    PR.main(["pairwise", "--chrom", "0", "-o", mout, "--config", cpath])
    assert rel_err(np.load(mout), g["pair_matrix_table_c0"]) < TOL
