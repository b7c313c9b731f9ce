"""k-mer generation.  G7 = rows produced by the reference's own build_dict (generate_kmers.py, exec'd from
its AST by tests/golden/make_golden.py::g7_kmers), sorted lexicographically."""
import numpy as np
import pytest

from oracle import kmers as OK
from tests.helpers import gold

CASES = [(d, c, k) for d, c in ((0, 2), (2, 1), (1, 3)) for k in (2, 3, 4, 5)]


def _clusters(g):
    out, o = [], 0
    for n in g["cl_len"]:
        out.append(g["cl_flat"][o:o + int(n)])
        o += int(n)
    return out


@pytest.mark.parametrize("min_dis,cutoff,k", CASES)
def test_oracle_matches_reference_build_dict(min_dis, cutoff, k):
    g = gold("g7_kmers.npz")
    rows, freq = OK.generate_kmers(_clusters(g), k, min_dis, int(g["max_size"]), cutoff)
    assert np.array_equal(rows, g[f"kmers_d{min_dis}_c{cutoff}_k{k}"]) and np.array_equal(freq, g[f"freq_d{min_dis}_c{cutoff}_k{k}"])


@pytest.mark.gpu
@pytest.mark.parametrize("min_dis,cutoff,k", CASES)
def test_hip_kmers_match_reference(min_dis, cutoff, k):
    from matcha_amd import kmers as KM
    g = gold("g7_kmers.npz")
    cl = _clusters(g)
    want_r, want_f = g[f"kmers_d{min_dis}_c{cutoff}_k{k}"], g[f"freq_d{min_dis}_c{cutoff}_k{k}"]
    rows, freq = KM.generate_kmers(cl, k, min_dis, int(g["max_size"]), cutoff, n_nodes=int(g["n_nodes"]))
    assert rows.dtype == np.int64 and rows.shape == want_r.shape
    assert np.array_equal(rows, want_r) and np.array_equal(freq, want_f)
    # 128-bit key path: pretend the node-id space is huge (k * bits > 63); same answer
    rows2, freq2 = KM.generate_kmers(cl, k, min_dis, int(g["max_size"]), cutoff, n_nodes=(1 << 24) - 1)
    assert np.array_equal(rows2, want_r) and np.array_equal(freq2, want_f)


@pytest.mark.gpu
def test_hip_kmers_split_launches_and_random_clusters(monkeypatch):
    """Bigger random input against the oracle; then the same with the per-launch candidate limit forced low, which exercises
    the multi-launch merge (a k-mer's occurrences are spread over launches)."""
    from matcha_amd import kmers as KM
    rng = np.random.default_rng(5)
    cl = []
    for _ in range(1500):
        n = int(rng.integers(2, 15))
        centre = int(rng.integers(10, 290))
        c = np.unique(np.clip(centre + rng.integers(-12, 13, size=2 * n), 1, 300))[:n]
        if len(c) >= 2:
            cl.append(c.astype(np.int64))
    for k, min_dis, cutoff in ((3, 1, 2), (4, 0, 3), (6, 0, 2)):
        want_r, want_f = OK.generate_kmers(cl, k, min_dis, 12, cutoff)
        rows, freq = KM.generate_kmers(cl, k, min_dis, 12, cutoff, n_nodes=300)
        assert len(want_r) > 0 and np.array_equal(rows, want_r) and np.array_equal(freq, want_f)
        monkeypatch.setattr(KM, "MAX_COMBOS", 5000)
        rows, freq = KM.generate_kmers(cl, k, min_dis, 12, cutoff, n_nodes=300)
        monkeypatch.undo()
        assert np.array_equal(rows, want_r) and np.array_equal(freq, want_f)
    # nothing qualifies: empty result with the right shapes
    rows, freq = KM.generate_kmers(cl, 5, 1000, 12, 1, n_nodes=300)
    assert rows.shape == (0, 5) and freq.shape == (0,)


@pytest.mark.gpu
def test_kmers_cli_writes_the_files_train_reads(tmp_path):
    """python -m matcha_amd.kmers: config.JSON + temp_dir/{edge_list,chrom_range}.npy -> all_<k>_counter.npy / _freq_counter.npy
    (generate_kmers.py:133-139), the inputs of matcha_amd.train.load_kmers."""
    import json
    import os
    from matcha_amd import kmers as KM
    g = gold("g7_kmers.npz")
    cl = _clusters(g)
    temp = os.path.join(tmp_path, "Temp")
    os.makedirs(temp)
    obj = np.empty(len(cl), dtype=object)
    for i, c in enumerate(cl):
        obj[i] = list(map(int, c))
    np.save(os.path.join(temp, "edge_list.npy"), obj, allow_pickle=True)
    np.save(os.path.join(temp, "chrom_range.npy"), np.array([[1, 31], [31, int(g["n_nodes"]) + 1]]))
    cfg = {"temp_dir": temp, "max_cluster_size": int(g["max_size"]), "k-mer_size": [2, 3, 5], "min_distance": 0, "min_freq_cutoff": 2}
    path = os.path.join(tmp_path, "config.JSON")
    with open(path, "w") as f:
        json.dump(cfg, f)
    KM.main(["--config", path])
    for k in (2, 3, 5):
        rows = np.load(os.path.join(temp, "all_%d_counter.npy" % k))
        freq = np.load(os.path.join(temp, "all_%d_freq_counter.npy" % k))
        assert np.array_equal(rows, g[f"kmers_d0_c2_k{k}"]) and np.array_equal(freq, g[f"freq_d0_c2_k{k}"])
