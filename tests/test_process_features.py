"""node dictionaries, cluster parsing, cooler pixels -> adjacency, correlation / z-score features.
G9 = outputs of the reference's own process.py functions (+ main.py:571-575, Modules.py:146-152) on a synthetic genome,
tests/golden/make_golden.py::g9_process."""
import numpy as np
import pytest

from matcha_amd import process as PR
from oracle import hypersagnn as O
from oracle import process as OPR
from tests.helpers import gold

CHROMS = ["chr1", "chr2", "chrX"]
RES = 1000000


def _dicts(tmp_path):
    g = gold("g9_process.npz")
    p = tmp_path / "sizes.txt"
    p.write_text(str(g["sizes_text"]))
    return g, PR.build_node_dict(str(p), CHROMS, RES, str(tmp_path / "temp"))


def test_build_node_dict_matches_reference(tmp_path):
    g, (bin2node, node2bin, node2chrom, chrom_range) = _dicts(tmp_path)
    assert np.array_equal(chrom_range, g["chrom_range"])
    assert list(bin2node.keys()) == list(g["bin2node_keys"]) and list(bin2node.values()) == list(g["bin2node_vals"])
    N = int(chrom_range.max()) - 1
    assert [node2chrom[i] for i in range(1, N + 1)] == list(g["node2chrom"])
    assert [node2bin[i] for i in range(1, N + 1)] == list(g["node2bin"])
    saved = np.load(tmp_path / "temp" / "bin2node.npy", allow_pickle=True).item()                  # the files the other scripts load
    assert saved == bin2node and np.array_equal(np.load(tmp_path / "temp" / "chrom_range.npy"), g["chrom_range"])
    with pytest.raises(ValueError):
        PR.build_node_dict(str(tmp_path / "sizes.txt"), ["chr1", "chr9"], RES)


def test_parse_clusters_matches_reference(tmp_path):
    g, (bin2node, _, _, _) = _dicts(tmp_path)
    p = tmp_path / "x.cluster"
    p.write_text(str(g["cluster_text"]))
    got = PR.parse_clusters(str(p), bin2node, CHROMS, RES, int(g["max_cluster_size"]), str(tmp_path / "temp"))
    assert [len(c) for c in got] == list(g["edge_len"]) and [v for c in got for v in c] == list(g["edge_flat"])
    saved = np.load(tmp_path / "temp" / "edge_list.npy", allow_pickle=True)
    assert len(saved) == len(got) and list(saved[0]) == got[0]
    bad = tmp_path / "bad.cluster"
    bad.write_text("c1\tchr1:5\tchr2\n")
    with pytest.raises(EOFError):
        PR.parse_clusters(str(bad), bin2node, CHROMS, RES, 6)


@pytest.mark.parametrize("key", ["balanced", "count"])
def test_oracle_pixels_to_adj_matches_reference(tmp_path, key):
    g, (bin2node, _, node2chrom, chrom_range) = _dicts(tmp_path)
    i2n = PR.cool_index2node(g["bins_chrom"], g["bins_start"], g["chrom_names"], CHROMS, bin2node)
    assert (i2n == 0).sum() == 3                                                                  # the chrY bins
    intra, inter = OPR.pixels_to_adj(g["bin1"], g["bin2"], g[key], i2n, node2chrom, int(chrom_range.max()) - 1)
    assert np.array_equal(intra, g[f"intra_{key}"]) and np.array_equal(inter, g[f"inter_{key}"])


def test_oracle_features_match_reference():
    g = gold("g9_process.npz")
    feats = O.corrcoef_features(g["intra_balanced"].astype("float32"), g["chrom_range"])
    for ci, f in enumerate(feats):
        assert np.array_equal(f, g[f"corr_{ci}"])
    assert np.array_equal(O.zscore_inter(g["inter_balanced"].astype("float32")), g["inter_zscore"])


# ---- the device path -------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("key", ["balanced", "count"])
def test_hip_pixels_to_adj_matches_reference(tmp_path, key):
    g, (bin2node, _, node2chrom, chrom_range) = _dicts(tmp_path)
    N = int(chrom_range.max()) - 1
    i2n = PR.cool_index2node(g["bins_chrom"], g["bins_start"], g["chrom_names"], CHROMS, bin2node)
    intra, inter = PR.pixels_to_adj(g["bin1"], g["bin2"], g[key], i2n, node2chrom, N)
    assert np.array_equal(intra.cpu().numpy(), g[f"intra_{key}"]) and np.array_equal(inter.cpu().numpy(), g[f"inter_{key}"])   # bit-exact
    # streamed in three chunks into the same matrices
    out = None
    for s in np.array_split(np.arange(len(g["bin1"])), 3):
        out = PR.pixels_to_adj(g["bin1"][s], g["bin2"][s], g[key][s], i2n, node2chrom, N, out=out)
    assert np.array_equal(out[0].cpu().numpy(), g[f"intra_{key}"]) and np.array_equal(out[1].cpu().numpy(), g[f"inter_{key}"])
    PR.save_adj(str(tmp_path / "temp"), intra, inter)
    assert np.array_equal(np.load(tmp_path / "temp" / "intra_adj.npy"), g[f"intra_{key}"])


@pytest.mark.gpu
def test_hip_pixels_to_adj_large_random():
    """hg38-1Mb-sized node set, 2 M integer-count pixels with repeats: sums of integers are exact in float64, so the atomics
    must reproduce the sequential loop bit for bit; symmetric; intra and inter supports are disjoint."""
    rng = np.random.default_rng(4)
    num = [250, 244, 200, 192, 183]
    N = sum(num)
    n2c = np.concatenate([[-1]] + [np.full(n, c) for c, n in enumerate(num)]).astype(np.int32)
    n_index = N + 40
    i2n = np.zeros(n_index, dtype=np.int32)
    i2n[rng.permutation(n_index)[:N]] = np.arange(1, N + 1)
    P = 2_000_000
    b1, b2 = rng.integers(0, n_index, size=P), rng.integers(0, n_index, size=P)
    cnt = rng.integers(1, 100, size=P).astype(np.float64)
    cnt[rng.random(P) < 0.01] = np.nan
    intra, inter = PR.pixels_to_adj(b1, b2, cnt, i2n, n2c, N)
    intra, inter = intra.cpu().numpy(), inter.cpu().numpy()
    ok = (i2n[b1] > 0) & (i2n[b2] > 0) & ~np.isnan(cnt)
    r, s, c = i2n[b1][ok] - 1, i2n[b2][ok] - 1, cnt[ok]
    same = n2c[r + 1] == n2c[s + 1]
    ref = [np.zeros((N, N)), np.zeros((N, N))]
    for m, sel in ((ref[0], same), (ref[1], ~same)):
        np.add.at(m, (r[sel], s[sel]), c[sel])
        np.add.at(m, (s[sel], r[sel]), c[sel])
    assert np.array_equal(intra, ref[0]) and np.array_equal(inter, ref[1])
    assert np.array_equal(intra, intra.T) and not ((intra != 0) & (inter != 0)).any()


@pytest.mark.gpu
def test_hip_features_match_reference():
    import torch
    from matcha_amd import features as F
    g = gold("g9_process.npz")
    feats = F.corrcoef_features(g["intra_balanced"].astype("float32"), g["chrom_range"])
    for ci, f in enumerate(feats):
        assert f.dtype == torch.float32 and np.allclose(f.cpu().numpy(), g[f"corr_{ci}"], rtol=0, atol=2e-7)   # float64 arithmetic, float32 rounding
    z = F.zscore_rows_(torch.from_numpy(g["inter_balanced"].astype("float32")).cuda()).cpu().numpy()
    assert np.allclose(z, g["inter_zscore"], rtol=1e-5, atol=1e-6)      # the reference accumulates in float32, the kernel in float64
    assert ((z != 0) == (g["inter_zscore"] != 0)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [[2, 3, 63, 64, 65], [300, 129, 1000]])
def test_hip_corrcoef_blocks_at_size(sizes):
    """Blocks that do not fill the 64 x 64 tiles, contact-map-like rows (distance decay), dead bins (all-zero rows -> the
    0 / 0 of np.corrcoef -> 0), against np.corrcoef in float64; symmetric, unit diagonal on live rows."""
    from matcha_amd import features as F
    rng = np.random.default_rng(sum(sizes))
    N = sum(sizes)
    adj = np.zeros((N, N), dtype=np.float32)
    cr, lo = [], 1
    for n in sizes:
        i = np.arange(n)
        a = rng.gamma(2.0, 1.0, size=(n, n)) / (np.abs(i[:, None] - i[None, :]) + 1.0)
        a = ((a + a.T) / 2).astype(np.float32)
        dead = rng.random(n) < 0.05
        a[dead, :] = 0
        a[:, dead] = 0
        adj[lo - 1:lo - 1 + n, lo - 1:lo - 1 + n] = a
        cr.append([lo, lo + n])
        lo += n
    adj += (rng.random((N, N)) < 0.01).astype(np.float32) * (adj == 0)          # clutter outside the blocks must not be read
    feats = F.corrcoef_features(adj, cr)
    ref = O.corrcoef_features(adj, np.asarray(cr))
    for f, r in zip(feats, ref):
        f = f.cpu().numpy()
        assert f.shape == r.shape and np.abs(f - r).max() <= 2e-6
        assert np.array_equal(f, f.T)
        live = np.diag(r) != 0
        assert np.allclose(np.diag(f)[live], 1.0, atol=1e-6) and (np.diag(f)[~live] == 0).all()


@pytest.mark.gpu
def test_hip_zscore_rows_edge_cases():
    import torch
    from matcha_amd import features as F
    rng = np.random.default_rng(12)
    m = rng.gamma(2.0, 0.05, size=(200, 3067)).astype(np.float32) * (rng.random((200, 3067)) < 0.6)
    m[0, :] = 0                                  # no positive entry
    m[1, :] = 0
    m[1, 7] = 3.0                                # a single positive entry: std 0 -> NaN -> 0
    m[2, :] = -1.0                               # negatives are left alone
    m[2, 5] = 2.0
    m[2, 9] = 4.0
    m[3, 11] = np.nan                            # a NaN input becomes 0
    ref = O.zscore_inter(m)
    z = F.zscore_rows_(torch.from_numpy(m.copy()).cuda()).cpu().numpy()
    assert np.allclose(z, ref, rtol=2e-5, atol=2e-6)
    assert (z[0] == 0).all() and (z[1] == 0).all() and z[2, 0] == -1.0 and z[2, 5] == -1.0 and z[2, 9] == 1.0 and z[3, 11] == 0
    pos = m > 0
    rows = pos.sum(1) > 1
    assert np.allclose(np.where(pos, z, 0).sum(1)[rows] / pos.sum(1)[rows], 0, atol=1e-5)         # zero mean, unit variance per row
    assert np.allclose((np.where(pos, z, 0) ** 2).sum(1)[rows] / pos.sum(1)[rows], 1, atol=1e-4)
    with pytest.raises(ValueError):
        F.zscore_rows_(torch.zeros(4, 4))
