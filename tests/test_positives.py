"""positive-weight preprocessing (main.py:551-566, :594-597, :646-660).  G8 = scikit-learn's own
QuantileTransformer outputs (tests/golden/make_golden.py::g8_positives); the oracle restates its algorithm, the HIP path
(csrc/quantile.hip) must reproduce both bit for bit -- the transform is float64 arithmetic rounded once to float32."""
import numpy as np
import pytest

from oracle import positives as OP
from tests.helpers import gold

COLUMNS = ["counts_u", "counts_heavy", "short", "real", "constant", "single", "pair", "big_counts", "big_real"]


@pytest.mark.parametrize("name", COLUMNS)
def test_oracle_matches_sklearn(name):
    g = gold("g8_positives.npz")
    q, r = OP.fit_quantiles(g[f"{name}_freq"])
    assert np.array_equal(q, g[f"{name}_quantiles"])
    assert np.array_equal(OP.transform_uniform(g[f"{name}_freq"], q, r), g[f"{name}_weight"])


@pytest.mark.parametrize("cutoff", [0.6, 0.4])
def test_oracle_selection_and_weights(cutoff):
    g = gold("g8_positives.npz")
    rows, w = OP.select_positives([g["sel_data_k2"], g["sel_data_k3"]], [g["counts_u_freq"], g["counts_heavy_freq"]], cutoff)
    rows = np.concatenate([np.pad(r, ((0, 0), (0, 3 - r.shape[1]))) for r in rows])
    assert np.array_equal(rows, g[f"sel_rows_c{cutoff}"]) and np.array_equal(w, g[f"sel_weight_c{cutoff}"])
    assert np.array_equal(OP.normalise_weights(w, 3), g[f"sel_norm_c{cutoff}"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", COLUMNS)
def test_hip_quantile_matches_sklearn(name):
    from matcha_amd import positives as P
    g = gold("g8_positives.npz")
    w, q = P.quantile_uniform(g[f"{name}_freq"], return_quantiles=True)
    assert np.array_equal(q.cpu().numpy(), g[f"{name}_quantiles"])
    assert np.array_equal(w.cpu().numpy(), g[f"{name}_weight"])                 # bit-exact


@pytest.mark.gpu
@pytest.mark.parametrize("cutoff", [0.6, 0.4])
def test_hip_selection_and_weights(cutoff):
    import torch
    from matcha_amd import positives as P
    g = gold("g8_positives.npz")
    rows, w = P.select_positives([g["sel_data_k2"], g["sel_data_k3"]], [g["counts_u_freq"], g["counts_heavy_freq"]], cutoff)
    assert rows.dtype == torch.int64 and np.array_equal(rows.cpu().numpy(), g[f"sel_rows_c{cutoff}"])
    assert np.array_equal(w.cpu().numpy(), g[f"sel_weight_c{cutoff}"])
    norm = P.normalise_weights(w, 3).cpu().numpy()
    assert np.allclose(norm, g[f"sel_norm_c{cutoff}"], rtol=1e-6, atol=0)       # float32 mean: summation order differs
    assert abs(float(norm.mean()) - 3.0) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,nq,kind", [(1 << 20, 1000, "counts"), (3_000_001, 1000, "real"), (5000, 4096, "counts"), (999, 1000, "real"),
                                       (1000, 1000, "counts"), (1001, 1000, "real"), (100_000, 7, "counts")])
def test_hip_quantile_matches_oracle_at_size(n, nq, kind):
    """Sizes scikit-learn would subsample: against the restatement (every row fitted), still bit for bit; plus the
    size-independent properties of the transform: monotone in the frequency, ties share a value, range [0, 1]."""
    from matcha_amd import positives as P
    rng = np.random.default_rng(n + nq)
    col = (np.floor(rng.pareto(1.1, size=n)) + 2 if kind == "counts" else rng.gamma(2.0, 1.0, size=n)).astype("float32")
    w, q = P.quantile_uniform(col, n_quantiles=nq, return_quantiles=True)
    w, q = w.cpu().numpy(), q.cpu().numpy()
    oq, r = OP.fit_quantiles(col, nq)
    assert np.array_equal(q, oq)
    assert np.array_equal(w, OP.transform_uniform(col, oq, r))
    order = np.argsort(col, kind="stable")
    assert (np.diff(w[order]) >= 0).all() and w.min() == 0.0 and w.max() == 1.0
    same = np.diff(col[order]) == 0
    assert (np.diff(w[order])[same] == 0).all()


@pytest.mark.gpu
def test_hip_quantile_argument_errors():
    import ctypes as C
    import torch
    from matcha_amd import _lib, positives as P
    lib = _lib.load()
    assert lib.matcha_quantile_workspace_bytes(0) == 0
    x = torch.ones(8, device="cuda")
    with pytest.raises(_lib.MatchaHipError):
        P.quantile_uniform(x, n_quantiles=5000)
    assert P.quantile_uniform(np.zeros(0, dtype="float32")).numel() == 0
    ws = torch.empty(16, dtype=torch.uint8, device="cuda")
    assert lib.matcha_quantile_uniform(_lib.ptr(x), 8, 1000, _lib.ptr(x.clone()), None, _lib.ptr(ws), 16, C.c_void_p(0)) != 0   # workspace too small
