"""CPU-only checks: the C-ABI library loads and exports every symbol include/matcha_hip.h declares, the host-side
mirror keeps the reference's surface (state_dict keys, pickles), and the product path fails loudly -- instead of
falling back to a CPU implementation -- when there is no GPU tensor.  No compute calls are made here."""
import io
import os
import re

import numpy as np
import pytest
import torch

from matcha_amd import _lib, synth
from oracle import hypersagnn as O
from oracle import rng as R
from tests.helpers import GOLD, gold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "matcha_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(matcha_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"libmatcha_hip.so does not export {name}"
        assert name in _lib.SIGNATURES, f"ctypes binding missing for {name}"
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.matcha_abi_version() == _lib.ABI_VERSION == 7
    assert lib.matcha_device_count() >= 0


def test_graft_build_entry_point_runs():
    """The driver's "does it build" check: __graft_entry__.build() (make is a no-op when the library is up to date) must
    pass, including its ABI-version assertion against include/matcha_hip.h."""
    import re
    import __graft_entry__ as G
    G.build()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "matcha_hip.h")).read()
    assert int(re.search(r"#define MATCHA_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION


def test_ctypes_structs_match_header_field_order():
    src = open(os.path.join(ROOT, "include", "matcha_hip.h")).read()
    body = src[src.index("typedef struct matcha_tensors {"):src.index("} matcha_tensors;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"float\*\s+([a-z0-9_]+);", body)
    assert fields == _lib.TENSOR_FIELDS
    import ctypes as C
    assert C.sizeof(_lib.Shape) == 24 and C.sizeof(_lib.Tensors) == 8 * len(fields)


def test_workspace_query_and_argument_errors():
    lib = _lib.load()
    import ctypes as C
    shp = _lib.Shape(64, 24, 3067, 23, 0, 250)
    small, big = lib.matcha_workspace_bytes(C.byref(shp), 384, 5), lib.matcha_workspace_bytes(C.byref(shp), 768, 5)
    assert 0 < small < big
    bad = _lib.Shape(20, 24, 10, 0, 0, 0)          # d=20: unsupported
    assert lib.matcha_workspace_bytes(C.byref(bad), 4, 3) == 0
    assert b"embed_dim" in lib.matcha_last_error()
    assert lib.matcha_workspace_bytes(C.byref(shp), 4, 9) == 0      # L > 8
    assert lib.matcha_hashset_bytes(1000) >= 4 * 2000


def _build(mode):
    import Modules as M
    num, d = synth.LAYOUTS["tiny"], 16
    attr = O.attribute_table(num)
    N = int(np.sum(num))
    if mode == "table":
        ne = M.Wrap_Embedding(N + 1, d, padding_idx=0)
    else:
        intra, inter = synth.make_adjacency(np.random.default_rng(1), num)
        ne = M.MultipleEmbedding(O.corrcoef_features(intra, synth.chrom_range(num)), d, False, torch.as_tensor(np.cumsum(num)),
                                 synth.chrom_range(num), inter.copy())
    return M.Classifier(n_head=8, d_model=d, d_k=d, d_v=d, node_embedding=ne, diag_mask=True, bottle_neck=d, attribute_dict=attr)


@pytest.mark.parametrize("mode", ["adj", "table"])
def test_state_dict_surface_matches_reference(mode):
    ref = gold(f"g1_tiny_{mode}_refinit.npz")
    clf = _build(mode)
    sd = clf.state_dict()
    assert list(sd.keys()) == list(ref.files)                  # names AND registration order
    for k in ref.files:
        assert tuple(sd[k].shape) == ref[k].shape, k
    frozen = [n for n, p in clf.named_parameters() if not p.requires_grad]
    assert frozen == ["attribute_dict_embedding.weight"]        # Modules.py:247
    clf.load_state_dict({k: torch.from_numpy(ref[k]) for k in ref.files})      # reference checkpoint keys load


def test_inter_zscore_matches_reference_preprocessing():
    g = gold("g5_tiny_preproc.npz")
    import Modules as M
    num = synth.LAYOUTS["tiny"]
    intra, inter = synth.make_adjacency(np.random.default_rng(11 + 1000), num)
    ne = M.MultipleEmbedding(O.corrcoef_features(intra, synth.chrom_range(num)), 16, False, torch.as_tensor(np.cumsum(num)),
                             synth.chrom_range(num), inter.copy())
    np.testing.assert_allclose(ne.inter_initial.embedding.cpu().numpy(), g["inter_z"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("mode", ["adj", "table"])
def test_reference_pickle_unpickles_into_our_classes(mode):
    import Modules  # noqa: F401
    clf = torch.load(os.path.join(GOLD, f"ref_model2load_tiny_{mode}"), map_location="cpu", weights_only=False)
    assert type(clf).__module__ == "Modules" and type(clf).__name__ == "Classifier"
    from matcha_amd.Modules import Classifier
    assert isinstance(clf, Classifier)
    buf = io.BytesIO()
    torch.save(clf, buf)                                        # and pickles back out under the same GLOBAL names
    assert b"Modules" in buf.getvalue() and b"matcha_amd" not in buf.getvalue()


def test_no_cpu_fallback():
    """A CPU model must raise, not silently compute on the host."""
    clf = _build("table").to("cpu")
    with pytest.raises(_lib.MatchaHipError):
        clf(torch.tensor([[1, 2, 3]]))
    with pytest.raises(_lib.MatchaHipError):
        clf.get_node_embeddings(torch.tensor([[1], [2]]))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "matcha_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "/root/reference" not in txt or f == "Modules.py", f
    assert not re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(ROOT, "Modules.py")).read(), flags=re.M)


def test_data_generator_contract():
    """Modules.py:620-681: > num_batch*batch rows per size after duplication, exactly that many per call, wrap."""
    import Modules as M
    rng = np.random.default_rng(0)
    e2, e3 = synth.make_edges(rng, 64, 2, 50), synth.make_edges(rng, 64, 3, 70)
    edges = [r for r in e2] + [r for r in e3]
    w = np.arange(len(edges), dtype=np.float32)
    np.random.seed(0)
    gen = M.DataGenerator(edges, w, batch_size=8, num_batch_per_iter=10, min_size=2, max_size=3)
    assert len(gen.edges[2]) > 80 and len(gen.edges[3]) > 80
    for _ in range(5):
        e, ww = gen.next_iter()
        assert e.shape == (160, 3) and ww.shape == (160,)
        k = (e != 0).sum(1)
        assert (k[:80] == 2).all() and (k[80:] == 3).all()
        # weights stay attached to their edges
        lut = {tuple(r.tolist()): i for i, r in enumerate(edges)}
        for row, wi in zip(e, ww):
            assert lut[tuple(row[row != 0].tolist())] == int(wi)


def test_g10_data_generator_and_sync_shuffle_bit_exact_vs_reference():
    """Round 6 fixture g10 (make_golden.py::g10_host_streams): the reference's DataGenerator.__init__ / next_iter (Modules.py:620-681)
    under np.random.seed and sync_shuffle (utils.py:142-149) under torch.manual_seed, uniform-k inputs -- integer work, so the host
    mirror must return the SAME rows in the SAME order, bit for bit (the duplication rule, the permutation calls and their order,
    the wrap-around with its reshuffle)."""
    import Modules as M
    from matcha_amd import utils as U
    g = gold_g10()
    rng = np.random.default_rng(1010)
    for k, m in ((2, 700), (3, 1300)):
        edges = synth.make_edges(rng, 300, k, m)
        weight = rng.uniform(0.1, 3.0, size=m).astype(np.float32)
        assert np.array_equal(edges, g[f"edges_k{k}"]) and np.array_equal(weight, g[f"weight_k{k}"])   # (the generator's inputs)
        for bs, nb in ((96, 10), (250, 3), (96, 2)):
            np.random.seed(77 + k)
            dg = M.DataGenerator(edges.copy(), weight.copy(), bs, nb, min_size=k, max_size=k, flag=(bs == 250))
            tag = f"k{k}_b{bs}_n{nb}"
            assert len(dg.edges[k]) == int(g[f"dg_len_{tag}"])
            for it in range(5):
                e, w = dg.next_iter()
                assert np.array_equal(e, g[f"dg_e_{tag}_{it}"]), (tag, it)
                assert np.array_equal(w, g[f"dg_w_{tag}_{it}"]), (tag, it)
    for n in (1, 7, 384, 1000):
        a = np.arange(n, dtype=np.int64) * 3 + 1
        b = g[f"ss_in_b_{n}"]
        torch.manual_seed(5 + n)
        sa, sb = U.sync_shuffle([torch.from_numpy(a), torch.from_numpy(b)])
        assert np.array_equal(sa.numpy(), g[f"ss_a_{n}"]) and np.array_equal(sb.numpy(), g[f"ss_b_{n}"])
        torch.manual_seed(5 + n)
        sa10, = U.sync_shuffle([torch.from_numpy(a)], min(n, 10))
        assert np.array_equal(sa10.numpy(), g[f"ss_a10_{n}"])


def gold_g10():
    return np.load(os.path.join(ROOT, "tests", "golden", "g10_host_streams.npz"), allow_pickle=False)


def test_rng_spec_is_stable():
    """Known-answer vectors of the counter RNG (oracle/rng.py == matcha_amd/csrc/common.hpp)."""
    assert int(R.lowbias32(0)) == 0 and int(R.lowbias32(1)) == 1753845952 and int(R.lowbias32(0xFFFFFFFF)) == 1734902346
    key = R.make_key(123456789012345, R.STREAM_DROP_FC1)
    vals = R.rand_u32(key, np.arange(3), np.arange(3))
    assert vals.dtype == np.uint32 and len(set(vals.tolist())) == 3
    m = R.dropout_mask(7, R.STREAM_DROP_PFF, 0.4, 2000, 16)
    assert set(np.unique(m).tolist()) == {0.0, np.float32(1.0) / (np.float32(1.0) - np.float32(0.4))}
    assert abs(float((m == 0).mean()) - 0.4) < 0.02


def test_attribute_structure_detection_is_exact():
    """matcha_amd.Modules._attr_structure: get_attributes' table (main.py:497-512) is recognised with its bounds and scale; anything
    else -- a wrong coordinate by one ulp, two ones in a row, chromosomes out of order, a non-zero padding row -- is refused, so the
    kernels only ever rebuild rows that equal the table's bit for bit."""
    import torch
    from matcha_amd import synth
    from matcha_amd.Modules import _attr_structure
    from oracle import hypersagnn as O
    for name in ("tiny", "c1", "c23", "hg38_1mb"):
        num = synth.LAYOUTS[name]
        bounds, scale = _attr_structure(torch.from_numpy(O.attribute_table(num)))
        assert bounds == [0] + list(np.cumsum(num)) and scale == float(num[0])
    base = O.attribute_table(synth.LAYOUTS["c23"])
    for edit in ("ulp", "two_ones", "pad_row", "order", "dense"):
        t = base.copy()
        if edit == "ulp":
            t[9, -1] = np.nextafter(t[9, -1], np.float32(2.0))
        elif edit == "two_ones":
            t[4, 3] = 1.0
        elif edit == "pad_row":
            t[0, 0] = 1.0
        elif edit == "order":
            t[[1, 140]] = t[[140, 1]]
        else:
            t = np.random.default_rng(0).normal(size=t.shape).astype(np.float32)
        assert _attr_structure(torch.from_numpy(t)) is None, edit
