"""Deterministic synthetic inputs for the hyperedge-classifier path (numpy only).

Used by bench.py (there is no network for real 4DN data), by the parity tests and by
tests/golden/make_golden.py, so that golden fixtures only need to store *outputs*: inputs and
weights are regenerated from a seed with ``np.random.default_rng`` (bit-stable across numpy
versions).  Layouts follow SURVEY.md §8(d2) / BASELINE.json ``configs``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List

import numpy as np

# hg38, 1 Mb bins: ceil(size/res)+1 bins per chromosome (reference process.py:23-31), chr1..22 + chrX
HG38_1MB = [250, 244, 200, 192, 183, 172, 161, 147, 140, 135, 137, 135, 116, 109, 103, 92, 85, 82,
            60, 66, 48, 52, 158]
# hg38 chromosome sizes (Mb, rounded up to 0.1) -> 100 kb bins
_HG38_MB = [248.96, 242.20, 198.30, 190.22, 181.54, 170.81, 159.35, 145.14, 138.40, 133.80, 135.09,
            133.28, 114.37, 107.05, 101.99, 90.34, 83.26, 80.38, 58.62, 64.45, 46.71, 50.82, 156.05]
HG38_100KB = [int(np.ceil(mb * 10)) + 1 for mb in _HG38_MB]

LAYOUTS = {
    "tiny": [16, 16, 16, 16],          # golden-fixture size
    "c1": [128, 128, 128, 128],        # BASELINE.json configs[0]: 512 bins
    "c23": [12, 11, 10, 9, 8, 8, 8, 7, 7, 7, 6, 6, 6, 6, 5, 5, 5, 5, 4, 4, 4, 4, 3],   # 23 chromosomes (n_attr = 24 like hg38), 150 bins: full d = 64 goldens
    "hg38_1mb": HG38_1MB,              # configs[1], configs[2]: N = 3067
    "hg38_100kb": HG38_100KB,          # configs[3]: N ~ 30.4 k
    "c5": [50000] * 20,                # configs[4]: 1 M nodes (20 synthetic chromosomes of 50 000 bins; table front end only)
}


def chrom_range(num: List[int]) -> np.ndarray:
    """[C,2] int64, 1-based [start,end) node ids per chromosome (reference temp_dir/chrom_range.npy)."""
    b = np.concatenate([[0], np.cumsum(num)])
    return np.stack([b[:-1] + 1, b[1:] + 1], axis=1).astype(np.int64)


def bounds(num: List[int]) -> List[int]:
    return [0] + [int(v) for v in np.cumsum(num)]


def node2chrom(num: List[int]) -> np.ndarray:
    """int32 [N+1]; entry 0 (pad) = -1."""
    out = np.full(int(np.sum(num)) + 1, -1, dtype=np.int32)
    b = bounds(num)
    for c in range(len(num)):
        out[b[c] + 1:b[c + 1] + 1] = c
    return out


def make_edges(rng: np.random.Generator, n_nodes: int, k: int, m: int) -> np.ndarray:
    """m distinct hyperedges of k distinct nodes (ids 1..n_nodes), each row ascending; int64 [m,k]."""
    seen, rows = set(), []
    while len(rows) < m:
        cand = rng.integers(1, n_nodes + 1, size=(2 * (m - len(rows)) + 16, k))
        cand.sort(axis=1)
        ok = (np.diff(cand, axis=1) > 0).all(axis=1)
        for r in cand[ok]:
            t = tuple(int(v) for v in r)
            if t not in seen:
                seen.add(t)
                rows.append(t)
                if len(rows) == m:
                    break
    return np.asarray(rows, dtype=np.int64)


def make_edges_fast(rng: np.random.Generator, n_nodes: int, k: int, m: int) -> np.ndarray:
    """Vectorised variant for large m (bench sizes): rows ascending and duplicate-free within a row;
    duplicate rows are removed with np.unique and topped up."""
    out = np.zeros((0, k), dtype=np.int64)
    while len(out) < m:
        cand = rng.integers(1, n_nodes + 1, size=(int(1.3 * (m - len(out))) + 64, k))
        cand.sort(axis=1)
        cand = cand[(np.diff(cand, axis=1) > 0).all(axis=1)]
        out = np.unique(np.concatenate([out, cand]), axis=0)
    sel = rng.permutation(len(out))[:m]
    return out[np.sort(sel)]


def make_freq(rng: np.random.Generator, m: int) -> np.ndarray:
    """Occurrence counts as generate_kmers.py would store them: integers U[2,50)."""
    return rng.integers(2, 50, size=m).astype(np.float32)


def make_adjacency(rng: np.random.Generator, num: List[int]):
    """(intra_adj, inter_adj) float32 [N,N] symmetric contact maps:
    intra = Gamma(2,1)/(|i-j|+1) inside a chromosome, inter = Gamma(2,0.05) between chromosomes."""
    n = int(np.sum(num))
    b = bounds(num)
    g = rng.gamma(2.0, 1.0, size=(n, n)).astype(np.float32)
    g = (g + g.T) * 0.5
    idx = np.arange(n)
    intra = np.zeros((n, n), dtype=np.float32)
    inter = (g * 0.05).astype(np.float32)
    for c in range(len(num)):
        lo, hi = b[c], b[c + 1]
        dist = np.abs(idx[lo:hi, None] - idx[None, lo:hi]) + 1
        intra[lo:hi, lo:hi] = g[lo:hi, lo:hi] / dist
        inter[lo:hi, lo:hi] = 0.0
    return intra, inter


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _ln(rng, d):
    return (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32), (0.1 * rng.standard_normal(d)).astype(np.float32)


def make_state_dict(rng: np.random.Generator, num: List[int], d: int, mode: str,
                    attr_table: np.ndarray, n_head: int = 8) -> "OrderedDict[str, np.ndarray]":
    """Weights for every key of the reference ``Classifier.state_dict()`` (same names, shapes and
    registration order; scales close to the reference's initialisers, LayerNorm affine made
    non-trivial so that parity tests exercise it).  mode: 'adj' (MultipleEmbedding) | 'table'."""
    C = len(num)
    N = int(np.sum(num))
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def linear(prefix, out_f, in_f, bias=True):
        sd[prefix + ".weight"] = _uniform(rng, (out_f, in_f), 1.0 / np.sqrt(in_f))
        if bias:
            sd[prefix + ".bias"] = _uniform(rng, (out_f,), 1.0 / np.sqrt(in_f))

    def conv(prefix, out_f, in_f):
        sd[prefix + ".weight"] = _uniform(rng, (out_f, in_f, 1), 1.0 / np.sqrt(in_f))
        sd[prefix + ".bias"] = _uniform(rng, (out_f,), 1.0 / np.sqrt(in_f))

    def lnorm(prefix, n):
        sd[prefix + ".weight"], sd[prefix + ".bias"] = _ln(rng, n)

    conv("pff_classifier.PWF_Conv0", 1, d)
    lnorm("pff_classifier.layer_norm", 1)
    if mode == "adj":
        linear("node_embedding.next_w.FF_Linear0", d, d)
        for i, n in enumerate(num):
            p = f"node_embedding.Embedding_Linear{i}."
            sd[p + "tied weight_0"] = _uniform(rng, (d, n), 1.0 / np.sqrt(n))
            sd[p + "tied bias1"] = _uniform(rng, (d,), 1.0 / np.sqrt(d))
            sd[p + "tied bias2"] = _uniform(rng, (n,), 1.0 / np.sqrt(d))
            sd[p + "tied weight_1"] = _uniform(rng, (d, d), 1.0 / np.sqrt(d))
            linear(f"node_embedding.Embedding_recon{i}.FF_Linear0", n, d)
    elif mode == "table":
        w = rng.standard_normal((N + 1, d)).astype(np.float32)
        w[0] = 0.0                                    # nn.Embedding(padding_idx=0)
        sd["node_embedding.weight"] = w
    else:
        raise ValueError(mode)
    for enc in ("encode1", "encode2"):
        a = enc + ".mul_head_attn."
        std = np.sqrt(2.0 / (d + d))
        for nm in ("w_qs", "w_ks", "w_vs"):
            sd[a + nm + ".weight"] = (std * rng.standard_normal((n_head * d, d))).astype(np.float32)
        linear(a + "fc1", d, n_head * d)
        linear(a + "fc2", d, n_head * d)
        for nm in ("layer_norm1", "layer_norm2", "layer_norm3"):
            lnorm(a + nm, d)
        for pff in ("pff_n1", "pff_n2"):
            conv(f"{enc}.{pff}.PWF_Conv0", d, d)
            conv(f"{enc}.{pff}.PWF_Conv1", d, d)
            lnorm(f"{enc}.{pff}.layer_norm", d)
    lnorm("layer_norm1", d)
    lnorm("layer_norm2", d)
    linear("next_w.FF_Linear0", d, d)
    sd["attribute_dict_embedding.weight"] = attr_table.astype(np.float32)
    linear("attribute_nn", d, C + 1)
    sd["attribute_dict.weight"] = sd["attribute_dict_embedding.weight"]
    return sd


def make_batch(rng: np.random.Generator, n_nodes: int, ks: List[int], rows_per_k: int, L: int = 0):
    """A zero-padded mixed-k batch: x int64 [B,L] (rows ascending, 0 = pad), y, w float32 [B,1]."""
    L = max(L, max(ks))
    xs = []
    for k in ks:
        e = make_edges(rng, n_nodes, k, rows_per_k)
        xs.append(np.pad(e, ((0, 0), (0, L - k))))
    x = np.concatenate(xs, axis=0)
    perm = rng.permutation(len(x))
    x = x[perm]
    y = (rng.random((len(x), 1)) < 0.25).astype(np.float32)
    w = np.where(y > 0, rng.uniform(0.5, 4.0, size=y.shape), 1.0).astype(np.float32)
    return x, y, w


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4] ("C5"): 1 M nodes, up to 100 M hyperedges with k uniform in {2..8}, built ON THE DEVICE
# (node ids uniform -- the worst case for caches --, int32 CSR shards per rank).
# ------------------------------------------------------------------------------------------------------------------
def make_edges_device(n_nodes: int, n_edges: int, ks=(2, 3, 4, 5, 6, 7, 8), seed: int = 5, device="cuda", chunk: int = 1 << 24, zipf: bool = False):
    """int64 [n_edges, max(ks)] zero-padded rows on ``device``: k uniform over ``ks``, the k nodes a uniform k-subset of
    1..n_nodes in ascending order.  Built without rejection: k sorted draws u_0 <= ... <= u_{k-1} from [0, n_nodes - k] plus
    their rank i are strictly ascending and uniform over the k-subsets (the classic bijection between k-multisets of
    [0, n - k] and k-subsets of [0, n - 1]).  Duplicate ROWS are possible (k = 2: ~1e-4 of 100 M rows at 1 M nodes) and
    harmless: the hash set keeps one copy.  Generated in chunks so that the sort's scratch stays small.
    ``zipf``: node ids drawn with P(id) ~ 1 / id (Zipf(1.0), SURVEY.md §8 d2's optional cached-regime run: id = floor((N + 1)^u)),
    made distinct inside a row by pushing equal neighbours up by one."""
    import torch
    L = max(ks)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ks_t = torch.tensor(list(ks), device=device)
    out = torch.empty((n_edges, L), dtype=torch.long, device=device)
    col = torch.arange(L, device=device).view(1, L)
    for lo in range(0, n_edges, chunk):
        m = min(chunk, n_edges - lo)
        k = ks_t[torch.randint(len(ks), (m,), generator=g, device=device)].view(m, 1)
        if zipf:
            import math
            r = torch.rand((m, L), generator=g, device=device, dtype=torch.float64)
            ids = torch.exp(r * math.log(n_nodes + 1.0)).long().clamp_(1, n_nodes)
            ids = torch.where(col < k, ids, torch.full_like(ids, n_nodes + 2 * L))
            ids, _ = torch.sort(ids, dim=1)
            for c in range(1, L):                        # strictly ascending: equal neighbours move up by one
                ids[:, c] = torch.maximum(ids[:, c], ids[:, c - 1] + 1)
            over = (torch.where(col < k, ids, torch.zeros_like(ids)).amax(1, keepdim=True) - n_nodes).clamp_(min=0)
            ids = ids - over                             # a row pushed past n_nodes slides back down (still ascending, >= 1 for n_nodes >> L)
            out[lo:lo + m] = torch.where(col < k, ids.clamp_(min=1), torch.zeros_like(ids))
            continue
        u = (torch.rand((m, L), generator=g, device=device, dtype=torch.float64) * (n_nodes - k + 1).to(torch.float64)).long()
        u = torch.minimum(u, (n_nodes - k).expand(m, L))
        u = torch.where(col < k, u, torch.full_like(u, n_nodes + L))            # unused slots sort to the end
        u, _ = torch.sort(u, dim=1)
        out[lo:lo + m] = torch.where(col < k, u + col + 1, torch.zeros_like(u))
    return out


def edges_to_csr(edges, rank: int = 0, world: int = 1):
    """The rank's shard of a zero-padded edge list as int32 CSR: rows rank, rank + world, ... ->
    (offsets int64 [m + 1], ids int32 [nnz]) on the same device."""
    import torch
    e = edges[rank::world]
    k = (e != 0).sum(1)
    offsets = torch.zeros(len(e) + 1, dtype=torch.long, device=e.device)
    torch.cumsum(k, 0, out=offsets[1:])
    ids = e[e != 0].to(torch.int32)
    return offsets, ids


def csr_to_padded(offsets, ids, L: int, rows=None):
    """Rows ``rows`` (LongTensor of row indices; default all) of a CSR shard as int64 [m, L] zero-padded -- what the sampler
    and the model take (main.py:433 hands int64; 0 = padding)."""
    import torch
    if rows is None:
        rows = torch.arange(len(offsets) - 1, device=offsets.device)
    lo, hi = offsets[rows], offsets[rows + 1]
    col = torch.arange(L, device=offsets.device).view(1, L)
    idx = lo.view(-1, 1) + col
    valid = idx < hi.view(-1, 1)
    out = ids[torch.where(valid, idx, torch.zeros_like(idx))].to(torch.long)
    return torch.where(valid, out, torch.zeros_like(out))
