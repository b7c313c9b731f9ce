"""Positive-weight preprocessing on the device: from ``all_<k>_counter.npy`` / ``all_<k>_freq_counter.npy``
to the (edges, weights) the training loop consumes and the membership set the negative sampler checks.

Reference flow (Code/main.py):
  :551-566   per k-mer size: QuantileTransformer(n_quantiles=1000, 'uniform') of the frequencies, keep rows whose transformed
             frequency exceeds ``quantile_cutoff_for_positive``; concatenate rows and weights over the sizes
  :594-597   ``weight /= mean(weight); weight *= neg_num``
  :599-605   shuffle, 80 / 20 train / validation split
  :646-667   the same transform with ``quantile_cutoff_for_unlabel`` selects the rows of the membership set (``build_hash``,
             utils.py:75-97) -- here the exact device hash set of ``utils.build_hash`` instead of pybloom_live's Bloom filters

The transform runs in ``matcha_quantile_uniform`` (csrc/quantile.hip): bit-identical to scikit-learn wherever scikit-learn is
deterministic, and fitted on every row where scikit-learn would draw a random 10 000-row subsample.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

N_QUANTILES = 1000            # main.py:555, :653


def quantile_uniform(freq, n_quantiles: int = N_QUANTILES, return_quantiles: bool = False, device="cuda"):
    """float32 [n] frequencies -> float32 [n] in [0, 1] on the device (``QuantileTransformer(...).fit_transform`` of one
    column).  ``return_quantiles``: also the fitted float64 landmarks (``quantiles_``)."""
    lib = _lib.load()
    x = torch.as_tensor(freq).to(device=device, dtype=torch.float32).contiguous().reshape(-1)
    n = x.numel()
    if n == 0:
        out, q = torch.empty(0, dtype=torch.float32, device=x.device), torch.empty(0, dtype=torch.float64, device=x.device)
        return (out, q) if return_quantiles else out
    ws_bytes = lib.matcha_quantile_workspace_bytes(n)
    if ws_bytes == 0:
        raise _lib.MatchaHipError("matcha_quantile_workspace_bytes: n out of range")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    out = torch.empty_like(x)
    q = torch.empty(min(int(n_quantiles), n), dtype=torch.float64, device=x.device) if return_quantiles else None
    st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(lib.matcha_quantile_uniform(_lib.ptr(x), n, int(n_quantiles), _lib.ptr(out), _lib.ptr(q), _lib.ptr(ws), ws_bytes, st),
               "matcha_quantile_uniform")
    return (out, q) if return_quantiles else out


def select_positives(kmers: Sequence, freqs: Sequence, cutoff: float, device="cuda") -> Tuple[torch.Tensor, torch.Tensor]:
    """main.py:551-566 for in-memory arrays: ``kmers[i]`` int [M_i, k_i], ``freqs[i]`` [M_i].  Returns (edges int64 [M, max_k]
    zero-padded on the right -- 0 is the padding id --, transformed weights float32 [M]) on the device, sizes in the given
    order, rows in file order."""
    L = max(int(np.shape(k)[1]) for k in kmers)
    edges, weights = [], []
    for data, freq in zip(kmers, freqs):
        data = torch.as_tensor(np.asarray(data)).to(device=device, dtype=torch.int64)
        w = quantile_uniform(freq, device=device)
        keep = w > cutoff
        rows = data[keep]
        edges.append(torch.nn.functional.pad(rows, (0, L - rows.shape[1])))
        weights.append(w[keep])
    return torch.cat(edges, dim=0), torch.cat(weights, dim=0)


def load_kmers(temp_dir: str, size_list: Sequence[int], cutoff: float, device="cuda") -> Tuple[torch.Tensor, torch.Tensor]:
    """``select_positives`` of the files generate_kmers.py writes (main.py:552-553)."""
    kmers = [np.load(os.path.join(temp_dir, "all_%d_counter.npy" % k)).astype(np.int64) for k in size_list]
    freqs = [np.load(os.path.join(temp_dir, "all_%d_freq_counter.npy" % k)).astype("float32") for k in size_list]
    return select_positives(kmers, freqs, cutoff, device)


def normalise_weights(weight: torch.Tensor, neg_num: float) -> torch.Tensor:
    """main.py:594-597: weights with mean ``neg_num`` (float32)."""
    w = weight.to(torch.float32)
    return w / w.mean() * neg_num


def train_test_split(n: int, rng: Optional[np.random.Generator] = None, train_frac: float = 0.8) -> Tuple[np.ndarray, np.ndarray]:
    """main.py:599-605: a shuffled index split (the reference shuffles with the global numpy state)."""
    index = np.arange(n)
    (rng or np.random.default_rng()).shuffle(index)
    split = int(train_frac * n)
    return index[:split], index[split:]


def prepare(temp_dir: str, size_list: Sequence[int], cutoff_positive: float, cutoff_unlabel: float, neg_num: float,
            rng: Optional[np.random.Generator] = None, device="cuda"):
    """Everything between the k-mer files and ``train()``: ((train_edges, train_w), (test_edges, test_w), membership rows).
    Edges int64 [*, max_k] zero-padded, on the device."""
    data, weight = load_kmers(temp_dir, size_list, cutoff_positive, device)
    weight = normalise_weights(weight, neg_num)
    tr, te = train_test_split(len(data), rng)
    tr_d, te_d = torch.from_numpy(tr).to(data.device), torch.from_numpy(te).to(data.device)
    known, _ = load_kmers(temp_dir, size_list, cutoff_unlabel, device)
    return (data[tr_d], weight[tr_d]), (data[te_d], weight[te_d]), known
