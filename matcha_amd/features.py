"""Feature construction for the adj front end on the device.

  corrcoef_features   main.py:569-577     per chromosome: np.corrcoef of the intra-chromosomal contact block (float32 in,
                                          float64 arithmetic on the f64 MFMA, float32 out), NaN -> 0
  zscore_rows_        Modules.py:146-152  per row of the inter-chromosomal matrix: z-score of the strictly positive entries in
                                          place, NaN -> 0 (what ``MultipleEmbedding.__init__`` does to ``inter_initial``)

Both are one-off preprocessing; at 100 kb bins (N = 30 344) they are 0.3 TFLOP of float64 GEMM and a 3.7 GB in-place pass,
which the reference does with numpy / a python loop over rows.  Kernels: csrc/features.hip.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib


def corrcoef_features(intra_adj, chrom_range, device="cuda") -> List[torch.Tensor]:
    """``intra_adj`` [N, N] (numpy or tensor, any float dtype; used as float32 like main.py:570), ``chrom_range`` [C, 2] of
    1-based [start, end) node ids -> list of float32 [n_i, n_i] correlation matrices on the device."""
    lib = _lib.load()
    adj = torch.as_tensor(intra_adj).to(device=device, dtype=torch.float32).contiguous()
    N = adj.shape[1]
    cr = np.asarray(chrom_range, dtype=np.int64)
    n_max = int((cr[:, 1] - cr[:, 0]).max())
    ws_bytes = lib.matcha_corrcoef_workspace_bytes(n_max)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=adj.device)
    st = C.c_void_p(torch.cuda.current_stream(adj.device).cuda_stream)
    out = []
    for lo, hi in cr:
        n = int(hi - lo)
        o = torch.empty((n, n), dtype=torch.float32, device=adj.device)
        blk = C.c_void_p(adj.data_ptr() + 4 * ((int(lo) - 1) * N + (int(lo) - 1)))
        _lib.check(lib.matcha_corrcoef_block(blk, N, n, _lib.ptr(o), _lib.ptr(ws), ws_bytes, st), "matcha_corrcoef_block")
        out.append(o)
    return out


def zscore_rows_(inter: torch.Tensor) -> torch.Tensor:
    """In place on a contiguous float32 device matrix; returns it."""
    lib = _lib.load()
    if not (inter.is_cuda and inter.dtype == torch.float32 and inter.is_contiguous() and inter.dim() == 2):
        raise ValueError("zscore_rows_ needs a contiguous float32 [rows, cols] tensor on the GPU")
    st = C.c_void_p(torch.cuda.current_stream(inter.device).cuda_stream)
    _lib.check(lib.matcha_zscore_rows(_lib.ptr(inter), inter.shape[0], inter.shape[1], st), "matcha_zscore_rows")
    return inter


def build_features(temp_dir: str, chrom_range, device="cuda") -> Tuple[List[torch.Tensor], torch.Tensor]:
    """main.py:568-577 from the files process.py writes: (per-chromosome correlation features, RAW inter matrix float32) on
    the device -- ``MultipleEmbedding`` z-scores the inter matrix itself, as in the reference."""
    inter = torch.from_numpy(np.load(os.path.join(temp_dir, "inter_adj.npy")).astype("float32")).to(device)
    adj = np.load(os.path.join(temp_dir, "intra_adj.npy")).astype("float32")
    return corrcoef_features(adj, chrom_range, device), inter
