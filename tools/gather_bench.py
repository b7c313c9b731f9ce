"""HBM roofline of the embedding gather alone (north_star target >= 40 % of the HBM-read roofline at
embed_dim 64): rows[t] = table[ids[t]] through the C ABI (matcha_node_embeddings, table mode -> gather_rows_kernel),
uniform random ids.  Algorithmic bytes per row: 4d + 8 read (+ 4d written because the rows are materialised)."""
import ctypes as C
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matcha_amd import _lib

lib = _lib.load()
HBM_PEAK = 8.0e12
print("| table | d | rows per launch | us | read GB/s (4d+8) | % of 8 TB/s read | read+write GB/s |")
print("|---|---:|---:|---:|---:|---:|---:|")
for name, N, d, T in (("C2 (3 067 nodes, L2-resident)", 3067, 64, 1 << 20), ("C5-like 1 M nodes", 1 << 20, 64, 1 << 24),
                      ("16 M nodes (4 GB, beyond the 256 MB cache)", 1 << 24, 64, 1 << 24), ("C5 1 M nodes", 1 << 20, 256, 1 << 22),
                      ("16 M nodes, one training step's tokens per launch", 1 << 24, 64, 65536 * 5),
                      ("C5 1 M nodes, one C5 step's tokens per launch", 1 << 20, 256, 16384 * 8)):
    table = torch.randn(N + 1, d, device="cuda")
    ids = torch.randint(1, N + 1, (T,), device="cuda", dtype=torch.int64)
    rows = torch.empty(T, d, device="cuda")
    shp = _lib.Shape()
    shp.d, shp.n_attr, shp.n_nodes, shp.n_chrom, shp.mode, shp.max_bins = d, 1, N, 1, 0, 0
    par, fro = _lib.Tensors(), _lib.Frozen()
    par.table = table.data_ptr()
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(lib.matcha_node_embeddings(C.byref(shp), C.byref(par), C.byref(fro), _lib.ptr(ids), T, _lib.ptr(rows), None, 0, None, st), "gather")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    assert torch.equal(rows[:1000], table[ids[:1000]])
    rd = T * (4 * d + 8) / (us * 1e-6)
    print(f"| {name} | {d} | {T} | {us:.1f} | {rd / 1e9:.0f} | {100 * rd / HBM_PEAK:.1f} | {T * (8 * d + 8) / (us * 1e-6) / 1e9:.0f} |")
    del table, ids, rows
