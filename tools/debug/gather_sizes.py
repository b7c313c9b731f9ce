"""Gather read fraction vs rows per launch (output footprint relative to the 256 MB Infinity Cache), HBM-resident tables."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from matcha_amd import _lib
lib = _lib.load()
for N, d in ((1 << 24, 64), (1 << 20, 256)):
    table = torch.randn(N + 1, d, device="cuda")
    for T in (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22):
        ids = torch.randint(1, N + 1, (T,), device="cuda", dtype=torch.int64)
        rows = torch.empty(T, d, device="cuda")
        shp = _lib.Shape(); shp.d, shp.n_attr, shp.n_nodes, shp.n_chrom, shp.mode, shp.max_bins = d, 1, N, 1, 0, 0
        par, fro = _lib.Tensors(), _lib.Frozen(); par.table = table.data_ptr()
        st = torch.cuda.current_stream().cuda_stream
        def run():
            _lib.check(lib.matcha_node_embeddings(C.byref(shp), C.byref(par), C.byref(fro), _lib.ptr(ids), T, _lib.ptr(rows), None, 0, None, st), "g")
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        rd = T * (4 * d + 8) / (us * 1e-6)
        print(f"N={N} d={d} T={T} out={T*d*4/2**20:.0f}MiB  {us:.1f} us  read {rd/1e9:.0f} GB/s = {100*rd/8e12:.1f}%")
