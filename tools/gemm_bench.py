"""Layer-shaped GEMMs of embed_dim 128 / 256 through the C ABI (matcha_gemm), timed with HIP events: wide (gemm_wide.hip) against the
64-wide kernels, plus ablations of the wide kernel (fused_dbg bits: 1 no epilogue, 2 no MFMAs, 4 no global loads in the loop)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from matcha_amd import _lib

lib = _lib.load()
T = 229_709


def run(op, M, N, K, reps=5):
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") if op == _lib.GEMM_NT else torch.randn(K, N, device="cuda")
    if op == _lib.GEMM_TN:
        A = torch.randn(K, M, device="cuda")          # [R, M]
        B = torch.randn(K, N, device="cuda")          # [R, N]
    out = torch.zeros(M, N, device="cuda")
    wsn = lib.matcha_gemm_tn_workspace_bytes(M, N, K) if op == _lib.GEMM_TN else 0
    ws = torch.empty(max(wsn, 256), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def go():
        _lib.check(lib.matcha_gemm(op, _lib.ptr(A), _lib.ptr(B), _lib.ptr(out), M, N, K, None, None, None, _lib.ptr(ws), wsn, st))
    go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * M * N * K / ms / 1e9


names = {_lib.GEMM_NT: "NT", _lib.GEMM_NN: "NN", _lib.GEMM_TN: "TN"}
shapes = [(_lib.GEMM_NT, T, 1024, 128), (_lib.GEMM_NT, T, 128, 1024), (_lib.GEMM_NT, T, 128, 128), (_lib.GEMM_NN, T, 1024, 128), (_lib.GEMM_NN, T, 128, 1024),
          (_lib.GEMM_TN, 1024, 128, T), (_lib.GEMM_TN, 128, 1024, T), (_lib.GEMM_TN, 128, 128, T),
          (_lib.GEMM_NT, 82_000, 2048, 256), (_lib.GEMM_NN, 82_000, 256, 2048), (_lib.GEMM_TN, 2048, 256, 82_000)]
for op, M, N, K in shapes:
    line = f"{names[op]} M={M} N={N} K={K}: "
    for label, wide, dbg in (("narrow", 0, 0), ("wide", 1, 0), ("wide-noepi", 1, 1), ("wide-nomfma", 1, 2), ("wide-noload", 1, 4)):
        if op == _lib.GEMM_TN and dbg:
            continue
        _lib.set_option("disable_wide_gemm", 0 if wide else 1)
        _lib.set_option("fused_dbg", dbg)
        ms, tf = run(op, M, N, K)
        line += f"{label} {ms:.3f} ms ({tf:.0f} TF)  "
    _lib.set_option("fused_dbg", 0)
    _lib.set_option("disable_wide_gemm", 0)
    print(line)
