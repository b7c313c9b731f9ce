"""Per-kernel parity of the HIP path (through the C ABI) against plain torch-CPU restatements / the oracle.
GPU only (-m gpu)."""
import ctypes as C

import numpy as np
import pytest
import torch

from matcha_amd import _lib
from oracle import rng as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gemm(op, A, B, M, N, K, epi=None, colsum=None, gather=None, C_init=None):
    lib = _lib.load()
    out = torch.zeros(M, N, device=DEV) if C_init is None else C_init.clone()
    ws = None
    wsn = 0
    if op == _lib.GEMM_TN:
        wsn = lib.matcha_gemm_tn_workspace_bytes(M, N, K)
        ws = torch.empty(wsn, dtype=torch.uint8, device=DEV)
    _lib.check(lib.matcha_gemm(op, _lib.ptr(A), _lib.ptr(B), _lib.ptr(out), M, N, K, None if epi is None else C.byref(epi),
                               _lib.ptr(colsum), _lib.ptr(gather), _lib.ptr(ws), wsn, _stream()), "matcha_gemm")
    torch.cuda.synchronize()
    return out


SHAPES = [(128, 64, 64), (1000, 64, 64), (37, 16, 16), (515, 512, 64), (515, 64, 512), (300, 128, 128), (70, 24, 20), (130, 96, 250),
          # the 128 x 128-tile kernels of gemm_wide.hip (N % 128 == 0, K % 32 == 0, K >= 64): embed_dim 128 / 256 layer shapes
          (1000, 1024, 128), (515, 128, 1024), (257, 256, 256), (1, 128, 64), (2049, 2048, 256), (129, 256, 2048)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_nt_nn(M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.2
    ref = (A.double() @ W.double().t()).float()
    out = _gemm(_lib.GEMM_NT, A.to(DEV), W.to(DEV), M, N, K).cpu()
    assert (out - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max())
    Wkn = W.t().contiguous()                       # [K,N]
    out2 = _gemm(_lib.GEMM_NN, A.to(DEV), Wkn.to(DEV), M, N, K).cpu()
    assert (out2 - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max())


def test_gemm_unaligned_rows():
    """K not a multiple of 4 (adj feature rows of n_i floats): the scalar-load path."""
    M, N, K = 200, 64, 137
    g = torch.Generator().manual_seed(5)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
    ref = (A.double() @ W.double().t()).float()
    out = _gemm(_lib.GEMM_NT, A.to(DEV), W.to(DEV), M, N, K).cpu()
    assert (out - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("R_,M,N", [(1000, 64, 64), (5000, 512, 64), (777, 64, 512), (333, 16, 16), (4096, 64, 24), (50, 128, 128),
                                    (5000, 1024, 128), (777, 128, 1024), (33, 256, 256), (1, 128, 128), (20000, 2048, 256)])
def test_gemm_tn(R_, M, N):
    g = torch.Generator().manual_seed(R_ + M + N)
    dY, X = torch.randn(R_, M, generator=g), torch.randn(R_, N, generator=g)
    ref = (dY.double().t() @ X.double()).float()
    refc = dY.double().sum(0).float()
    col = torch.zeros(M, device=DEV)
    out = _gemm(_lib.GEMM_TN, dY.to(DEV), X.to(DEV), M, N, R_, colsum=col).cpu()
    tol = 3e-5 * max(1.0, ref.abs().max()) * max(1.0, (R_ / 1000) ** 0.5)
    assert (out - ref).abs().max() <= tol
    assert (col.cpu() - refc).abs().max() <= tol
    # accumulate flag + determinism
    epi = _lib.GemmEpilogue()
    epi.flags = _lib.EPI_ACCUM
    base = torch.ones(M, N, device=DEV)
    out2 = _gemm(_lib.GEMM_TN, dY.to(DEV), X.to(DEV), M, N, R_, epi=epi, C_init=base).cpu()
    assert (out2 - 1.0 - ref).abs().max() <= tol
    out3 = _gemm(_lib.GEMM_TN, dY.to(DEV), X.to(DEV), M, N, R_).cpu()
    assert torch.equal(out, out3)


@pytest.mark.parametrize("M,N", [(128, 128), (1024, 128), (128, 1024), (2048, 256), (256, 2048)])
def test_gemm_tn_wide_large_r_vs_fp64(M, N):
    """gemm_tn_wide_kernel at the row count of a real step (R = 229 710 tokens: 65 536 mixed-k rows) and the weight shapes of
    embed_dim 128 / 256 against an fp64 product (torch on the device as the checker; the narrow-kernel comparison above is a
    self-comparison).  Weight gradients dW = dY^T X of Modules.py:527-529 / :572."""
    R_ = 229710
    g = torch.Generator(device=DEV).manual_seed(M * 3 + N)
    dY, X = torch.randn(R_, M, generator=g, device=DEV), torch.randn(R_, N, generator=g, device=DEV)
    ref = (dY.double().t() @ X.double())
    refc = dY.double().sum(0)
    col = torch.zeros(M, device=DEV)
    out = _gemm(_lib.GEMM_TN, dY, X, M, N, R_, colsum=col)
    scale = float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) <= 1e-5 * scale, float((out.double() - ref).abs().max()) / scale
    assert float((col.double() - refc).abs().max()) <= 1e-5 * max(scale, float(refc.abs().max()))
    out2 = _gemm(_lib.GEMM_TN, dY, X, M, N, R_)
    assert torch.equal(out, out2)                                  # fixed-order slab reduction: bitwise reproducible


def test_gemm_tn_gather():
    R_, M, N, NT = 3000, 64, 24, 500
    g = torch.Generator().manual_seed(9)
    dY, tab = torch.randn(R_, M, generator=g), torch.randn(NT, N, generator=g)
    idx = torch.randint(0, NT, (R_,), generator=g)
    ref = (dY.double().t() @ tab[idx].double()).float()
    out = _gemm(_lib.GEMM_TN, dY.to(DEV), tab.to(DEV), M, N, R_, gather=idx.to(DEV)).cpu()
    assert (out - ref).abs().max() <= 5e-5 * max(1.0, ref.abs().max())


def test_gemm_epilogues():
    M, N, K = 300, 64, 64
    g = torch.Generator().manual_seed(3)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    aux = torch.tanh(torch.randn(M, N, generator=g))
    ids = torch.randint(0, 3, (M,), generator=g)
    seed = torch.tensor([123456789012345], dtype=torch.int64)
    p = 0.3
    lin = (A.double() @ W.double().t()).float() + bias
    mask = torch.from_numpy(R.dropout_mask(int(seed[0]), R.STREAM_DROP_FC1, p, M, N))
    # forward-style: bias, tanh, residual, dropout, rowmask
    ref = (torch.tanh(lin) + res) * mask * (ids != 0).float()[:, None]
    epi = _lib.GemmEpilogue()
    epi.flags = _lib.EPI_BIAS | _lib.EPI_TANH | _lib.EPI_RESIDUAL | _lib.EPI_DROPOUT | _lib.EPI_ROWMASK
    keep = [bias.to(DEV), res.to(DEV), ids.to(DEV), seed.to(DEV), aux.to(DEV)]
    epi.bias, epi.residual, epi.row_ids, epi.seed = (t.data_ptr() for t in keep[:4])
    epi.stream_id, epi.p_drop, epi.aux_scale = R.STREAM_DROP_FC1, p, 1.0
    out = _gemm(_lib.GEMM_NT, A.to(DEV), W.to(DEV), M, N, K, epi=epi).cpu()
    assert (out - ref).abs().max() <= 2e-5
    # the mask itself must be bit-identical to the oracle's: zeros in the same places
    assert torch.equal((out == 0) | (ids == 0)[:, None], (mask == 0) | (ids == 0)[:, None])
    # backward-style: dropout (same counter), dtanh with aux_scale
    epi2 = _lib.GemmEpilogue()
    epi2.flags = _lib.EPI_DROPOUT | _lib.EPI_DTANH
    epi2.seed, epi2.aux = keep[3].data_ptr(), keep[4].data_ptr()
    epi2.stream_id, epi2.p_drop, epi2.aux_scale = R.STREAM_DROP_FC1, p, 0.7
    ref2 = (A.double() @ W.double().t()).float() * mask * (1 - (aux * 0.7) ** 2)
    out2 = _gemm(_lib.GEMM_NT, A.to(DEV), W.to(DEV), M, N, K, epi=epi2).cpu()
    assert (out2 - ref2).abs().max() <= 2e-5 * max(1.0, ref2.abs().max())


@pytest.mark.parametrize("M,N,K", [(300, 128, 128), (1000, 1024, 128), (130, 128, 1024)])
def test_gemm_wide_tiles_equal_narrow_tiles_and_epilogues(M, N, K):
    """gemm_wide.hip against the 64-wide kernels it replaces at embed_dim >= 128 (option disable_wide_gemm) with the epilogues
    the model uses, dropout masks bit-identical (counter RNG on (row, column)); TN including the column sums."""
    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ids = torch.randint(0, 3, (M,), generator=g)
    seed = torch.tensor([987654321], dtype=torch.int64)
    keep = [bias.to(DEV), res.to(DEV), ids.to(DEV), seed.to(DEV)]
    outs = {}
    for wide in (True, False):
        with _lib.option("disable_wide_gemm", 0 if wide else 1):
            epi = _lib.GemmEpilogue()
            epi.flags = _lib.EPI_BIAS | _lib.EPI_ROWMASK | _lib.EPI_DROPOUT           # fc1's epilogue (Modules.py:572)
            epi.bias, epi.row_ids, epi.seed = keep[0].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr()
            epi.stream_id, epi.p_drop, epi.aux_scale = R.STREAM_DROP_FC1, 0.3, 1.0
            a = _gemm(_lib.GEMM_NT, A.to(DEV), W.to(DEV), M, N, K, epi=epi).cpu()
            epi2 = _lib.GemmEpilogue()
            epi2.flags = _lib.EPI_RESIDUAL | _lib.EPI_ROWMASK                          # pff_n1 conv0 backward
            epi2.residual, epi2.row_ids = keep[1].data_ptr(), keep[2].data_ptr()
            b = _gemm(_lib.GEMM_NN, A.to(DEV), W.t().contiguous().to(DEV), M, N, K, epi=epi2).cpu()
            col = torch.zeros(K, device=DEV)
            dY = (A @ torch.ones(K, N) * 0.01 + res).contiguous()                      # [M, N] "upstream gradient"
            c = _gemm(_lib.GEMM_TN, A.to(DEV), dY.to(DEV), K, N, M, colsum=col).cpu()  # A^T . dY : [K, N], column sums of A
            outs[wide] = (a, b, c, col.cpu())
    ref = (A.double() @ W.double().t()).float()
    mask = torch.from_numpy(R.dropout_mask(int(seed[0]), R.STREAM_DROP_FC1, 0.3, M, N))
    want = (ref + bias) * mask * (ids != 0).float()[:, None]
    assert (outs[True][0] - want).abs().max() <= 2e-5 * max(1.0, float(want.abs().max()))
    assert torch.equal(outs[True][0] == 0, outs[False][0] == 0)                          # same dropout pattern
    for x, y in zip(outs[True], outs[False]):
        assert (x - y).abs().max() <= 3e-5 * max(1.0, float(y.abs().max()))


def _attn_ref(Q, K, V, B, L, d):
    H = _lib.N_HEAD
    q = Q.view(B, L, H, d).permute(0, 2, 1, 3)
    k = K.view(B, L, H, d).permute(0, 2, 1, 3)
    v = V.view(B, L, H, d).permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2) / np.sqrt(d)
    s = s.masked_fill(torch.eye(L, dtype=torch.bool), -1e32)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * L, H * d)
    return o, p


@pytest.mark.parametrize("d", [16, 64, 128, 256])
@pytest.mark.parametrize("L", [1, 2, 3, 5, 7, 8])
@pytest.mark.parametrize("ragged", [False, True])
def test_attention_fwd_bwd(d, L, ragged):
    """Ragged attention kernel vs the padded dense computation the reference performs: hyperedge b has k_b real tokens,
    its L - k_b padding slots all carry the ONE shared padding token's K/V (and are attended, fact 7).  d = 256 is the
    BASELINE configs[4] instantiation of attention_wide.hip."""
    _attention_case(d, L, ragged, 37)


@pytest.mark.parametrize("d,L", [(128, 5), (256, 5), (256, 8)])
def test_attention_wide_many_rows(d, L):
    """attention_wide.hip (embed_dim >= 128) on >= 4 096 token rows -- many workgroups, the streamed-operand path, the pad-token
    gradient slabs reduced over many blocks -- against the same dense reference (Modules.py:449-458)."""
    _attention_case(d, L, True, 1100)


def _attention_case(d, L, ragged, B):
    lib = _lib.load()
    H = _lib.N_HEAD
    g = torch.Generator().manual_seed(d * 10 + L + (1000 if ragged else 0))
    ks = torch.randint(1, L + 1, (B,), generator=g) if ragged else torch.full((B,), L)
    row_off = torch.zeros(B + 1, dtype=torch.int32)
    row_off[1:] = torch.cumsum(ks, 0).to(torch.int32)
    Tr = int(row_off[B])
    # compact tensors: Tr real rows + the padding token's row
    Qc, Kc, Vc = (torch.randn(Tr + 1, H * d, generator=g).requires_grad_(True) for _ in range(3))
    # dense [B*L] view built by gathering compact rows (pads -> row Tr)
    idx = torch.full((B, L), Tr, dtype=torch.long)
    for b in range(B):
        idx[b, :ks[b]] = torch.arange(int(row_off[b]), int(row_off[b + 1]))
    real = (idx != Tr).reshape(-1)
    Q, K, V = Qc[idx.reshape(-1)], Kc[idx.reshape(-1)], Vc[idx.reshape(-1)]
    o_ref, p_ref = _attn_ref(Q, K, V, B, L, d)
    dO_dense = torch.randn(B * L, H * d, generator=g) * real[:, None].float()     # pad queries' outputs are masked downstream
    o_ref.backward(dO_dense)
    dOc = torch.zeros(Tr + 1, H * d)
    dOc[idx.reshape(-1)[real]] = dO_dense[real]
    Qd, Kd, Vd, dOd, ro = (t.detach().to(DEV) for t in (Qc, Kc, Vc, dOc, row_off))
    O = torch.zeros_like(Qd)
    P = torch.zeros(B, H, L, L, device=DEV)
    _lib.check(lib.matcha_attn_fwd(_lib.ptr(Qd), _lib.ptr(Kd), _lib.ptr(Vd), _lib.ptr(ro), B, L, d, _lib.ptr(O), _lib.ptr(P), _stream()))
    dQ, dK, dV = (torch.zeros_like(Qd) for _ in range(3))
    wsn = lib.matcha_attn_bwd_workspace_bytes(B, d)
    ws = torch.empty(wsn, dtype=torch.uint8, device=DEV)
    _lib.check(lib.matcha_attn_bwd(_lib.ptr(Qd), _lib.ptr(Kd), _lib.ptr(Vd), _lib.ptr(P), _lib.ptr(dOd), _lib.ptr(ro), B, L, d, _lib.ptr(dQ),
                                   _lib.ptr(dK), _lib.ptr(dV), _lib.ptr(ws), wsn, _stream()))
    torch.cuda.synchronize()
    o_dense = o_ref.detach()[real]
    assert (O.cpu()[:Tr] - o_dense).abs().max() <= 2e-5 * max(1.0, o_dense.abs().max())
    # probabilities of the real columns
    Pc = P.cpu()
    for b in range(0, B, 5 if B < 100 else 41):
        k = int(ks[b])
        assert (Pc[b, :, :k, :k] - p_ref.detach()[b, :, :k, :k]).abs().max() <= 1e-5
        if k < L:
            assert (Pc[b, :, :k, k] - p_ref.detach()[b, :, :k, k]).abs().max() <= 1e-5
    # gradients: Q of real tokens; K/V of real tokens and of the shared padding token (sum over all its slots)
    for got, leaf in ((dQ, Qc), (dK, Kc), (dV, Vc)):
        ref = leaf.grad
        if leaf is Qc:
            ref = ref.clone()
            ref[Tr] = 0         # pad queries never influence anything downstream
        assert (got.cpu() - ref).abs().max() <= 5e-5 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("d", [16, 64, 128, 256])
def test_embed_and_ln3(d):
    lib = _lib.load()
    T, N, n_attr = 1000, 300, 24
    g = torch.Generator().manual_seed(d)
    table = torch.randn(N + 1, d, generator=g)
    table[0] = 0
    attr = torch.randn(N + 1, n_attr, generator=g)
    attr[0] = 0
    Wa, ba = torch.randn(d, n_attr, generator=g) * 0.2, torch.randn(d, generator=g)
    x = torch.randint(0, N + 1, (T,), generator=g)
    ref = table[x] + attr[x] @ Wa.t() + ba
    dev = [t.to(DEV) for t in (x, table, attr, Wa, ba)]
    x0 = torch.empty(T, d, device=DEV)
    _lib.check(lib.matcha_embed_fwd(_lib.ptr(dev[0]), T, d, _lib.ptr(dev[1]), None, _lib.ptr(dev[2]), n_attr, _lib.ptr(dev[3]),
                                    _lib.ptr(dev[4]), _lib.ptr(x0), _stream()))
    torch.cuda.synchronize()
    assert (x0.cpu() - ref).abs().max() <= 1e-5 * max(1.0, ref.abs().max())
    # scatter-add backward (padding row untouched)
    dx0 = torch.randn(T, d, generator=g)
    dtab = torch.zeros(N + 1, d, device=DEV)
    _lib.check(lib.matcha_embed_scatter_bwd(_lib.ptr(dev[0]), T, d, _lib.ptr(dx0.to(DEV)), _lib.ptr(dtab), _stream()))
    torch.cuda.synchronize()
    refg = torch.zeros(N + 1, d).index_add_(0, x, dx0)
    refg[0] = 0
    assert (dtab.cpu() - refg).abs().max() <= 1e-4
    assert float(dtab[0].abs().max()) == 0.0
    # LayerNorm x3
    X = torch.randn(T, d, generator=g)
    gb = [torch.randn(d, generator=g) for _ in range(6)]
    outs = [torch.empty(T, d, device=DEV) for _ in range(3)]
    stats = torch.empty(T, 2, device=DEV)
    gbd = [t.to(DEV) for t in gb]
    _lib.check(lib.matcha_ln3_fwd(_lib.ptr(X.to(DEV)), T, d, *[_lib.ptr(t) for t in gbd], *[_lib.ptr(t) for t in outs],
                                  _lib.ptr(stats), _stream()))
    torch.cuda.synchronize()
    for i in range(3):
        ref = torch.nn.functional.layer_norm(X, (d,), gb[2 * i], gb[2 * i + 1], 1e-5)
        assert (outs[i].cpu() - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max())


def _oracle_lib():
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libmatcha_oracle.so")
    assert os.path.exists(path), "build it with __graft_entry__.build() (make -C oracle/c)"
    return C.CDLL(path)


@pytest.mark.parametrize("five_launches", [False, True])
@pytest.mark.parametrize("B,L,ks,bad", [(1, 1, [1], False), (7, 3, [0, 1, 3], False), (384, 5, [2, 3, 4, 5], False), (1024, 8, [0, 2, 8], True),
                                        (1000, 8, [8], False), (900, 2, [0], False), (4096, 5, [2, 3, 4, 5], False), (65536, 5, [2, 3, 4, 5], False),
                                        (20000, 8, [0, 2, 8], True), (3000, 8, [8], False), (5000, 2, [0], False)])
def test_ragged_plan_bit_exact_vs_c_oracle(B, L, ks, bad, five_launches):
    """The execution plan is integer / index work: the plan kernels -- five launches, or ONE for batches of up to 1024 rows
    (plan_small_kernel; ``five_launches`` forces the general path there too) -- must equal the plain-C restatement
    (oracle/c/ragged_plan.c) bit for bit -- CSR offsets, compact token lists, per-token keys, tile list -- including rows that
    are all padding, pads in the middle of a row, and node ids outside [0, N] (flagged, read as 0)."""
    import ctypes as C
    lib = _lib.load()
    if five_launches and B > 1024:
        pytest.skip("the general path is the only one at this size")
    _lib.set_option("disable_small_batch", 1 if five_launches else 0)
    ora = _oracle_lib()
    rng = np.random.default_rng(B + L)
    N = 1000
    x = np.zeros((B, L), dtype=np.int64)
    kk = rng.choice(ks, size=B)
    kk = np.minimum(kk, L)
    col = np.argsort(rng.random((B, L)), axis=1)                       # random slots: pads may sit anywhere in the row
    vals = rng.integers(1, N + 1, size=(B, L))
    mask = np.arange(L)[None, :] < kk[:, None]
    np.put_along_axis(x, col, np.where(mask, vals, 0), axis=1)
    if bad:
        x[5, np.nonzero(x[5])[0][:1]] = N + 5
        x[17, np.nonzero(x[17])[0][:1]] = -2
    T = B * L
    xt = torch.from_numpy(x).cuda()
    ws = torch.zeros(lib.matcha_ragged_plan_bytes(B, L), dtype=torch.uint8, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    view = _lib.RaggedView()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        _lib.check(lib.matcha_ragged_plan(_lib.ptr(xt), B, L, N, _lib.ptr(status), _lib.ptr(ws), ws.numel(), C.byref(view), st), "matcha_ragged_plan")
        torch.cuda.synchronize()
    finally:
        _lib.set_option("disable_small_batch", 0)
    base = ws.data_ptr()

    def grab(p, n, dt):
        nbytes = n * torch.empty(0, dtype=dt).element_size()
        return ws[p - base:p - base + nbytes].view(dt).cpu().numpy()
    cap = int(view.tiles_cap)
    got = dict(row_off=grab(view.row_off, B + 1, torch.int32), tok_slot=grab(view.tok_slot, T + 1, torch.int32),
               tok_id=grab(view.tok_id, T + 1, torch.int64), tok_key=grab(view.tok_key, T + 1, torch.int32),
               tok_pos=grab(view.tok_pos, T + 1, torch.int32), count=grab(view.count, 3, torch.int32),
               tile_meta=grab(view.tile_meta, cap * 4, torch.int32))
    hcap = int(view.halves_cap)
    got_half = grab(view.half_meta, hcap * 4, torch.int32)
    got_tt = grab(view.tok_tile, T + 1, torch.int32)
    got_nh = int(grab(view.count, 4, torch.int32)[3])
    ref = dict(row_off=np.zeros(B + 1, np.int32), tok_slot=np.zeros(T + 1, np.int32), tok_id=np.zeros(T + 1, np.int64),
               tok_key=np.zeros(T + 1, np.int32), tok_pos=np.zeros(T + 1, np.int32), count=np.zeros(3, np.int32),
               tile_meta=np.zeros(cap * 4, np.int32))
    ref_status = np.zeros(1, np.int32)
    ora.matcha_oracle_ragged_plan.restype = C.c_int64
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    nt = ora.matcha_oracle_ragged_plan(p(x), C.c_int64(B), C.c_int32(L), C.c_int64(N), p(ref["row_off"]), p(ref["tok_slot"]), p(ref["tok_id"]),
                                       p(ref["tok_key"]), p(ref["tok_pos"]), p(ref["count"]), p(ref["tile_meta"]), C.c_int64(cap), p(ref_status))
    assert nt >= 1
    Tr = int(ref["count"][1])
    assert np.array_equal(got["count"], ref["count"]) and np.array_equal(got["row_off"], ref["row_off"])
    for k_ in ("tok_slot", "tok_id", "tok_pos"):
        assert np.array_equal(got[k_][:Tr + 1], ref[k_][:Tr + 1]), k_            # entries past the shared padding token are scratch
    assert np.array_equal(got["tok_key"], ref["tok_key"])                        # zero in every unused entry
    assert np.array_equal(got["tile_meta"], ref["tile_meta"])
    assert int(status[0].item()) == int(ref_status[0]) == (1 if bad else 0)
    # the tiles partition the hyperedges into runs of whole hyperedges with <= 63 tokens
    tm = ref["tile_meta"].reshape(-1, 4)[:nt]
    assert tm[:, 1].max() <= 63 and tm[:, 3].sum() == B and tm[:, 1].sum() == Tr
    # half tiles (<= 31 tokens, one wavefront of the fused forward each) and the token -> (tile, row) map
    ref_half, ref_tt = np.zeros(hcap * 4, np.int32), np.zeros(T + 1, np.int32)
    ora.matcha_oracle_ragged_halves.restype = C.c_int64
    nh = ora.matcha_oracle_ragged_halves(p(ref["row_off"]), C.c_int64(B), p(ref["tile_meta"]), C.c_int64(nt), p(ref_half), C.c_int64(hcap), p(ref_tt))
    assert nh >= 1 and nh == got_nh
    assert np.array_equal(got_half, ref_half)
    assert np.array_equal(got_tt[:Tr], ref_tt[:Tr])
    hm = ref_half.reshape(-1, 4)[:nh]
    assert hm[:, 1].max() <= 31 and hm[:, 3].sum() == B and hm[:, 1].sum() == Tr
    assert np.array_equal(hm[1:, 0], hm[:-1, 0] + hm[:-1, 1]) and np.array_equal(hm[1:, 2], hm[:-1, 2] + hm[:-1, 3])     # contiguous cover


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_hip_ops_equal_cpu_twins():
    """the byte / index / optimizer entry points against their plain-C `*_cpu` twins (oracle/c/ops_cpu.c, pinned to
    numpy / torch.optim.AdamW by tests/test_cpu_twins.py).  Gather: bit-exact.  Scatter-add: float atomics, order-dependent -> 1e-5.
    AdamW: the same float roundings in both -> <= 1 ulp-level difference (sqrt / division rounding of the device)."""
    lib, cpu = _lib.load(), _oracle_lib()
    rng = np.random.default_rng(5)
    N, d, T = 5000, 64, 40000
    table = rng.standard_normal((N + 1, d)).astype(np.float32)
    ids = rng.integers(0, N + 1, size=T).astype(np.int64)
    rows_cpu = np.zeros((T, d), np.float32)
    assert cpu.matcha_gather_rows_cpu(_p(ids), C.c_int64(T), C.c_int32(d), _p(table), C.c_int64(N), _p(rows_cpu), None) == 0
    tab_d, ids_d = torch.from_numpy(table).cuda(), torch.from_numpy(ids).cuda()
    x0 = torch.empty(T, d, device="cuda")
    shp, ten, fro, status = _lib.Shape(d, 1, N, 0, 0, 0), _lib.Tensors(), _lib.Frozen(), torch.zeros(4, dtype=torch.int32, device="cuda")
    ten.table = tab_d.data_ptr()
    _lib.check(lib.matcha_node_embeddings(C.byref(shp), C.byref(ten), C.byref(fro), _lib.ptr(ids_d), T, _lib.ptr(x0), None, 0, _lib.ptr(status),
                                          _stream()), "matcha_node_embeddings")                       # table mode -> gather_rows_kernel
    assert np.array_equal(x0.cpu().numpy(), rows_cpu) and status.tolist()[0] == 0

    dx0 = rng.standard_normal((T, d)).astype(np.float32)
    dtab_cpu = np.zeros((N + 1, d), np.float32)
    assert cpu.matcha_embed_scatter_bwd_cpu(_p(ids), C.c_int64(T), C.c_int32(d), _p(dx0), _p(dtab_cpu)) == 0
    dtab = torch.zeros(N + 1, d, device="cuda")
    _lib.check(lib.matcha_embed_scatter_bwd(_lib.ptr(ids_d), T, d, _lib.ptr(torch.from_numpy(dx0).cuda()), _lib.ptr(dtab), _stream()), "scatter")
    assert np.abs(dtab.cpu().numpy() - dtab_cpu).max() <= 1e-5 and float(dtab[0].abs().max()) == 0.0

    n = 100_003
    seg_off = np.array([0, 1000, 1001, 50_000, n], np.int64)
    seg_group = np.array([0, 1, 1, 2], np.int32)
    flat = rng.standard_normal(n).astype(np.float32)
    m, v, step = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(4, np.int32)
    dev = {k: torch.from_numpy(a.copy()).cuda() for k, a in dict(p=flat, m=m, v=v, step=step, off=seg_off, grp=seg_group).items()}
    coef = torch.zeros(12, device="cuda")
    for it in range(5):
        g = (rng.standard_normal(n) * 0.05).astype(np.float32)
        touched = np.array([1, it != 2, it != 3], np.int32)
        g_d, t_d = torch.from_numpy(g.copy()).cuda(), torch.from_numpy(touched).cuda()
        assert cpu.matcha_adamw_step_cpu(_p(flat), _p(g), _p(m), _p(v), C.c_int64(n), _p(seg_off), C.c_int32(4), _p(seg_group), _p(touched), _p(step),
                                         C.c_double(1e-3), C.c_double(0.9), C.c_double(0.999), C.c_double(1e-8), C.c_double(1e-2), C.c_double(0.5)) == 0
        _lib.check(lib.matcha_adamw_step(_lib.ptr(dev["p"]), _lib.ptr(g_d), _lib.ptr(dev["m"]), _lib.ptr(dev["v"]), n, _lib.ptr(dev["off"]), 4,
                                         _lib.ptr(dev["grp"]), _lib.ptr(t_d), _lib.ptr(dev["step"]), _lib.ptr(coef), 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0.5,
                                         _stream()), "adamw")
        assert np.array_equal(g_d.cpu().numpy() == 0, g == 0)                      # the same segments had their gradients consumed
    assert dev["step"].cpu().numpy().tolist() == step.tolist() == [5, 4, 4, 4]
    for k, a in (("p", flat), ("m", m), ("v", v)):
        assert np.abs(dev[k].cpu().numpy() - a).max() <= 3e-7 * max(1.0, np.abs(a).max()), k
