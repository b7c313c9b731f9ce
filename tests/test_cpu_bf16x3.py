"""The arithmetic of the fused forward's products (round 5; matcha_amd/csrc/fused_fwd32.hip, "the weight stream"), restated in numpy.

Every operand is split into three bf16 planes, v = h + m + l with h = bf16(v), m = bf16(v - h), l = bf16(v - h - m) (round to nearest
even, what v_cvt_pk_bf16_f32 does), and a product is the six plane products  Al Bh + Ah Bl + Am Bm + Am Bh + Ah Bm + Ah Bh  accumulated
in f32 on the bf16 matrix pipe.  This test pins what the kernel's comment claims: the split reconstructs an f32 value to 2^-27, and a
64-term dot product computed this way is as close to the exact result as an f32 fma chain is (a few 1e-8 of sum |a b|) -- far inside the
1e-4 parity tolerance, and NOT a reduced-precision path."""
import numpy as np


def bf16_rne(x):
    """float32 -> the nearest bfloat16 (ties to even), returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, bf16_rne(r2)


def dot_bf16x3(a, b):
    """rows of a . rows of b with the kernel's six plane products, f32 accumulation in the kernel's order (per 16-slot chunk, smallest first)"""
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    acc = np.zeros(a.shape[0], np.float32)
    for c in range(0, a.shape[1], 16):
        s = slice(c, c + 16)
        for x, y in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
            # one MFMA: 16 exact bf16 x bf16 products summed in f32 (the order inside the instruction is not specified; float64 here)
            acc = (acc.astype(np.float64) + (x[:, s].astype(np.float64) * y[:, s].astype(np.float64)).sum(1)).astype(np.float32)
    return acc


def test_split_reconstructs_f32_to_2_pow_minus_27():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000), rng.standard_normal(1000) * 1e-20, rng.standard_normal(1000) * 1e20]).astype(np.float32)
    h, m, l = split3(x)
    rec = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.all(np.abs(rec - x.astype(np.float64)) <= 2.0 ** -26 * np.abs(x.astype(np.float64)))
    # and summed back in f32 -- (h + m) + l, what a kernel would do to recover the value -- it is the value itself
    back = ((h + m).astype(np.float32) + l).astype(np.float32)
    assert np.array_equal(back, x)


def test_six_plane_products_match_an_f32_fma_chain():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((4096, 64)).astype(np.float32)
    b = rng.standard_normal((4096, 64)).astype(np.float32)
    exact = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    scale = (np.abs(a.astype(np.float64)) * np.abs(b.astype(np.float64))).sum(1)
    got = dot_bf16x3(a, b)
    chain = np.zeros(4096, np.float32)
    for k in range(64):                                # the f32 MFMA's arithmetic: a k-ordered fmaf chain
        chain = (chain.astype(np.float64) + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32)
    err_split = np.abs(got - exact) / scale
    err_chain = np.abs(chain - exact) / scale
    assert err_split.max() < 1.5e-7, err_split.max()
    assert err_split.mean() < 2.0 * err_chain.mean() + 1e-9, (err_split.mean(), err_chain.mean())
