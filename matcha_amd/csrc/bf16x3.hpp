// fp32-accurate products on the bf16 matrix pipe, shared by the kernels that keep their GEMM operands in LDS as bf16 PLANES
// (fused_bwd.hip: embed_dim 64; enc128.hip: embed_dim 128).
//
//   v = h + m + l,  h = bf16(v), m = bf16(v - h), l = bf16(v - h - m)  (round to nearest even: v_cvt_pk_bf16_f32), |v - (h + m + l)| <= 2^-27 |v|
//   a . b = al bh + ah bl + am bm + am bh + ah bm + ah bh   (smallest first; bf16 x bf16 is exact in f32, f32 accumulate; the dropped plane
//   products are below 2^-26) -- six v_mfma_f32_16x16x32_bf16 per 32 contraction indices.  tests/test_cpu_bf16x3.py restates the arithmetic.
//
// A plane tile is [32 token rows][PS bf16] x 3 planes (PLANE = 32 PS bf16 apart).  Two fragment shapes:
//   row fragment     eight consecutive bf16 of a row (contraction over FEATURES): one ds_read_b128 per plane
//   column fragment  eight tokens of ONE column per lane (contraction over TOKENS): two ds_read_b64_tr_b16 per plane -- the hardware transpose
//                    read hands lane i of a 16-lane group column c0 + i of a 4 row x 16 column block; the second read sits HI rows below
//                    the first (which token a contraction slot holds is free as long as both operands of a product agree)
// V8: eight consecutive floats as four packed pairs (v_pk_mul / v_pk_fma: two flops per lane and instruction).
#pragma once
#include "common.hpp"

namespace matcha {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct V8 { f2 a, b, c, d; };
__device__ __forceinline__ V8 ld8(const float* __restrict__ p) {
  const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
  V8 v;
  v.a = f2{x.x, x.y}; v.b = f2{x.z, x.w}; v.c = f2{y.x, y.y}; v.d = f2{y.z, y.w};
  return v;
}
__device__ __forceinline__ void st8(float* __restrict__ p, const V8& v) {
  *reinterpret_cast<float4*>(p) = make_float4(v.a.x, v.a.y, v.b.x, v.b.y);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v.c.x, v.c.y, v.d.x, v.d.y);
}
__device__ __forceinline__ float dot8(const V8& u, const V8& v) {
  f2 s = u.a * v.a;
  s = __builtin_elementwise_fma(u.b, v.b, s);
  s = __builtin_elementwise_fma(u.c, v.c, s);
  s = __builtin_elementwise_fma(u.d, v.d, s);
  return s.x + s.y;
}
__device__ __forceinline__ V8 scale8(float w, const V8& x) {
  const f2 ww = {w, w};
  V8 y;
  y.a = ww * x.a; y.b = ww * x.b; y.c = ww * x.c; y.d = ww * x.d;
  return y;
}
__device__ __forceinline__ void axpy8(V8& y, float w, const V8& x) {
  const f2 ww = {w, w};
  y.a = __builtin_elementwise_fma(ww, x.a, y.a); y.b = __builtin_elementwise_fma(ww, x.b, y.b);
  y.c = __builtin_elementwise_fma(ww, x.c, y.c); y.d = __builtin_elementwise_fma(ww, x.d, y.d);
}
__device__ __forceinline__ void add8(V8& y, const V8& x) { y.a += x.a; y.b += x.b; y.c += x.c; y.d += x.d; }
__device__ __forceinline__ V8 zero8() { V8 z; z.a = f2{0.f, 0.f}; z.b = z.a; z.c = z.a; z.d = z.a; return z; }

struct Frag3 { u32x4 h, m, l; };
struct P3 { uint32_t h, m, l; };
__device__ __forceinline__ P3 split2(float a, float b) {
  const f2 v = {a, b};
  const bf16x2 hb = __builtin_convertvector(v, bf16x2);                 // v_cvt_pk_bf16_f32: round to nearest even
  const f2 r1 = v - __builtin_convertvector(hb, f2);
  const bf16x2 mb = __builtin_convertvector(r1, bf16x2);
  const f2 r2 = r1 - __builtin_convertvector(mb, f2);
  const bf16x2 lb = __builtin_convertvector(r2, bf16x2);
  return P3{__builtin_bit_cast(uint32_t, hb), __builtin_bit_cast(uint32_t, mb), __builtin_bit_cast(uint32_t, lb)};
}
__device__ __forceinline__ Frag3 split8(const float* v) {
  const P3 a = split2(v[0], v[1]), b = split2(v[2], v[3]), c = split2(v[4], v[5]), d = split2(v[6], v[7]);
  Frag3 f;
  f.h = (u32x4){a.h, b.h, c.h, d.h}; f.m = (u32x4){a.m, b.m, c.m, d.m}; f.l = (u32x4){a.l, b.l, c.l, d.l};
  return f;
}
__device__ __forceinline__ Frag3 split8(const V8& v) {
  const float t[8] = {v.a.x, v.a.y, v.b.x, v.b.y, v.c.x, v.c.y, v.d.x, v.d.y};
  return split8(t);
}

// PS = bf16 per plane row, HI = rows between the two transposed reads of a column fragment
template <int PS, int HI>
struct Planes {
  static constexpr int kPlane = 32 * PS;
  static __device__ __forceinline__ Frag3 row(const short* __restrict__ p) {
    Frag3 f;
    f.h = *reinterpret_cast<const u32x4*>(p); f.m = *reinterpret_cast<const u32x4*>(p + kPlane); f.l = *reinterpret_cast<const u32x4*>(p + 2 * kPlane);
    return f;
  }
  static __device__ __forceinline__ void store(short* __restrict__ p, const Frag3& f) {
    *reinterpret_cast<u32x4*>(p) = f.h; *reinterpret_cast<u32x4*>(p + kPlane) = f.m; *reinterpret_cast<u32x4*>(p + 2 * kPlane) = f.l;
  }
  // p = this lane's address inside its 4 row x 16 column transpose block (EXEC must be all ones: the GEMM phases are)
  static __device__ __forceinline__ u32x4 tr8(const short* __restrict__ p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + HI * PS));
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    return (u32x4){a.x, a.y, b.x, b.y};
  }
  static __device__ __forceinline__ Frag3 col(const short* __restrict__ p) {
    Frag3 f;
    f.h = tr8(p); f.m = tr8(p + kPlane); f.l = tr8(p + 2 * kPlane);
    return f;
  }
};

#define MFMA16B(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (A)), __builtin_bit_cast(bf16x8, (B)), (C), 0, 0, 0)
// acc += A . B over 32 contraction indices: the six plane products above 2^-26, smallest first
__device__ __forceinline__ f32x4 mma6(f32x4 acc, const Frag3& a, const Frag3& b) {
  acc = MFMA16B(a.l, b.h, acc); acc = MFMA16B(a.h, b.l, acc); acc = MFMA16B(a.m, b.m, acc);
  acc = MFMA16B(a.m, b.h, acc); acc = MFMA16B(a.h, b.m, acc); acc = MFMA16B(a.h, b.h, acc);
  return acc;
}

}  // namespace matcha
