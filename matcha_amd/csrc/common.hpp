// Shared device/host helpers for the gfx950 kernels. CDNA4 only: 64-lane wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/matcha_hip.h"

namespace matcha {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

#define MATCHA_CHECK_ARG(cond, ...)              \
  do {                                           \
    if (!(cond)) {                               \
      ::matcha::set_error(__VA_ARGS__);          \
      return MATCHA_EINVAL;                      \
    }                                            \
  } while (0)

// Launch log (matcha_launch_log / matcha_launch_log_read): every launch site names its kernel here, so a test can assert WHICH
// kernels an entry point ran -- a size rule must not move a parity test onto another kernel unnoticed.  Off: one branch.
extern int g_launch_log;
void note_launch(const char* name);

#define MATCHA_CHECK_LAUNCH(name)                                                       \
  do {                                                                                  \
    if (::matcha::g_launch_log) ::matcha::note_launch(name);                            \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess) {                                                            \
      ::matcha::set_error("launch of %s failed: %s", name, hipGetErrorString(e__));     \
      return MATCHA_EHIP;                                                               \
    }                                                                                   \
  } while (0)

#define MATCHA_TRY(expr)          \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != MATCHA_OK) return rc__; \
  } while (0)

// Optional per-kernel-class timing with HIP events on the launch stream (matcha_profile_select/_read):
// bench.py uses it to time the dominant kernel live, inside the timed region.  Off by default (one branch).
extern int g_prof_class;
void prof_record(bool start, double work, hipStream_t st);
struct ProfScope {
  bool on;
  hipStream_t st;
  ProfScope(int cls, double work, hipStream_t s) : on(cls == g_prof_class), st(s) { if (on) prof_record(true, work, st); }
  ~ProfScope() { if (on) prof_record(false, 0.0, st); }
};

// Compute units of the CURRENT device (hipGetDevice), cached per device id: grids and slab counts of the persistent kernels are sized from
// it, and a process may drive more than one device (the per-file function-static caches kept the first device's count).
int device_cu_count();

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- counter-based RNG: the specification is oracle/rng.py (bit-identical) ---------------------
__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t rng_key(uint64_t seed, uint32_t stream) {
  uint32_t k = lowbias32(stream + 0x9E3779B9u);
  k = lowbias32((uint32_t)(seed >> 32) ^ k);
  k = lowbias32((uint32_t)(seed & 0xFFFFFFFFu) ^ k);
  return k;
}
__host__ __device__ __forceinline__ uint32_t rng_u32(uint32_t key, uint32_t hi, uint32_t lo) {
  return lowbias32(lo ^ lowbias32(hi ^ key));
}
__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
constexpr uint32_t kStreamDropAdj = 1, kStreamDropFc1 = 2, kStreamDropPff = 3, kStreamNeg = 16;

// ---- wave-level reductions over groups of 2^k adjacent lanes -----------------------------------
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
// Sum over groups of 8 adjacent lanes with DPP cross-lane moves (VALU rate) instead of ds_bpermute shuffles (each a
// trip through the LDS crossbar): quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror (lane i <-> 7 - i of
// each 8, which pairs the two quads).  Every lane ends with the group's total.
__device__ __forceinline__ float group_sum8_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}
// Sum over groups of 16 adjacent lanes (one DPP row): the 8-lane tree, then row_mirror (lane i <-> 15 - i).
__device__ __forceinline__ float group_sum16_dpp(float v) {
  v = group_sum8_dpp(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

// tanh through one v_exp: (e^{2x} - 1) / (e^{2x} + 1).  Absolute error ~1e-7 (the parity bar is 1e-4 of the output scale);
// libm's tanhf costs ~40 VALU instructions, which are not free next to the f32 MFMAs (tools/ubench/mfma_valu.hip).
__device__ __forceinline__ float fast_tanh(float x) {
  const float xc = fminf(fmaxf(x, -15.f), 15.f);
  const float e = __expf(2.f * xc);
  return (e - 1.f) * __builtin_amdgcn_rcpf(e + 1.f);
}

template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  return v;
}

}  // namespace matcha
