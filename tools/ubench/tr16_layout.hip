// What ds_read_b64_tr_b16 hands each lane (gfx950): fused_bwd.hip's column fragments rely on it.
// Tile [32 rows][72 columns] of 16-bit elements, element value = 100 * row + column.  Lane l of a 16-lane group g supplies the address of
// row 8 g + ((l & 15) >> 2), columns c0 + 4 (l & 3); expected: lane (g, i) receives column c0 + i of rows 8 g .. 8 g + 3.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr16_layout.hip -o tools/ubench/tr16_layout ; prints "OK" or the first mismatches
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int c0) {
  __shared__ __attribute__((aligned(16))) short tile[32 * 72];
  for (int i = threadIdx.x; i < 32 * 72; i += 64) tile[i] = (short)(100 * (i / 72) + i % 72);
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&tile[(8 * g + q) * 72 + c0 + 4 * p]);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  int bad = 0;
  for (int c0 = 0; c0 < 64; c0 += 16) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c0);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e) {
        const int want = 100 * (8 * (l >> 4) + e) + c0 + (l & 15);
        if (h[l * 4 + e] != want && bad++ < 8) printf("c0 %d lane %d elem %d: got %d want %d\n", c0, l, e, h[l * 4 + e], want);
      }
  }
  printf(bad ? "MISMATCH (%d)\n" : "OK: lane i of group g holds column c0 + i of rows 8 g .. 8 g + 3\n", bad);
  return bad != 0;
}
