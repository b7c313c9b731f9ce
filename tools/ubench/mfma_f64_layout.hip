// Prints which D[i][j] of v_mfma_f64_16x16x4f64 each (lane, register) holds, given A[i][k] from lane i + 16 k and
// B[k][j] from lane j + 16 k (only k == 0 populated, so D[i][j] = (i + 1) * 100 * (j + 1) identifies i and j).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
  const int l = threadIdx.x;
  const double a = (l / 16 == 0) ? (double)(l % 16 + 1) : 0.0;
  const double b = (l / 16 == 0) ? (double)(l % 16 + 1) * 100.0 : 0.0;
  f64x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  double* d;
  hipMalloc(&d, 256 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 5)
    for (int r = 0; r < 4; ++r) {
      const int v = (int)h[l * 4 + r];
      int i = -1, j = -1;
      for (int ii = 1; ii <= 16; ++ii) if (v % (ii * 100) == 0 && v / (ii * 100) <= 16 && v / (ii * 100) >= 1) { /* ambiguous */ }
      printf("lane %2d reg %d: %d\n", l, r, v);
    }
  return 0;
}
