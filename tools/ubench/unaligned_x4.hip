// Does a global_load_dwordx4 on a 4-byte-aligned (not 16-byte-aligned) address work on this device, and what does it cost?
// (rows of the adj front end's feature blocks start at arbitrary float offsets)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void k4(const float* __restrict__ in, float* __restrict__ out, int stride, int off, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const f4u v = *reinterpret_cast<const f4u*>(in + off + (size_t)stride * t);
  out[t] = v.x + 2.f * v.y + 3.f * v.z + 4.f * v.w;
}
__global__ void k1(const float* __restrict__ in, float* __restrict__ out, int stride, int off, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float* p = in + off + (size_t)stride * t;
  out[t] = p[0] + 2.f * p[1] + 3.f * p[2] + 4.f * p[3];
}
int main() {
  const int n = 1 << 22, stride = 7;
  std::vector<float> h((size_t)n * stride + 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 1000) * 0.001f;
  float *in, *o4, *o1;
  hipMalloc(&in, h.size() * 4); hipMalloc(&o4, n * 4); hipMalloc(&o1, n * 4);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int off = 0; off < 4; ++off) {
    hipLaunchKernelGGL(k4, dim3(n / 256), dim3(256), 0, 0, in, o4, stride, off, n);
    hipLaunchKernelGGL(k1, dim3(n / 256), dim3(256), 0, 0, in, o1, stride, off, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("off %d: FAULT\n", off); return 1; }
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), o4, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += a[i] != b[i];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms4, ms1;
    hipEventRecord(e0); for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k4, dim3(n / 256), dim3(256), 0, 0, in, o4, stride, off, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms4, e0, e1);
    hipEventRecord(e0); for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k1, dim3(n / 256), dim3(256), 0, 0, in, o1, stride, off, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
    printf("offset %d floats: dwordx4 on 4-byte alignment %s (%d mismatches), %.1f us vs four dword loads %.1f us\n", off, bad ? "WRONG" : "ok", bad, ms4 * 50.f, ms1 * 50.f);
  }
  return 0;
}
