// Calibrates the practical f32 MFMA ceiling on this chip: pure v_mfma_f32_32x32x2_f32 loop,
// 4 independent accumulators per wave, W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, int iters, float a0, float b0) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float a = a0 + threadIdx.x * 1e-7f, b = b0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 64 * 4);
  for (int wps = 1; wps <= 4; wps *= 2) {
    int blocks = 256 * wps, threads = 256, iters = 20000;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<<<blocks, threads>>>(d, 1000, 1.f, 1.f);
    hipEventRecord(s); k<<<blocks, threads>>>(d, iters, 0.5f, 0.25f); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double fl = (double)blocks * 4 /*waves*/ * iters * 4 * (2.0 * 32 * 32 * 2);
    printf("waves/SIMD %d: %.1f TF (%.2f ms)\n", wps, fl / ms / 1e9, ms);
  }
  return 0;
}
