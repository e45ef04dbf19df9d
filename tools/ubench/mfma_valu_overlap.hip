// Do VALU instructions of one wave hide under the MFMAs of another wave of the same SIMD (and of the same wave)?
// 512 threads per workgroup = 2 waves per SIMD, one workgroup per CU.  Variants:
//   0: waves 0-3 bf16 MFMA loop, waves 4-7 idle      1: waves 0-3 idle, waves 4-7 VALU loop (fma chains)
//   2: waves 0-3 MFMA, waves 4-7 VALU (two waves per SIMD, different roles)
//   3: every wave: MFMA and VALU instructions interleaved 1 : R in one stream
// prints time per variant; "sum" vs "max" behaviour tells whether the VALU port is shared with the matrix pipe's issue.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VAR, int R, int SHAPE>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  const bool do_m = VAR == 0 || (VAR == 2 && wave < 4) || VAR == 3;
  const bool do_v = VAR == 1 || (VAR == 2 && wave >= 4) || VAR == 3;
  if (VAR == 0 && wave >= 4) return;
  if (VAR == 1 && wave < 4) return;
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
      if (SHAPE == 32) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      }
    }
    if (do_v) {
#pragma unroll
      for (int r = 0; r < R; ++r) {       // 8 independent chains x R: 8 R VALU instructions per iteration (per 4 MFMAs)
        v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f); v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
        v4 = __builtin_fmaf(v4, 1.0001f, 0.5f); v5 = __builtin_fmaf(v5, 1.0001f, 0.5f); v6 = __builtin_fmaf(v6, 1.0001f, 0.5f); v7 = __builtin_fmaf(v7, 1.0001f, 0.5f);
      }
    }
  }
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
  for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}
template <int VAR, int R, int SHAPE> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VAR, R, SHAPE>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<VAR, R, SHAPE>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
  float* d; hipMalloc(&d, 4096);
  const int it = 20000;
  printf("4 MFMAs + 8 R VALU per iteration, %d iterations, 256 workgroups x 8 waves (us)\n", it);
  printf("32x32x16: R=1: mfma %.0f valu %.0f two-wave %.0f interleaved %.0f\n", run<0,1,32>(d,it), run<1,1,32>(d,it), run<2,1,32>(d,it), run<3,1,32>(d,it));
  printf("32x32x16: R=2: mfma %.0f valu %.0f two-wave %.0f interleaved %.0f\n", run<0,2,32>(d,it), run<1,2,32>(d,it), run<2,2,32>(d,it), run<3,2,32>(d,it));
  printf("32x32x16: R=4: mfma %.0f valu %.0f two-wave %.0f interleaved %.0f\n", run<0,4,32>(d,it), run<1,4,32>(d,it), run<2,4,32>(d,it), run<3,4,32>(d,it));
  printf("16x16x32: R=1: mfma %.0f valu %.0f two-wave %.0f interleaved %.0f\n", run<0,1,16>(d,it), run<1,1,16>(d,it), run<2,1,16>(d,it), run<3,1,16>(d,it));
  printf("16x16x32: R=2: mfma %.0f valu %.0f two-wave %.0f interleaved %.0f\n", run<0,2,16>(d,it), run<1,2,16>(d,it), run<2,2,16>(d,it), run<3,2,16>(d,it));
  return 0;
}
