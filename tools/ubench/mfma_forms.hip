// Cycles per v_mfma_f32_32x32x16_bf16 by operand FILE (VGPR / AGPR for A, B, C = D) and by the number of independent accumulator
// chains, one wave per SIMD (256-thread blocks, 160 KB of LDS each -> one block per CU).  Shader cycles from HW_REG_SHADER_CYCLES.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_forms tools/ubench/mfma_forms.hip && tools/ubench/mfma_forms
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MF(CD, CA, CB) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+" CD(acc[c]) : CA(a_), CB(b_))

template <int FORM, int CHAINS>
__global__ __launch_bounds__(256, 1) void k(int iters, float* out, unsigned* cyc) {
  extern __shared__ char lds[];
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 a_, b_;
  for (int i = 0; i < 8; ++i) { a_[i] = (__bf16)(0.001f * (threadIdx.x + i)); b_[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
  if (FORM == 3 || FORM == 5) asm volatile("" : "+a"(b_));
  if (FORM == 4) asm volatile("" : "+a"(a_));
  if (FORM == 1 || FORM == 5) for (int c = 0; c < 4; ++c) asm volatile("" : "+a"(acc[c]));
  unsigned t0 = __builtin_amdgcn_s_getreg((29 << 0) | (0 << 6) | (19 << 11));       // HW_REG_SHADER_CYCLES, 20 bits
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = u % CHAINS;
      if (FORM == 1) MF("a", "v", "v");
      if (FORM == 2) MF("v", "v", "v");
      if (FORM == 3) MF("v", "v", "a");
      if (FORM == 4) MF("v", "a", "v");
      if (FORM == 5) MF("a", "v", "a");
    }
  }
  unsigned t1 = __builtin_amdgcn_s_getreg((29 << 0) | (0 << 6) | (19 << 11));
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = (t1 - t0) & 0xfffff;
}

template <int FORM, int CHAINS>
void run(const char* name) {
  float* out; unsigned* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 4);
  auto kern = k<FORM, CHAINS>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int small = 1000, big = 40000;
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 150000, 0, small, out, cyc);
  unsigned c; hipMemcpy(&c, cyc, 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 150000, 0, big, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 150000, 0, big, out, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s chains %d: %.1f shader cycles / MFMA (short run), %.3f ms for %d MFMAs per wave = %.1f ns-cycles@2.4GHz / MFMA, %.0f TF\n", name, CHAINS,
         c / (small * 8.0), ms, big * 8, ms * 1e-3 * 2.4e9 / (big * 8.0), 256.0 * 4 * big * 8 * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1, 1>("C/D AGPR, A VGPR, B VGPR"); run<1, 2>("C/D AGPR, A VGPR, B VGPR"); run<1, 4>("C/D AGPR, A VGPR, B VGPR");
  run<2, 1>("C/D VGPR, A VGPR, B VGPR"); run<2, 2>("C/D VGPR, A VGPR, B VGPR"); run<2, 4>("C/D VGPR, A VGPR, B VGPR");
  run<3, 1>("C/D VGPR, A VGPR, B AGPR"); run<3, 2>("C/D VGPR, A VGPR, B AGPR");
  run<4, 1>("C/D VGPR, A AGPR, B VGPR"); run<4, 2>("C/D VGPR, A AGPR, B VGPR");
  run<5, 1>("C/D AGPR, A VGPR, B AGPR"); run<5, 2>("C/D AGPR, A VGPR, B AGPR");
  return 0;
}
