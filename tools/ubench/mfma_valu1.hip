// One wave per SIMD: how much of a wave's OWN VALU / LDS work hides under its own v_mfma_f32_32x32x16_bf16 stream?
// Per iteration 4 MFMAs (two accumulator chains), each followed by R other instructions:
//   kind 0: independent v_fma_f32 (R distinct registers)      kind 1: ONE dependent v_fma_f32 chain
//   kind 2: v_exp_f32 (transcendental), independent           kind 3: ds_read_b128 (independent, conflict-free)
//   kind 4: v_cvt_pk_bf16_f32 independent                     kind 5: v_pk_add_f32 independent
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/mfma_valu1 tools/ubench/mfma_valu1.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int R, int KIND, int MF>
__global__ __launch_bounds__(256, 1) void k(int iters, float* out) {
  extern __shared__ char lds[];
  f32x16 acc[2];
  for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 a_, b_;
  for (int i = 0; i < 8; ++i) { a_[i] = (__bf16)(0.001f * (threadIdx.x + i)); b_[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
  float x[16]; f32x2 y[16]; f32x4 z[16]; unsigned w[16];
  for (int i = 0; i < 16; ++i) { x[i] = 0.5f + i + threadIdx.x; y[i] = f32x2{x[i], x[i] + 1.f}; z[i] = f32x4{0.f, 0.f, 0.f, 0.f}; w[i] = i; }
  const float m1 = 0.999f, m2 = 1e-3f;
  const unsigned la = (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 1]) : "v"(a_), "v"(b_));
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = (u * R + r) & 15;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m1), "v"(m2));
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(m1), "v"(m2));
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        if (KIND == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(z[i]) : "v"(la), "i"(1024 * ((u * R + r) & 31)));
        if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(x[i]), "v"(m1));
        if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(y[(i + 1) & 15]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = 0.f;
  for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  for (int i = 0; i < 16; ++i) s += x[i] + y[i].x + z[i].x + w[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

template <int R, int KIND, int MF>
float run() {
  float* out;
  hipMalloc(&out, 256 * 256 * 4);
  auto kern = k<R, KIND, MF>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int big = 40000;
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 150000, 0, big, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 150000, 0, big, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms * 1e6f / (big * 4.0f);          // ns per (MFMA + R instructions)
}

template <int KIND>
void row(const char* name) {
  printf("%-28s ns per slot, with MFMA:  R=0 %.1f  R=2 %.1f  R=4 %.1f  R=6 %.1f  R=8 %.1f  R=12 %.1f  | without MFMA: R=4 %.1f  R=8 %.1f  R=12 %.1f\n", name,
         run<0, KIND, 1>(), run<2, KIND, 1>(), run<4, KIND, 1>(), run<6, KIND, 1>(), run<8, KIND, 1>(), run<12, KIND, 1>(), run<4, KIND, 0>(), run<8, KIND, 0>(),
         run<12, KIND, 0>());
}

int main() {
  row<0>("v_fma_f32 independent");
  row<1>("v_fma_f32 dependent chain");
  row<2>("v_exp_f32 independent");
  row<3>("ds_read_b128");
  row<4>("v_cvt_pk_bf16_f32");
  row<5>("v_pk_add_f32 (dep. ring)");
  return 0;
}
