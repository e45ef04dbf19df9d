// issue rate of the VALU instructions the three-way bf16 split is built from (one wave per SIMD, 8 independent chains each)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float v[8]; uint32_t u[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
      if (OP == 1) { bf16x2 p = {(__bf16)v[i], (__bf16)v[(i + 1) & 7]}; u[i] ^= __builtin_bit_cast(uint32_t, p); v[i] += 1.f; }   // cvt_pk + xor + add
      if (OP == 2) u[i] = __builtin_amdgcn_perm(u[i], u[(i + 1) & 7], 0x07060302);
      if (OP == 3) u[i] = (u[i] & 0xffff0000u) + 3u;
      if (OP == 4) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.001f;        // exp + mul
      if (OP == 5) u[i] = u[i] * 0x2C1B3C6Du;
      if (OP == 6) v[i] = v[i] - __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v[i]) & 0xffff0000u);   // and + sub
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}
template <int OP> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
  float* d; hipMalloc(&d, 4096);
  const int it = 20000;
  const float base = run<0>(d, it);
  printf("per-iteration groups of 8 independent ops, 1 wave per SIMD, %d iterations; time relative to v_fma_f32 (%.0f us):\n", it, base);
  printf("  cvt_pk_bf16 + xor + add : %.2f (3 instructions)\n", run<1>(d, it) / base);
  printf("  v_perm_b32              : %.2f\n", run<2>(d, it) / base);
  printf("  and + add (u32)         : %.2f (2 instructions)\n", run<3>(d, it) / base);
  printf("  v_exp_f32 + mul         : %.2f (2 instructions)\n", run<4>(d, it) / base);
  printf("  v_mul_lo_u32            : %.2f\n", run<5>(d, it) / base);
  printf("  and + sub (f32)         : %.2f (2 instructions)\n", run<6>(d, it) / base);
  return 0;
}
