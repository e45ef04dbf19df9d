// Where do the idle MFMA cycles of the f32 GEMM inner loop come from?  Same per-k-pair pattern as
// gemm.hip (4 LDS fragment reads + 4 v_mfma_f32_32x32x2_f32), no global traffic:
//   mode 0: MFMA only                      mode 1: + LDS reads, read right before use (gemm.hip today)
//   mode 2: + LDS reads one k-pair ahead   mode 3: mode 1 + a workgroup barrier every 8 k-pairs
//   mode 4: mode 3 + register-staged global tile loads (k-contiguous, HBM stream) + transposing ds_write + 2nd barrier
//   mode 5: mode 4 with the global loads confined to an L2-resident window
// hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
constexpr int SA = 129, BK = 16;
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters, const float* __restrict__ src, long src_floats) {
  __shared__ float lds[2 * BK * SA];
  for (int i = threadIdx.x; i < 2 * BK * SA; i += 256) lds[i] = 1e-3f * (i & 15);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* as = lds + (lane >> 5) * SA + (wave >> 1) * 64 + (lane & 31);
  const float* bs = lds + BK * SA + (lane >> 5) * SA + (wave & 1) * 64 + (lane & 31);
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float a0 = 1.f, a1 = 1.f, b0 = 1.f, b1 = 1.f;
  if (MODE == 2) { a0 = as[0]; a1 = as[32]; b0 = bs[0]; b1 = bs[32]; }
  // staging pattern of gemm.hip (k-contiguous operand): thread -> row tid/4 (+64), 4 consecutive k
  const long row_stride = 992;
  long goff = ((long)blockIdx.x * 128 + (threadIdx.x >> 2)) * row_stride + (threadIdx.x & 3) * 4;
  float4 ra[2], rb[2];
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 4) {
      const long base = (goff + (long)it * BK) % (src_floats - 70 * row_stride);
      ra[0] = *reinterpret_cast<const float4*>(src + base);
      ra[1] = *reinterpret_cast<const float4*>(src + base + 64 * row_stride);
      rb[0] = *reinterpret_cast<const float4*>(src + base + 8);
      rb[1] = *reinterpret_cast<const float4*>(src + base + 64 * row_stride + 8);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      if (MODE == 1 || MODE == 3) { a0 = as[kk * SA]; a1 = as[kk * SA + 32]; b0 = bs[kk * SA]; b1 = bs[kk * SA + 32]; }
      float na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (MODE == 2) { const int k2 = (kk + 2) % BK; na0 = as[k2 * SA]; na1 = as[k2 * SA + 32]; nb0 = bs[k2 * SA]; nb1 = bs[k2 * SA + 32]; }
      c0 = MFMA(a0, b0, c0); c1 = MFMA(a0, b1, c1); c2 = MFMA(a1, b0, c2); c3 = MFMA(a1, b1, c3);
      if (MODE == 2) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
    }
    if (MODE >= 3) __syncthreads();
    if (MODE >= 4) {
      float* wa = lds + (threadIdx.x & 3) * 4 * SA + (threadIdx.x >> 2);
      float* wb = wa + BK * SA;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wa[0 * SA + 64 * i] = ra[i].x * 1e-9f; wa[1 * SA + 64 * i] = ra[i].y * 1e-9f; wa[2 * SA + 64 * i] = ra[i].z * 1e-9f; wa[3 * SA + 64 * i] = ra[i].w * 1e-9f;
        wb[0 * SA + 64 * i] = rb[i].x * 1e-9f; wb[1 * SA + 64 * i] = rb[i].y * 1e-9f; wb[2 * SA + 64 * i] = rb[i].z * 1e-9f; wb[3 * SA + 64 * i] = rb[i].w * 1e-9f;
      }
      __syncthreads();
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int MODE> void run(float* d, const char* name, const float* src, long src_floats) {
  const int blocks = 256 * 4, iters = 4000;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  k<MODE><<<blocks, 256>>>(d, 100, src, src_floats);
  hipEventRecord(s); k<MODE><<<blocks, 256>>>(d, iters, src, src_floats); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  double fl = (double)blocks * 4 * iters * (BK / 2) * 4 * (2.0 * 32 * 32 * 2);
  printf("%-44s %.1f TF (%.2f ms)\n", name, fl / ms / 1e9, ms);
}
int main() {
  float* d; if (hipMalloc(&d, 1024 * 256 * 4) != hipSuccess) return 1;
  const long big = 1L << 30, small = 1L << 21;        // 4 GB stream, 8 MB window
  float* src; if (hipMalloc(&src, big * 4) != hipSuccess) return 1;
  hipMemset(src, 0, big * 4);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(d, "MFMA only", src, big);
    run<1>(d, "+ LDS fragment reads, just in time", src, big);
    run<2>(d, "+ LDS fragment reads, one k-pair ahead", src, big);
    run<3>(d, "just-in-time reads + barrier / 8 k-pairs", src, big);
    run<4>(d, "+ global tile loads (HBM) + ds_write + barrier", src, big);
    run<5>(d, "+ global tile loads (L2 window) + ds_write", src, small);
  }
  return 0;
}
