// Candidate GEMM main loop for round 2: both operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR
// staging, no ds_write), LDS image row-major [row][16 k] with an XOR swizzle of the four 16-byte chunks of a row
// (applied on the SOURCE side: lane -> which chunk it fetches), fragments read with ds_read_b128 (4 k-values per read,
// lane half h takes chunk 2 jj + h), two LDS stages, ONE barrier per k-tile.  Same traffic pattern as mfma_lds.hip mode 4.
// hipcc --offload-arch=gfx950 -O3 mfma_glds.hip -o mfma_glds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void glb_void;
constexpr int BK = 16, TILE_F = 128 * BK;      // floats per operand tile (8 KB)

__device__ __forceinline__ int swz(int row, int q) { return q ^ ((row >> 1) & 3); }

template <int STAGES>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters, const float* __restrict__ src, long src_floats) {
  __shared__ __attribute__((aligned(16))) float lds[STAGES * 2 * TILE_F];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const long row_stride = 992;
  // wave w fills LDS chunks [w*128 + i*64 + lane] (i = 0, 1) of each operand tile: chunk id s -> row s/4, slot s%4;
  // the lane fetches the k-quad q whose swizzled slot is s%4
  long goff[2];
  int ldsoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = wave * 128 + i * 64 + lane, row = s >> 2, slot = s & 3, q = swz(row, slot);
    goff[i] = ((long)blockIdx.x * 128 + row) * row_stride + q * 4;
    ldsoff[i] = (wave * 128 + i * 64) * 4;                // floats; lane * 16 B is added by the hardware
  }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  auto issue = [&](int it, int stage) {
    float* A = lds + stage * 2 * TILE_F;
    float* B = A + TILE_F;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long base = (goff[i] + (long)it * BK) % (src_floats - 140 * row_stride);
      __builtin_amdgcn_global_load_lds((glb_void*)(src + base), (lds_void*)(A + ldsoff[i]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)(src + base + 64 * row_stride), (lds_void*)(B + ldsoff[i]), 16, 0, 0);
    }
  };
  issue(0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  const int ar0 = (wave >> 1) * 64 + c, br0 = (wave & 1) * 64 + c;
  for (int it = 0; it < iters; ++it) {
    const int cur = it % STAGES;
    issue(it + 1, (it + 1) % STAGES);
    const float* A = lds + cur * 2 * TILE_F;
    const float* B = A + TILE_F;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int q = 2 * jj + h;
      const float4 a0 = *reinterpret_cast<const float4*>(&A[(ar0 * 4 + swz(ar0, q)) * 4]);
      const float4 a1 = *reinterpret_cast<const float4*>(&A[((ar0 + 32) * 4 + swz(ar0 + 32, q)) * 4]);
      const float4 b0 = *reinterpret_cast<const float4*>(&B[(br0 * 4 + swz(br0, q)) * 4]);
      const float4 b1 = *reinterpret_cast<const float4*>(&B[((br0 + 32) * 4 + swz(br0 + 32, q)) * 4]);
      c0 = MFMA(a0.x, b0.x, c0); c1 = MFMA(a0.x, b1.x, c1); c2 = MFMA(a1.x, b0.x, c2); c3 = MFMA(a1.x, b1.x, c3);
      c0 = MFMA(a0.y, b0.y, c0); c1 = MFMA(a0.y, b1.y, c1); c2 = MFMA(a1.y, b0.y, c2); c3 = MFMA(a1.y, b1.y, c3);
      c0 = MFMA(a0.z, b0.z, c0); c1 = MFMA(a0.z, b1.z, c1); c2 = MFMA(a1.z, b0.z, c2); c3 = MFMA(a1.z, b1.z, c3);
      c0 = MFMA(a0.w, b0.w, c0); c1 = MFMA(a0.w, b1.w, c1); c2 = MFMA(a1.w, b0.w, c2); c3 = MFMA(a1.w, b1.w, c3);
    }
    __builtin_amdgcn_s_waitcnt(0);            // this wave's LDS-DMA pieces of the next tile have landed
    __syncthreads();
  }
  out[blockIdx.x * 256 + tid] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int STAGES> void run(float* d, const char* name, const float* src, long n) {
  const int blocks = 256 * 4, iters = 4000;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  k<STAGES><<<blocks, 256>>>(d, 100, src, n);
  hipEventRecord(s); k<STAGES><<<blocks, 256>>>(d, iters, src, n); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  double fl = (double)blocks * 4 * iters * (BK / 2) * 4 * (2.0 * 32 * 32 * 2);
  printf("%-52s %.1f TF (%.2f ms)\n", name, fl / ms / 1e9, ms);
}
int main() {
  float* d; if (hipMalloc(&d, 1024 * 256 * 4) != hipSuccess) return 1;
  const long big = 1L << 30, small = 1L << 21;
  float* src; if (hipMalloc(&src, big * 4) != hipSuccess) return 1;
  hipMemset(src, 0, big * 4);
  for (int rep = 0; rep < 2; ++rep) {
    run<2>(d, "LDS-DMA tiles, b128 fragments, 2 stages (HBM stream)", src, big);
    run<2>(d, "LDS-DMA tiles, b128 fragments, 2 stages (L2 window)", src, small);
  }
  return 0;
}
