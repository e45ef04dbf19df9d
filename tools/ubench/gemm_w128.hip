// GEMM lab (round 3): 256 x 256 block tile, 4 waves, each wave a 128 x 128 sub-tile (4 x 4 blocks of
// v_mfma_f32_32x32x2_f32 = 256 accumulator registers, ONE wave per SIMD), persistent workgroups (one per CU).
// Why: the library yardstick (tools/mb_blas.py, hipBLASLt MT256x256x32 MIWT8_8 WG32_8_1) runs 15-30 % above the shipped
// 128 x 128 / 4-workgroups-per-CU kernel on the step's shapes.  With a 128 x 128 wave tile a k-pair needs 8 fragment reads
// for 16 MFMAs (shipped kernel: 4 for 4), every staged element is reused 256 x (128 x) and a barrier comes every 128-256
// MFMAs per wave, so the single wave of a SIMD can keep the matrix pipe busy if its LDS reads run one k-pair ahead, its
// LDS stores of tile t+1 and its global loads of tile t+2 are spread between the MFMAs of tile t.
//   y[M][N] = x[M][K] . W[N][K]^T   (A_KC = B_KC = true: both k-contiguous, the forward orientation) and
//   dx[M][K'] = dy[M][N'] . W[N'][K'] (A_KC = true, B_KC = false)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_w128.hip -I../../include -L../../hoisdf_amd -lhoisdf_hip \
//        -Wl,-rpath,'$ORIGIN/../../hoisdf_amd' -o gemm_w128
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "hoisdf.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int TM = 256, TN = 256, NT = 256;
constexpr int S_KC = 258;   // k-major LDS row stride for operands staged with transposing ds_write_b32
constexpr int S_MC = 260;   // ... for operands staged with ds_write_b128 (16-byte aligned rows)

template <int BK, bool KC>
__device__ __forceinline__ float4 g_load1(const float* __restrict__ src, int ld, int r0, int k0, int tid, int i) {
  constexpr int LPR = BK / 4, RPP = NT / LPR;
  if (KC) return *reinterpret_cast<const float4*>(src + (size_t)(r0 + tid / LPR + RPP * i) * ld + k0 + (tid % LPR) * 4);
  return *reinterpret_cast<const float4*>(src + (size_t)(k0 + (tid >> 6) + 4 * i) * ld + r0 + (tid & 63) * 4);
}
template <int BK, bool KC>
__device__ __forceinline__ void s_store1(const float4 reg, float* __restrict__ lds, int tid, int i) {
  constexpr int LPR = BK / 4, RPP = NT / LPR;
  if (KC) {
    const int r = tid / LPR + RPP * i, k = (tid % LPR) * 4;
    lds[(k + 0) * S_KC + r] = reg.x;
    lds[(k + 1) * S_KC + r] = reg.y;
    lds[(k + 2) * S_KC + r] = reg.z;
    lds[(k + 3) * S_KC + r] = reg.w;
  } else {
    *reinterpret_cast<float4*>(&lds[((tid >> 6) + 4 * i) * S_MC + (tid & 63) * 4]) = reg;
  }
}

template <int BK, bool A_KC, bool B_KC>
__global__ __launch_bounds__(NT, 1) void gemm_w128(const float* __restrict__ A, const float* __restrict__ B,
                                                   float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                                   int tiles_m, int tiles_n, int nostore) {
  constexpr int SA = A_KC ? S_KC : S_MC, SB = B_KC ? S_KC : S_MC;
  constexpr int STAGE = BK * (SA + SB);
  constexpr int NV = BK / 4, NKP = BK / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int ntile = tiles_m * tiles_n;
  const int nk = K / BK;
  // persistent: workgroup b sits on XCD b % 8; give every XCD a contiguous run of tiles per round so that the tiles sharing
  // an A row panel (consecutive tn) meet in one L2
  const int G = gridDim.x;
  const int bx = blockIdx.x & 7, bl = blockIdx.x >> 3;
  const int per_xcd = G >> 3;
  for (int base = 0; base < ntile; base += G) {
    const int t = base + bx * per_xcd + bl;
    if (t >= ntile) break;
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[NV], rb[NV];
    // prologue: tile 0 -> stage 0, tile 1 -> registers
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      ra[i] = g_load1<BK, A_KC>(A, lda, m0, 0, tid, i);
      rb[i] = g_load1<BK, B_KC>(B, ldb, n0, 0, tid, i);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      s_store1<BK, A_KC>(ra[i], lds, tid, i);
      s_store1<BK, B_KC>(rb[i], lds + BK * SA, tid, i);
    }
    {
      const int k1 = min(BK, K - BK);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        ra[i] = g_load1<BK, A_KC>(A, lda, m0, k1, tid, i);
        rb[i] = g_load1<BK, B_KC>(B, ldb, n0, k1, tid, i);
      }
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
      const float* As = lds + (kt & 1) * STAGE;
      const float* Bs = As + BK * SA;
      float* An = lds + ((kt + 1) & 1) * STAGE;
      float* Bn = An + BK * SA;
      // unconditional staging (no branches inside the k-tile body, so hipcc counts vmcnt exactly): past the end the last
      // k-tile is simply re-loaded / re-stored into the stage nobody reads any more
      const int k2 = min((kt + 2) * BK, K - BK);
      float a[2][4], b[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[0][i] = As[kh * SA + wm * 128 + i * 32 + l31];
        b[0][i] = Bs[kh * SB + wn * 128 + i * 32 + l31];
      }
#pragma unroll
      for (int kp = 0; kp < NKP; ++kp) {
        const int cur = kp & 1, nxt = cur ^ 1;
        if (kp + 1 < NKP) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            a[nxt][i] = As[(2 * (kp + 1) + kh) * SA + wm * 128 + i * 32 + l31];
            b[nxt][i] = Bs[(2 * (kp + 1) + kh) * SB + wn * 128 + i * 32 + l31];
          }
        }
        // staging work spread over the first k-pairs: piece kp of tile kt+1 goes from its registers to the other LDS stage and
        // the registers are re-loaded with the same piece of tile kt+2 right away (a whole k-tile of MFMAs to land)
        if (kp < NV) {
          s_store1<BK, A_KC>(ra[kp], An, tid, kp);
          s_store1<BK, B_KC>(rb[kp], Bn, tid, kp);
          ra[kp] = g_load1<BK, A_KC>(A, lda, m0, k2, tid, kp);
          rb[kp] = g_load1<BK, B_KC>(B, ldb, n0, k2, tid, kp);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = MFMA(a[cur][i], b[cur][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);      // keep every k-pair's staging work inside its own MFMA block
      }
      __syncthreads();
    }

    // epilogue: one row of four 32 x 32 blocks (32 x 128) at a time through the wave's private 16.5 KB LDS slice
    if (!nostore) {
      constexpr int ES = 132;
      float* w = lds + wave * (32 * ES);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2) + 4 * kh) * ES + j * 32 + l31] = acc[i][j][r];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const int rr = p * 2 + kh;
          const float4 v = *reinterpret_cast<const float4*>(w + rr * ES + l31 * 4);
          *reinterpret_cast<float4*>(C + (size_t)(m0 + wm * 128 + i * 32 + rr) * ldc + n0 + wn * 128 + l31 * 4) = v;
        }
      }
    }
    __syncthreads();
  }
}

struct Shape { int M, N, K; };

template <typename F>
static double time_ms(F&& f, int iters) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 2; ++i) f();
  hipEventRecord(s);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  hipEventDestroy(s); hipEventDestroy(e);
  return ms / iters;
}

static double max_abs_diff(const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  double m = 0;
  for (size_t i = 0; i < n; ++i) { double d = fabs((double)ha[i] - hb[i]); if (d > m) m = d; }
  return m;
}

template <int BK, bool A_KC, bool B_KC>
static void launch(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int nostore, int G = 256) {
  constexpr int SA = A_KC ? S_KC : S_MC, SB = B_KC ? S_KC : S_MC;
  const int lds_bytes = std::max(2 * BK * (SA + SB), 4 * 32 * 132) * 4;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute((const void*)gemm_w128<BK, A_KC, B_KC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    once = true;
  }
  const int tm = M / TM, tn = N / TN;
  const int g = std::min(G, ((tm * tn + 7) / 8) * 8);
  hipLaunchKernelGGL((gemm_w128<BK, A_KC, B_KC>), dim3(g), dim3(NT), lds_bytes, 0, A, B, C, M, N, K, lda, ldb, N, tm, tn, nostore);
}

int main() {
  const Shape shapes[] = {{65536, 1024, 256}, {65536, 256, 1024}, {65536, 768, 256}, {65536, 256, 256},
                          {65536, 512, 992}, {49152, 1024, 992}, {49152, 512, 512}, {294912, 256, 256}, {4096, 4096, 4096}};
  const size_t maxA = (size_t)294912 * 1024, maxC = (size_t)294912 * 1024;
  float *A, *B, *C, *C2;
  if (hipMalloc(&A, maxA * 4) || hipMalloc(&B, 4096 * 4096 * 4) || hipMalloc(&C, maxC * 4) || hipMalloc(&C2, maxC * 4)) return 1;
  {
    std::vector<float> h(1 << 24);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23); }
    for (size_t o = 0; o < maxA; o += h.size()) hipMemcpy(A + o, h.data(), (o + h.size() <= maxA ? h.size() : maxA - o) * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data() + 77, (size_t)4096 * 4096 * 4 - 400, hipMemcpyHostToDevice);
  }
  const char* names[] = {"shipped fwd", "w128 bk16", "w128 bk32", "bk32 nostore", "shipped dX", "w128dX bk16", "w128dX bk32"};
  printf("%-22s", "shape (M,N,K)");
  for (auto n : names) printf(" %12s", n);
  printf("\n");
  for (const Shape& s : shapes) {
    const double fl = 2.0 * s.M * s.N * s.K;
    const int K32 = s.K / 32 * 32;        // the lab kernels take whole k-tiles only (992 = 31 x 32)
    const double fl32 = 2.0 * s.M * s.N * K32;
    const int NVR = 7, R = 5;
    std::vector<std::vector<double>> t(NVR);
    for (int r = 0; r < R; ++r) {
      t[0].push_back(time_ms([&] { hoisdf_linear_fwd(A, s.K, B, s.K, nullptr, C, s.N, s.M, s.N, s.K, 0, 0.f, 0, nullptr, nullptr); }, 5));
      t[1].push_back(time_ms([&] { launch<16, true, true>(A, B, C2, s.M, s.N, K32, s.K, s.K, 0); }, 5));
      t[2].push_back(time_ms([&] { launch<32, true, true>(A, B, C2, s.M, s.N, K32, s.K, s.K, 0); }, 5));
      t[3].push_back(time_ms([&] { launch<32, true, true>(A, B, C2, s.M, s.N, K32, s.K, s.K, 1); }, 5));
      // grad-input orientation: dx[M][N] = dy[M][K] . W[K][N]   (contraction over K, B is [k][n] n-contiguous)
      t[4].push_back(time_ms([&] { hoisdf_linear_bwd_input(A, s.K, nullptr, 0.f, B, s.N, C, s.N, s.M, s.K, s.N, 0, nullptr); }, 5));
      t[5].push_back(time_ms([&] { launch<16, true, false>(A, B, C2, s.M, s.N, K32, s.K, s.N, 0); }, 5));
      t[6].push_back(time_ms([&] { launch<32, true, false>(A, B, C2, s.M, s.N, K32, s.K, s.N, 0); }, 5));
    }
    // correctness: same k order as the shipped kernel -> bitwise equal when K is a whole number of k-tiles
    double e1 = -1, e2 = -1, e3 = -1;
    if (K32 == s.K) {
      hoisdf_linear_fwd(A, s.K, B, s.K, nullptr, C, s.N, s.M, s.N, s.K, 0, 0.f, 0, nullptr, nullptr);
      hipMemset(C2, 0, (size_t)s.M * s.N * 4);
      launch<16, true, true>(A, B, C2, s.M, s.N, s.K, s.K, s.K, 0);
      e1 = max_abs_diff(C, C2, (size_t)s.M * s.N);
      hipMemset(C2, 0, (size_t)s.M * s.N * 4);
      launch<32, true, true>(A, B, C2, s.M, s.N, s.K, s.K, s.K, 0);
      e2 = max_abs_diff(C, C2, (size_t)s.M * s.N);
      hoisdf_linear_bwd_input(A, s.K, nullptr, 0.f, B, s.N, C, s.N, s.M, s.K, s.N, 0, nullptr);
      hipMemset(C2, 0, (size_t)s.M * s.N * 4);
      launch<32, true, false>(A, B, C2, s.M, s.N, s.K, s.K, s.N, 0);
      e3 = max_abs_diff(C, C2, (size_t)s.M * s.N);
    }
    char nm[64];
    snprintf(nm, sizeof nm, "(%d,%d,%d)", s.M, s.N, s.K);
    printf("%-22s", nm);
    for (int v = 0; v < NVR; ++v) {
      std::sort(t[v].begin(), t[v].end());
      printf(" %9.1f TF", ((v == 0 || v == 4) ? fl : fl32) / t[v][R / 2] / 1e9);
    }
    printf("   maxdiff %.1e %.1e %.1e\n", e1, e2, e3);
    fflush(stdout);
  }
  return 0;
}
