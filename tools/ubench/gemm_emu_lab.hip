// GEMM lab (round 3): fp32 GEMM EMULATED on the bf16 MFMA pipe.  Every f32 operand is split EXACTLY into three bf16 pieces
// (x = x0 + x1 + x2: 8 + 8 + 8 significand bits, bf16 has the f32 exponent range, so no scaling is needed) and the product
// is accumulated in f32 from six v_mfma_f32_32x32x16_bf16 products (x0y0 + x0y1 + x1y0 + x1y1 + x0y2 + x2y0; the three
// dropped terms are <= 2^-24 of the product).  Error vs fp64 is that of an f32 GEMM (tools: see the accuracy table this lab
// prints); the bf16 pipe runs 16 x the f32 MFMA rate, so six products leave 2.67 x.
//   y[M][N] = x[M][K] . W[N][K]^T.  x is read as f32 and split on its way into LDS; W is pre-split once into a "slab image"
//   (exactly the LDS image of a [TN rows][16 k] slab of the three planes, contiguous in memory).
// Tile TM x TN = 256 x 128 (4 waves as 2 x 2, wave tile 128 x 64 = 4 x 2 MFMA blocks, 128 accumulators), two workgroups / CU.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_emu_lab.hip -I../../include -L../../hoisdf_amd -lhoisdf_hip \
//        -Wl,-rpath,'$ORIGIN/../../hoisdf_amd' -o gemm_emu_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "hoisdf.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define MFB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int TM = 256, KS = 16, NT = 256;

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int nx = 8;
  int xcd = bid % nx, loc = bid / nx;
  int q = nblk / nx, r = nblk % nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}

// exact three-way split of 8 floats into 3 x 8 bf16
#define SPLIT1(x, i)                                   \
  do {                                                 \
    const __bf16 a_ = (__bf16)(x);                     \
    const float r1_ = (x) - (float)a_;                 \
    const __bf16 b_ = (__bf16)r1_;                     \
    const float r2_ = r1_ - (float)b_;                 \
    p0[i] = a_; p1[i] = b_; p2[i] = (__bf16)r2_;       \
  } while (0)
__device__ __forceinline__ void split3x8(const float4 u, const float4 w, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
  SPLIT1(u.x, 0); SPLIT1(u.y, 1); SPLIT1(u.z, 2); SPLIT1(u.w, 3);
  SPLIT1(w.x, 4); SPLIT1(w.y, 5); SPLIT1(w.z, 6); SPLIT1(w.w, 7);
}

__device__ __forceinline__ void split3x4(const float4 u, bf16x4& p0, bf16x4& p1, bf16x4& p2) {
  SPLIT1(u.x, 0); SPLIT1(u.y, 1); SPLIT1(u.z, 2); SPLIT1(u.w, 3);
}

// ---- weight image: for column tile tn (TN rows of W), slab s (16 k), plane p, chunk c (8 k), row r: 16 bytes at
// ((((tn * nslab + s) * 3 + p) * 2 + c) * TN + r) * 16
template <int TN>
__global__ void prep_weight(const float* __restrict__ W, int ldw, int N, int K, int nslab, u32x4* __restrict__ img) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (tn, s, c, r)
  const int r = idx % TN;
  const int c = (idx / TN) % 2;
  const int s = (idx / (2 * TN)) % nslab;
  const int tn = idx / ((long)2 * TN * nslab);
  const int n = tn * TN + r;
  if (tn * TN >= ((N + TN - 1) / TN) * TN) return;
  float4 u, w;
  {
    const int k = s * KS + c * 8;
    const float* src = W + (size_t)min(n, N - 1) * ldw;
    auto g = [&](int kk) { return (n < N && kk < K) ? src[kk] : 0.f; };
    u = make_float4(g(k), g(k + 1), g(k + 2), g(k + 3));
    w = make_float4(g(k + 4), g(k + 5), g(k + 6), g(k + 7));
  }
  bf16x8 p0, p1, p2;
  split3x8(u, w, p0, p1, p2);
  const size_t base = ((size_t)(tn * nslab + s) * 3) * 2 * TN;
  img[base + (0 * 2 + c) * TN + r] = __builtin_bit_cast(u32x4, p0);
  img[base + (1 * 2 + c) * TN + r] = __builtin_bit_cast(u32x4, p1);
  img[base + (2 * 2 + c) * TN + r] = __builtin_bit_cast(u32x4, p2);
}

template <int TN, int OCC, int VAR>
__global__ __launch_bounds__(NT, OCC) void emu_kc(const float* __restrict__ A, int lda, const u32x4* __restrict__ Bimg,
                                                  float* __restrict__ C, int ldc, int M, int N, int K, int tiles_m, int tiles_n,
                                                  int nostore) {
  constexpr int WN = TN / 2;                 // wave tile columns (2 waves along n)
  constexpr int NJ = WN / 32;                // B blocks per wave
  constexpr int RSA = VAR >= 4 ? TM + 4 : TM;   // 16-byte units per (plane, chunk) region of A (VAR >= 4: +64 B skew)
  constexpr int A_U4 = 3 * 2 * RSA;          // uint4 per stage, A planes
  constexpr int B_U4 = 3 * 2 * TN;
  constexpr int STAGE_U4 = A_U4 + B_U4;
  constexpr int NB = B_U4 / NT;              // 16-byte pieces of the B image per thread per slab
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = t / tiles_n, tn = t - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int nslab = (K + KS - 1) / KS;

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: thread = row of the A tile (16 consecutive floats of that row per slab); B image pieces tid + 256 q
  const float* arow = A + (size_t)min(m0 + tid, M - 1) * lda;
  const u32x4* bsrc = Bimg + (size_t)tn * nslab * B_U4 + tid;
  float4 ra[4];
  u32x4 rb[NB];
#define LOAD_SLAB(sl)                                                                                          \
  do {                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const float4*>(arow + (sl) * KS + q * 4); \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) rb[q] = bsrc[(size_t)(sl) * B_U4 + q * NT];                  \
  } while (0)
#define STORE_SLAB(st)                                                                                         \
  do {                                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                            \
      bf16x8 p0, p1, p2;                                                                                       \
      split3x8(ra[2 * c], ra[2 * c + 1], p0, p1, p2);                                                          \
      (st)[(0 * 2 + c) * TM + tid] = __builtin_bit_cast(u32x4, p0);                                            \
      (st)[(1 * 2 + c) * TM + tid] = __builtin_bit_cast(u32x4, p1);                                            \
      (st)[(2 * 2 + c) * TM + tid] = __builtin_bit_cast(u32x4, p2);                                            \
    }                                                                                                          \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) (st)[A_U4 + tid + q * NT] = rb[q];                           \
  } while (0)

#define LOAD_SLAB4(sl)                                                                                         \
  do {                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const float4*>(arow4[q] + (sl) * KS); \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) rb[q] = bsrc[(size_t)(sl) * B_U4 + q * NT];                  \
  } while (0)
#define STORE_SLAB4(st)                                                                                        \
  do {                                                                                                         \
    u32x2* s2_ = reinterpret_cast<u32x2*>(st);                                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                            \
      bf16x4 p0, p1, p2;                                                                                       \
      split3x4(ra[q], p0, p1, p2);                                                                             \
      const int o_ = (((tid & 3) >> 1) * RSA + (tid >> 2) + 64 * q) * 2 + (tid & 1);                           \
      s2_[o_ + 0 * 2 * RSA * 2] = __builtin_bit_cast(u32x2, p0);                                               \
      s2_[o_ + 1 * 2 * RSA * 2] = __builtin_bit_cast(u32x2, p1);                                               \
      s2_[o_ + 2 * 2 * RSA * 2] = __builtin_bit_cast(u32x2, p2);                                               \
    }                                                                                                          \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) (st)[A_U4 + tid + q * NT] = rb[q];                           \
  } while (0)
  const float* arow4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) arow4[q] = A + (size_t)min(m0 + (tid >> 2) + 64 * q, M - 1) * lda + (tid & 3) * 4;

  const int last = nslab - 1;
  if (VAR >= 4) {
    LOAD_SLAB4(0);
    STORE_SLAB4(lds);
    LOAD_SLAB4(min(1, last));
    __syncthreads();
  } else {
  LOAD_SLAB(0);
  STORE_SLAB(lds);
  LOAD_SLAB(min(1, last));
  __syncthreads();
  }

  for (int s = 0; s < nslab; ++s) {
    const u32x4* st = lds + (s & 1) * STAGE_U4;
    u32x4* nx = lds + ((s + 1) & 1) * STAGE_U4;
    const u32x4* sa = st + wm * 128 + l31;
    const u32x4* sb = st + A_U4 + wn * WN + l31;
    bf16x8 b0[NJ], b1[NJ], b2[NJ], a[4];
#define RD_B(dst, p) _Pragma("unroll") for (int j = 0; j < NJ; ++j) dst[j] = __builtin_bit_cast(bf16x8, sb[((p) * 2 + kh) * TN + j * 32])
#define RD_A(p) _Pragma("unroll") for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, sa[((p) * 2 + kh) * RSA + i * 32])
#define MM1(bx) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFB(a[i], bx[j], acc[i][j])
    if (VAR == 0) {
      RD_B(b0, 0); RD_A(2);
      STORE_SLAB(nx);
      LOAD_SLAB(min(s + 2, last));
      __builtin_amdgcn_sched_barrier(0);
      MM1(b0);
      RD_B(b1, 1); RD_A(1);
      MM1(b1); MM1(b0);
      RD_B(b2, 2); RD_A(0);
      MM1(b2); MM1(b1); MM1(b0);
    } else if (VAR == 1) {
      // long phase first: 24 MFMAs queue up right behind the barrier, the conversion of the next slab follows them
      RD_B(b0, 0); RD_A(0); RD_B(b1, 1); RD_B(b2, 2);
      MM1(b0); MM1(b1); MM1(b2);
      __builtin_amdgcn_sched_barrier(0);
      RD_A(1);
      STORE_SLAB(nx);
      LOAD_SLAB(min(s + 2, last));
      __builtin_amdgcn_sched_barrier(0);
      MM1(b0); MM1(b1);
      RD_A(2);
      MM1(b0);
    } else if (VAR == 2) {
      // as 1, but only the global loads are pinned: everything else may move across the marks
      RD_B(b0, 0); RD_A(0); RD_B(b1, 1); RD_B(b2, 2);
      MM1(b0); MM1(b1); MM1(b2);
      RD_A(1);
      STORE_SLAB(nx);
      __builtin_amdgcn_sched_barrier(0x78F);
      LOAD_SLAB(min(s + 2, last));
      __builtin_amdgcn_sched_barrier(0x78F);
      MM1(b0); MM1(b1);
      RD_A(2);
      MM1(b0);
    } else if (VAR == 4) {
      // VAR 0 with the A tile staged 4 lanes per row (64 contiguous bytes per row and instruction, 16 rows per wave
      // instruction) instead of one row per lane (64 different cache lines per instruction)
      RD_B(b0, 0); RD_A(2);
      STORE_SLAB4(nx);
      LOAD_SLAB4(min(s + 2, last));
      __builtin_amdgcn_sched_barrier(0);
      MM1(b0);
      RD_B(b1, 1); RD_A(1);
      MM1(b1); MM1(b0);
      RD_B(b2, 2); RD_A(0);
      MM1(b2); MM1(b1); MM1(b0);
    } else if (VAR == 5) {
      // ablation of VAR 0: no f32 -> bf16x3 conversion (raw bits parked as planes; timing only, results are garbage)
      RD_B(b0, 0); RD_A(2);
#pragma unroll
      for (int q = 0; q < 4; ++q) nx[q * TM + tid] = __builtin_bit_cast(u32x4, ra[q]);
      nx[4 * TM + tid] = __builtin_bit_cast(u32x4, ra[0]);
      nx[5 * TM + tid] = __builtin_bit_cast(u32x4, ra[1]);
#pragma unroll
      for (int q = 0; q < NB; ++q) nx[A_U4 + tid + q * NT] = rb[q];
      LOAD_SLAB(min(s + 2, last));
      __builtin_amdgcn_sched_barrier(0);
      MM1(b0);
      RD_B(b1, 1); RD_A(1);
      MM1(b1); MM1(b0);
      RD_B(b2, 2); RD_A(0);
      MM1(b2); MM1(b1); MM1(b0);
    } else if (VAR == 6) {
      // ablation of VAR 0: no global loads and no LDS stores at all (the MFMA + fragment-read pipeline alone)
      RD_B(b0, 0); RD_A(2);
      MM1(b0);
      RD_B(b1, 1); RD_A(1);
      MM1(b1); MM1(b0);
      RD_B(b2, 2); RD_A(0);
      MM1(b2); MM1(b1); MM1(b0);
    } else if (VAR == 7) {
      // ablation: MFMAs only (fragments read once per slab from plane 0, 48 MFMAs on them), no staging
      RD_B(b0, 0); RD_A(0);
      MM1(b0); MM1(b0); MM1(b0); MM1(b0); MM1(b0); MM1(b0);
    } else {
      // no pins at all (the compiler's own schedule)
      RD_B(b0, 0); RD_A(0); RD_B(b1, 1); RD_B(b2, 2);
      STORE_SLAB(nx);
      LOAD_SLAB(min(s + 2, last));
      MM1(b0); MM1(b1); MM1(b2);
      RD_A(1);
      MM1(b0); MM1(b1);
      RD_A(2);
      MM1(b0);
    }
    __syncthreads();
  }

  if (!nostore) {
    constexpr int ES = WN + 4;
    float* w = reinterpret_cast<float*>(lds) + wave * (32 * ES);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2) + 4 * kh) * ES + j * 32 + l31] = acc[i][j][r];
      constexpr int LPR = WN / 4;            // lanes per row (float4 each)
      constexpr int RPI = 64 / LPR;          // rows per wave-instruction
#pragma unroll
      for (int p = 0; p < 32 / RPI; ++p) {
        const int rr = p * RPI + lane / LPR, cc = (lane % LPR) * 4;
        const float4 v = *reinterpret_cast<const float4*>(w + rr * ES + cc);
        const int row = m0 + wm * 128 + i * 32 + rr;
        if (row < M) *reinterpret_cast<float4*>(C + (size_t)row * ldc + n0 + wn * WN + cc) = v;
      }
    }
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

template <int TN, int OCC, int VAR = 0>
static void launch(const float* A, int lda, const u32x4* img, float* C, int M, int N, int K, int nostore) {
  const int lds_bytes = 2 * (3 * 2 * (VAR >= 4 ? TM + 4 : TM) + 3 * 2 * TN) * 16;
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute((const void*)emu_kc<TN, OCC, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    once = true;
  }
  const int tm = (M + TM - 1) / TM, tn = N / TN;
  hipLaunchKernelGGL((emu_kc<TN, OCC, VAR>), dim3(tm * tn), dim3(NT), lds_bytes, 0, A, lda, img, C, N, M, N, K, tm, tn, nostore);
}

int main(int argc, char** argv) {
  const Shape shapes[] = {{65536, 1024, 256}, {65536, 256, 1024}, {65536, 768, 256}, {65536, 256, 256},
                          {65536, 512, 992}, {49152, 1024, 992}, {49152, 512, 512}, {294912, 256, 256}, {4096, 4096, 4096}};
  const size_t maxA = (size_t)294912 * 1024, maxC = (size_t)294912 * 1024;
  float *A, *B, *C, *C2;
  u32x4* img;
  if (hipMalloc(&A, maxA * 4) || hipMalloc(&B, (size_t)4096 * 4096 * 4) || hipMalloc(&C, maxC * 4) || hipMalloc(&C2, maxC * 4) ||
      hipMalloc(&img, (size_t)4096 * 4096 * 6 + (1 << 20))) return 1;
  {
    std::vector<float> h(1 << 24);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23); }
    for (size_t o = 0; o < maxA; o += h.size()) hipMemcpy(A + o, h.data(), (o + h.size() <= maxA ? h.size() : maxA - o) * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data() + 77, (size_t)4096 * 4096 * 4 - 400, hipMemcpyHostToDevice);
  }
  if (argc > 2) {           // PMC driver: gemm_emu_lab <variant 0 = emu128, 1 = emu256, 2 = shipped f32> <shape index>
    const int v = atoi(argv[1]);
    const Shape s = shapes[atoi(argv[2])];
    const int nslab = (s.K + KS - 1) / KS;
    if (v == 0) hipLaunchKernelGGL(prep_weight<128>, dim3((unsigned)(((long)(s.N / 128) * nslab * 2 * 128 + 255) / 256)), dim3(256), 0, 0, B, s.K, s.N, s.K, nslab, img);
    if (v == 1) hipLaunchKernelGGL(prep_weight<256>, dim3((unsigned)(((long)(s.N / 256) * nslab * 2 * 256 + 255) / 256)), dim3(256), 0, 0, B, s.K, s.N, s.K, nslab, img);
    for (int i = 0; i < 4; ++i) {
      if (v == 0) launch<128, 2, 4>(A, s.K, img, C2, s.M, s.N, s.K, 0);
      else if (v == 1) launch<256, 1, 1>(A, s.K, img, C2, s.M, s.N, s.K, 0);
      else hoisdf_linear_fwd(A, s.K, B, s.K, nullptr, C, s.N, s.M, s.N, s.K, 0, 0.f, 0, nullptr, nullptr);
    }
    hipDeviceSynchronize();
    return 0;
  }
  printf("%-22s %12s %12s %12s %12s %12s [then emu128 v2, v4]  accuracy vs fp64 on 64 sampled rows (max |err| / sum|a||b|): f32 kernel, emulated\n", "shape (M,N,K)",
         "shipped f32", "emu128 v0", "v5 noconv", "v6 nostage", "v7 mfma");
  for (const Shape& s : shapes) {
    const double fl = 2.0 * s.M * s.N * s.K;
    const int nslab = (s.K + KS - 1) / KS;
    const int R = 5;
    std::vector<std::vector<double>> t(7);
    for (int r = 0; r < R; ++r) {
      t[0].push_back(time_ms([&] { hoisdf_linear_fwd(A, s.K, B, s.K, nullptr, C, s.N, s.M, s.N, s.K, 0, 0.f, 0, nullptr, nullptr); }, 5));
      hipLaunchKernelGGL(prep_weight<128>, dim3((unsigned)(((long)(s.N / 128) * nslab * 2 * 128 + 255) / 256)), dim3(256), 0, 0, B, s.K, s.N, s.K, nslab, img);
      t[1].push_back(time_ms([&] { launch<128, 2, 0>(A, s.K, img, C2, s.M, s.N, s.K, 0); }, 5));
      t[2].push_back(time_ms([&] { launch<128, 2, 5>(A, s.K, img, C2, s.M, s.N, s.K, 0); }, 5));
      t[5].push_back(time_ms([&] { launch<128, 2, 2>(A, s.K, img, C2, s.M, s.N, s.K, 0); }, 5));
      t[6].push_back(time_ms([&] { launch<128, 2, 4>(A, s.K, img, C2, s.M, s.N, s.K, 0); }, 5));
      hipLaunchKernelGGL(prep_weight<256>, dim3((unsigned)(((long)(s.N / 256) * nslab * 2 * 256 + 255) / 256)), dim3(256), 0, 0, B, s.K, s.N, s.K, nslab, img);
      t[3].push_back(time_ms([&] { launch<128, 2, 6>(A, s.K, img, C2, s.M, s.N, s.K, 0); }, 5));
      t[4].push_back(time_ms([&] { launch<128, 2, 7>(A, s.K, img, C2, s.M, s.N, s.K, 0); }, 5));
    }
    // accuracy: 64 rows spread over M against a host fp64 dot product
    hoisdf_linear_fwd(A, s.K, B, s.K, nullptr, C, s.N, s.M, s.N, s.K, 0, 0.f, 0, nullptr, nullptr);
    hipLaunchKernelGGL(prep_weight<128>, dim3((unsigned)(((long)(s.N / 128) * nslab * 2 * 128 + 255) / 256)), dim3(256), 0, 0, B, s.K, s.N, s.K, nslab, img);
    hipMemset(C2, 0, (size_t)s.M * s.N * 4);
    launch<128, 2, 4>(A, s.K, img, C2, s.M, s.N, s.K, 0);
    hipDeviceSynchronize();
    std::vector<float> hb((size_t)s.N * s.K), ha(s.K), hc(s.N), hc2(s.N);
    hipMemcpy(hb.data(), B, hb.size() * 4, hipMemcpyDeviceToHost);
    double e32 = 0, eem = 0;
    for (int q = 0; q < 64; ++q) {
      const size_t row = (size_t)q * (s.M / 64) + (q * 37) % (s.M / 64);
      hipMemcpy(ha.data(), A + row * s.K, s.K * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hc.data(), C + row * s.N, s.N * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hc2.data(), C2 + row * s.N, s.N * 4, hipMemcpyDeviceToHost);
      for (int n = 0; n < s.N; ++n) {
        double ref = 0, den = 0;
        for (int k = 0; k < s.K; ++k) { const double p = (double)ha[k] * hb[(size_t)n * s.K + k]; ref += p; den += fabs(p); }
        e32 = std::max(e32, fabs(hc[n] - ref) / den);
        eem = std::max(eem, fabs(hc2[n] - ref) / den);
      }
    }
    char nm[64];
    snprintf(nm, sizeof nm, "(%d,%d,%d)", s.M, s.N, s.K);
    printf("%-22s", nm);
    for (int v = 0; v < 7; ++v) { std::sort(t[v].begin(), t[v].end()); printf(" %9.1f TF", fl / t[v][R / 2] / 1e9); }
    printf("   %.2e %.2e\n", e32, eem);
    fflush(stdout);
  }
  return 0;
}
