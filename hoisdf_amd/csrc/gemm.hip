// Exact-fp32 GEMM family on the gfx950 f32 MFMA pipe (v_mfma_f32_32x32x2_f32, 157 TF peak,
// bit-for-bit an fmaf chain).  One template serves the three contractions a linear layer needs:
//   forward      y  = x . W^T      A[m][k] k-contiguous, B[n][k] k-contiguous
//   grad input   dx = dy . W       A[m][k] k-contiguous, B[k][n] n-contiguous
//   grad weight  dW = dy^T . x     A[k][m] m-contiguous, B[k][n] n-contiguous   (split-K)
// Block tile 128x128x16, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator
// VGPRs).  LDS holds both operands k-major ([k][m]) so the MFMA fragment read is always one
// conflict-free ds_read_b32 per operand per k-pair; only the global->LDS staging differs with
// the operand's memory orientation.  Register staging: the next k-tile's global loads are in
// flight during the 32 MFMAs of the current one, then ds_write + barrier into the single 16.5 KB LDS
// stage; 4 workgroups per CU (<= 128 VGPRs) hide that hand-over (a second LDS stage measured +-0).
// Block ids are remapped so that the tiles sharing an A row-panel sit on one XCD (shared L2).
//
// Fusions: bias + ReLU + dropout in the forward epilogue, which also emits a 1-bit/element sign map
// (wave ballot); in both backward contractions the ReLU/dropout backward (dy * [y > 0] / (1-p)) is
// applied from that bitmap while staging the dy operand (an L2-resident 1/32-size side input), so
// the pre-activation gradient is never written to HBM; the bias gradient (column sums of dy) is
// accumulated from the staged dy tiles of the grad-weight kernel.  Split-K (grad-weight always; forward /
// grad-input when the grid is only a few tiles) accumulates with float atomics into a zeroed output, every
// k-slice on one XCD; a caller workspace selects partial tiles + a deterministic reduce kernel instead.
#include <stdlib.h>

#include "common.h"

namespace hoisdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef HOISDF_GEMM_BK
#define HOISDF_GEMM_BK 16
#endif
constexpr int BM = 128, BN = 128, BK = HOISDF_GEMM_BK, NT = 256;
constexpr int NV = BK / 8;            // float4 per thread per operand tile
constexpr int KC_LPR = BK / 4;        // lanes per row for k-contiguous staging
constexpr int KC_RPP = NT / KC_LPR;   // rows per pass
constexpr int LDS_KC = BM + 1;   // k-contiguous source: transposing ds_write_b32, stride = 1 mod 32
constexpr int LDS_MC = BM + 4;   // m-contiguous source: ds_write_b128, 16-byte aligned rows

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const uint32_t* abits;   // optional 1-bit mask of A ([rows][ldbits] words along A's contiguous
                           // "column" index of the ORIGINAL [M][N] dy): A is used as A * bit * ascale
  uint32_t* bits_out;      // optional (forward): 1 bit per output element, set iff y > 0
  float* colsum;           // optional (grad-weight only): column sums of (masked) A
  int M, N, K;
  int lda, ldb, ldc, ldbits;
  int act;
  float drop_p, inv_keep, ascale;
  uint32_t thresh;
  uint64_t seed;
  int splitk, k_per_split, atomic_out, partial;
  long c_split_stride, colsum_split_stride;
  int tiles_m, tiles_n;
  int vecA, vecB, vecC, nofast;
  int beta;                // 1: C += result (grad-input accumulating into an existing gradient), non-atomic paths
  int occ;                 // workgroups per CU to run at (0 = the kernel's natural 4), see pick_occupancy
};

// Stage one 128 x 32 operand tile from global memory into registers (4 x float4 per thread).
// KC = true : element (r, k) at src[r*ld + k]   (k contiguous)
// KC = false: element (r, k) at src[k*ld + r]   (r contiguous)
template <bool KC>
__device__ __forceinline__ void stage_load(float4 (&reg)[NV], const float* __restrict__ src, int ld,
                                           int r0, int R, int k0, int kend, int vec, int tid) {
  // Fast path (wave-uniform test): 16-byte aligned rows and the whole k-range of the tile in bounds.
  // k-contiguous: out-of-range rows are clamped to the last valid one (their products only reach
  // output rows that are never stored); r-contiguous: whole 4-wide chunks are predicated.
  const bool fast = vec && (k0 + BK <= kend) && (KC ? (R > 0) : ((R & 3) == 0 && R >= 4));
  if (fast) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (KC) {
        const int r = min(r0 + (tid / KC_LPR) + KC_RPP * i, R - 1);
        reg[i] = *reinterpret_cast<const float4*>(src + (size_t)r * ld + k0 + (tid % KC_LPR) * 4);
      } else {
        // R is a multiple of 4, so a 4-wide chunk is entirely inside or entirely outside the tile's valid rows
        const int r = r0 + (tid & 31) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)(k0 + (tid >> 5) + 8 * i) * ld + min(r, R - 4));
        reg[i] = r < R ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      int r = r0 + (tid / KC_LPR) + KC_RPP * i;
      int k = k0 + (tid % KC_LPR) * 4;
      if (r < R) {
        const float* p = src + (size_t)r * ld + k;
        if (vec && k + 3 < kend) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (k + 0 < kend) v.x = p[0];
          if (k + 1 < kend) v.y = p[1];
          if (k + 2 < kend) v.z = p[2];
          if (k + 3 < kend) v.w = p[3];
        }
      }
    } else {
      int k = k0 + (tid >> 5) + 8 * i;
      int r = r0 + (tid & 31) * 4;
      if (k < kend) {
        const float* p = src + (size_t)k * ld + r;
        if (vec && r + 3 < R) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (r + 0 < R) v.x = p[0];
          if (r + 1 < R) v.y = p[1];
          if (r + 2 < R) v.z = p[2];
          if (r + 3 < R) v.w = p[3];
        }
      }
    }
    reg[i] = v;
  }
}

// 4 mask bits per staged float4, from the forward's ReLU/dropout sign bitmap.
// KC (dy as [m][n], tile rows = m, k = n): word (row, k0/32), nibble at (tid&7)*4.
// MC (dy as [k=m][i=n], tile "rows" r = n, k = m): word (m, n/32), nibble at (tid&7)*4.
template <bool KC>
__device__ __forceinline__ void load_bits(uint32_t (&w)[NV], const uint32_t* __restrict__ bits, int ldbits, int r0,
                                          int R, int k0, int kend, int tid) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    uint32_t v = 0u;
    int sh;
    if (KC) {
      const int r = r0 + (tid / KC_LPR) + KC_RPP * i;
      const int k = k0 + (tid % KC_LPR) * 4;
      sh = k & 31;
      if (r < R && k < kend) v = bits[(size_t)r * ldbits + (k >> 5)];
    } else {
      const int k = k0 + (tid >> 5) + 8 * i;
      const int r = r0 + (tid & 31) * 4;
      sh = r & 31;
      if (k < kend && r < R) v = bits[(size_t)k * ldbits + (r >> 5)];
    }
    w[i] = (v >> sh) & 0xFu;
  }
}
__device__ __forceinline__ void apply_bits(float4 (&a)[NV], const uint32_t (&w)[NV], float sc) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    a[i].x = (w[i] & 1u) ? a[i].x * sc : 0.f;
    a[i].y = (w[i] & 2u) ? a[i].y * sc : 0.f;
    a[i].z = (w[i] & 4u) ? a[i].z * sc : 0.f;
    a[i].w = (w[i] & 8u) ? a[i].w * sc : 0.f;
  }
}

template <bool KC>
__device__ __forceinline__ void stage_store(const float4 (&reg)[NV], float* __restrict__ lds, int tid) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (KC) {
      int r = (tid / KC_LPR) + KC_RPP * i;
      int k = (tid % KC_LPR) * 4;
      lds[(k + 0) * LDS_KC + r] = reg[i].x;
      lds[(k + 1) * LDS_KC + r] = reg[i].y;
      lds[(k + 2) * LDS_KC + r] = reg[i].z;
      lds[(k + 3) * LDS_KC + r] = reg[i].w;
    } else {
      int k = (tid >> 5) + 8 * i;
      int r = (tid & 31) * 4;
      *reinterpret_cast<float4*>(&lds[k * LDS_MC + r]) = reg[i];
    }
  }
}

// ---- interior-tile fast path: the tile lies fully inside both operands, every k-tile is full and all rows are 16-byte
// aligned, so the staging is a fixed per-thread pointer that advances by one k-tile - no bounds tests, clamps or
// fast/slow dispatch inside the main loop.  Used where it measures faster (see launch_gemm).
template <bool KC>
__device__ __forceinline__ const float* fast_ptr(const float* __restrict__ src, int ld, int r0, int k0, int tid) {
  return KC ? src + (size_t)(r0 + tid / KC_LPR) * ld + k0 + (tid % KC_LPR) * 4
            : src + (size_t)(k0 + (tid >> 5)) * ld + r0 + (tid & 31) * 4;
}
template <bool KC>
__device__ __forceinline__ void fast_load(float4 (&reg)[NV], const float* __restrict__ p, int ld) {
#pragma unroll
  for (int i = 0; i < NV; ++i)
    reg[i] = *reinterpret_cast<const float4*>(p + (size_t)(KC ? KC_RPP * i : 8 * i) * ld);
}
template <bool KC>
__device__ __forceinline__ void fast_bits(uint32_t (&w)[NV], const uint32_t* __restrict__ bits, int ldbits, int r0, int k0,
                                          int tid) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (KC) {
      const int r = r0 + (tid / KC_LPR) + KC_RPP * i, k = k0 + (tid % KC_LPR) * 4;
      w[i] = (bits[(size_t)r * ldbits + (k >> 5)] >> (k & 31)) & 0xFu;
    } else {
      const int k = k0 + (tid >> 5) + 8 * i, r = r0 + (tid & 31) * 4;
      w[i] = (bits[(size_t)k * ldbits + (r >> 5)] >> (r & 31)) & 0xFu;
    }
  }
}

// The k-loop of one output tile.  FAST = interior tile (see above); otherwise the guarded staging.
template <bool A_KC, bool B_KC, bool MASK, bool FAST>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& g, float* __restrict__ As, float* __restrict__ Bs,
                                              f32x16 (&acc)[2][2], float4& csum, const bool do_colsum, const int m0,
                                              const int n0, const int kbeg, const int kend, const int nk, const int tid,
                                              const int lane, const int wm, const int wn) {
  constexpr int SA = A_KC ? LDS_KC : LDS_MC;
  constexpr int SB = B_KC ? LDS_KC : LDS_MC;
  float4 ra[NV], rb[NV];
  uint32_t rm[NV];
  const float* pa = FAST ? fast_ptr<A_KC>(g.A, g.lda, m0, kbeg, tid) : nullptr;
  const float* pb = FAST ? fast_ptr<B_KC>(g.B, g.ldb, n0, kbeg, tid) : nullptr;
  const size_t stepa = A_KC ? (size_t)BK : (size_t)BK * g.lda;
  const size_t stepb = B_KC ? (size_t)BK : (size_t)BK * g.ldb;
  if (nk > 0) {
    if (FAST) {
      fast_load<A_KC>(ra, pa, g.lda);
      if (MASK) fast_bits<A_KC>(rm, g.abits, g.ldbits, m0, kbeg, tid);
      fast_load<B_KC>(rb, pb, g.ldb);
    } else {
      stage_load<A_KC>(ra, g.A, g.lda, m0, g.M, kbeg, kend, g.vecA, tid);
      if (MASK) load_bits<A_KC>(rm, g.abits, g.ldbits, m0, g.M, kbeg, kend, tid);
      stage_load<B_KC>(rb, g.B, g.ldb, n0, g.N, kbeg, kend, g.vecB, tid);
    }
    if (MASK) apply_bits(ra, rm, g.ascale);
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < NV; ++i) { csum.x += ra[i].x; csum.y += ra[i].y; csum.z += ra[i].z; csum.w += ra[i].w; }
    }
    stage_store<A_KC>(ra, As, tid);
    stage_store<B_KC>(rb, Bs, tid);
  }
  __syncthreads();

  const int arow = wm * 64 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);
  const int khalf = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      if (FAST) {
        pa += stepa;
        pb += stepb;
        fast_load<A_KC>(ra, pa, g.lda);
        if (MASK) fast_bits<A_KC>(rm, g.abits, g.ldbits, m0, kbeg + (kt + 1) * BK, tid);
        fast_load<B_KC>(rb, pb, g.ldb);
      } else {
        stage_load<A_KC>(ra, g.A, g.lda, m0, g.M, kbeg + (kt + 1) * BK, kend, g.vecA, tid);
        if (MASK) load_bits<A_KC>(rm, g.abits, g.ldbits, m0, g.M, kbeg + (kt + 1) * BK, kend, tid);
        stage_load<B_KC>(rb, g.B, g.ldb, n0, g.N, kbeg + (kt + 1) * BK, kend, g.vecB, tid);
      }
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a0 = As[(kk + khalf) * SA + arow];
      float a1 = As[(kk + khalf) * SA + arow + 32];
      float b0 = Bs[(kk + khalf) * SB + brow];
      float b1 = Bs[(kk + khalf) * SB + brow + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();      // every wave is done reading this stage
    if (kt + 1 < nk) {
      if (MASK) apply_bits(ra, rm, g.ascale);
      if (do_colsum) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { csum.x += ra[i].x; csum.y += ra[i].y; csum.z += ra[i].z; csum.w += ra[i].w; }
      }
      stage_store<A_KC>(ra, As, tid);
      stage_store<B_KC>(rb, Bs, tid);
      __syncthreads();
    }
  }
}

template <bool A_KC, bool B_KC, bool MASK, bool ATOMIC>
__global__ __launch_bounds__(NT, BK == 16 ? 4 : (BK == 32 ? 3 : 2)) void gemm_f32_kernel(GemmArgs g) {
  constexpr int SA = A_KC ? LDS_KC : LDS_MC;
  constexpr int SB = B_KC ? LDS_KC : LDS_MC;
  // one small LDS stage (BK = 16: 16.5 KB, 8 staging VGPRs per operand) + register staging keeps the
  // kernel under 128 VGPRs: 4 workgroups per CU (4 waves / SIMD) hide the barrier + global-load
  // latency.  Measured on MI355X (tools/microbench.py, 65536x512x992 fwd / masked dX / masked dW, TF/s):
  //   BK 32 double-buffered, 2 WG/CU:  93 / 53 / 52   (PMC: MFMA pipe 65 % busy, 22-57 % of wave cycles parked)
  //   BK 32 single stage,    3 WG/CU: 101 / 85 / 72   BK 64, 2 WG/CU: 94 / 72 / 71
  //   BK 16 single stage,    4 WG/CU:  97 / 97 / 86   <- this build
  constexpr int STAGE_FLOATS = BK * SA + BK * SB;
  static_assert(STAGE_FLOATS >= 4 * 32 * 32, "the epilogue parks one 32x32 block per wave in the staging buffer");
  __shared__ __attribute__((aligned(16))) float lds[STAGE_FLOATS];
  float* As = lds;
  float* Bs = lds + BK * SA;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int ntile = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  int split, t;
  if (g.splitk > 1) {
    // split-K (grad-weight): every tile of one k-slice runs on the SAME XCD (block b -> XCD b % 8), so the
    // dy / x row band the slice streams through is fetched once into that XCD's L2 and shared by all of
    // the slice's output tiles (PMC before: 34 % L2 hit rate, 3.7x HBM over-fetch).
    split = (bid & 7) + 8 * (bid / (8 * ntile));
    t = (bid >> 3) % ntile;
    if (split >= g.splitk) return;
  } else {
    split = 0;
    t = xcd_remap(bid, ntile);
  }
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;
  const bool do_colsum = (!A_KC) && g.colsum != nullptr && tn == 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int khalf = lane >> 5;
  // block-uniform choice of the staging code: interior tiles (all of them for the step's aligned shapes) take the
  // branch-free loop
  const bool interior = !g.nofast && g.vecA && g.vecB && (m0 + BM <= g.M) && (n0 + BN <= g.N) && ((kend - kbeg) % BK == 0);
  if (interior)
    gemm_mainloop<A_KC, B_KC, MASK, true>(g, As, Bs, acc, csum, do_colsum, m0, n0, kbeg, kend, nk, tid, lane, wm, wn);
  else
    gemm_mainloop<A_KC, B_KC, MASK, false>(g, As, Bs, acc, csum, do_colsum, m0, n0, kbeg, kend, nk, tid, lane, wm, wn);

  // bias gradient: this block column (tn == 0) has seen every dy element of its (m-tile, k-slice)
  if (do_colsum) {
    float* red = lds;                       // all LDS readers are past the final barrier
    *reinterpret_cast<float4*>(&red[(tid >> 5) * 128 + (tid & 31) * 4]) = csum;
    __syncthreads();
    if (tid < 128) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += red[j * 128 + tid];
      const int col = m0 + tid;
      if (col < g.M) {
        if (g.partial) g.colsum[(size_t)split * g.colsum_split_stride + col] = s;
        else atomicAdd(&g.colsum[col], s);
      }
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Three straight-line passes (value, store, sign bits) so the 64 stores of a lane issue back to back.
  float* Cb = g.C + (size_t)split * g.c_split_stride;
  const int rbase = m0 + wm * 64 + 4 * khalf;
  const int cbase = n0 + wn * 64 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = cbase + j * 32;
    const float bv = (g.bias != nullptr && split == 0 && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bv;
        if (g.act == 1) v = fmaxf(v, 0.f);
        acc[i][j][r] = v;
      }
  }
  if (g.drop_p > 0.f) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
        const uint32_t rk = drop_rowkey(g.seed, (uint32_t)row);
        acc[i][0][r] *= drop_scale(rk, (uint32_t)cbase, g.thresh, g.inv_keep);
        acc[i][1][r] *= drop_scale(rk, (uint32_t)(cbase + 32), g.thresh, g.inv_keep);
      }
  }
  const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N);
  if (ATOMIC) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2), col = cbase + j * 32;
          if (full || (row < g.M && col < g.N)) atomicAdd(Cb + (size_t)row * g.ldc + col, acc[i][j][r]);
        }
  } else {
    if (A_KC && full && g.vecC) {
      // Full tile, 16-byte aligned rows: through LDS.  Each wave parks one 32x32 accumulator block at a time in its
      // private 4 KB slice of the (now idle) staging buffer and reads it back row-wise, so one global_store_dwordx4
      // covers 8 complete 128-byte row segments: 16 store instructions per lane instead of 64 dword stores of two
      // row segments each (the tile's store tail is issue-bound; gemm_lab: +2.5-3 %).  All waves are past the
      // main loop's final barrier; the slices are wave-private, so no further barrier is needed.
      float* w = lds + wave * (32 * 32);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2) + 4 * khalf) * 32 + (lane & 31)] = acc[i][j][r];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int rr = p * 8 + (lane >> 3), cc = (lane & 7) * 4;
            float4 v = *reinterpret_cast<const float4*>(w + rr * 32 + cc);
            float4* cp = reinterpret_cast<float4*>(Cb + (size_t)(m0 + wm * 64 + i * 32 + rr) * g.ldc + n0 + wn * 64 + j * 32 + cc);
            if (g.beta) {
              const float4 old = *cp;
              v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
            }
            *cp = v;
          }
        }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2), col = cbase + j * 32;
            if (full || (row < g.M && col < g.N)) {
              float* cp = Cb + (size_t)row * g.ldc + col;
              *cp = g.beta ? *cp + acc[i][j][r] : acc[i][j][r];
            }
          }
    }
    if (g.bits_out) {
      // lanes 0-31 hold 32 consecutive columns of one row, lanes 32-63 of the row 4 below: one ballot is two
      // mask words.  Each lane collects the two words (j = 0, 1) of "its" row of the wave's 64 x 64 sub-tile
      // and writes them once (2 stores per lane instead of 64 two-lane stores).
      uint32_t w0 = 0u, w1 = 0u;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = i * 32 + (r & 3) + 8 * (r >> 2);      // row (within the sub-tile) of lanes 0-31
          const unsigned long long m0 = __ballot(acc[i][0][r] > 0.f);
          const unsigned long long m1 = __ballot(acc[i][1][r] > 0.f);
          if (lane == rl) { w0 = (uint32_t)m0; w1 = (uint32_t)m1; }
          if (lane == rl + 4) { w0 = (uint32_t)(m0 >> 32); w1 = (uint32_t)(m1 >> 32); }
        }
      const int row = m0 + wm * 64 + lane;
      const int wcol = (n0 + wn * 64) >> 5;
      const int nvalid = g.N - (n0 + wn * 64);              // columns of this sub-tile inside the matrix
      if (row < g.M) {
        if (nvalid > 0) g.bits_out[(size_t)row * g.ldbits + wcol] = nvalid >= 32 ? w0 : (w0 & ((1u << nvalid) - 1u));
        if (nvalid > 32) g.bits_out[(size_t)row * g.ldbits + wcol + 1] = nvalid >= 64 ? w1 : (w1 & ((1u << (nvalid - 32)) - 1u));
      }
    }
  }
}

static int aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Workgroups per CU for a grid of `nwg` equal tiles on 256 CUs.  With the natural 4 per CU a grid of 1025..1536 tiles
// runs one full round and leaves a half-empty second one; at 3 per CU the same grid is two full rounds (768 slots) and
// 3 waves per SIMD still hide the hand-over (tools/mb_occ.py, forward: 1536 tiles +2...+9 %; 768 / 3072 / 4096 tiles
// +-1 %; 2 per CU never wins; the split-k grad-weight grids lose at 3).  HOISDF_GEMM_OCC={2,3,4} overrides (experiments).
static int pick_occupancy(int nwg, int splitk) {
  if (const char* e = getenv("HOISDF_GEMM_OCC")) {
    const int v = atoi(e);
    if (v >= 2 && v <= 4) return v;
  }
  return (splitk <= 1 && nwg > 1024 && nwg <= 1536) ? 3 : 4;
}

template <bool A_KC, bool B_KC>
static int launch_gemm(GemmArgs g, hipStream_t st) {
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, BN);
  // vector (16-byte) global loads need 16-byte aligned rows along the contiguous dimension
  // (a k-chunk that crosses the end of the contraction range falls back to guarded scalars)
  g.vecA = aligned16(g.A) && (g.lda % 4 == 0) && (A_KC ? (g.k_per_split % 4 == 0) : true);
  g.vecB = aligned16(g.B) && (g.ldb % 4 == 0) && (B_KC ? (g.k_per_split % 4 == 0) : true);
  g.vecC = aligned16(g.C) && (g.ldc % 4 == 0) && (g.c_split_stride % 4 == 0);
  // Measured A/B (tools/mb_ab.py, MI355X): the hoisted branch-free staging loop is +-0 on the forward, -5...-9 % on
  // grad-input and -3...-6 % on large grad-weight problems (the guarded loop's wave-uniform fast path schedules better
  // there), but +4...+7 % on the grad-weight problems with <= 4 output tiles - so only those take it.
  g.nofast = !(!A_KC && !B_KC && g.tiles_m * g.tiles_n <= 4);
  const int ntile = g.tiles_m * g.tiles_n;
  const int nwg = g.splitk > 1 ? ntile * 8 * cdiv(g.splitk, 8) : ntile;
  dim3 grid((unsigned)nwg);
  // workgroups per CU: the kernel's natural residency is 4 (<= 128 VGPRs, 16.5 KB LDS); asking for extra dynamic LDS
  // caps it at 3 or 2 when that removes a mostly-empty last round of tiles (pick_occupancy)
  const int occ = g.occ > 0 ? g.occ : pick_occupancy(nwg, g.splitk);
  const unsigned dyn = occ >= 4 ? 0u : (occ == 3 ? 25600u : 39936u);
  constexpr bool CAN_ATOMIC = true;   // grad-weight always; forward / grad-input when a small grid is split along k
  if (CAN_ATOMIC && g.atomic_out) {
    if (g.abits) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, CAN_ATOMIC>), grid, dim3(NT), dyn, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false, CAN_ATOMIC>), grid, dim3(NT), dyn, st, g);
  } else {
    if (g.abits) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, false>), grid, dim3(NT), dyn, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false, false>), grid, dim3(NT), dyn, st, g);
  }
  return check_launch("gemm_f32");
}

// out[i] = sum_s part[s*stride + i]   (float4 lanes; n4 = number of float4 elements)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, long stride, int splits,
                                                              float* __restrict__ out, long n) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  for (; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n) {
      float4 s = *reinterpret_cast<const float4*>(part + i);
      for (int k = 1; k < splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * stride + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(out + i) = s;
    } else {
      for (long j = i; j < n; ++j) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part[(size_t)k * stride + j];
        out[j] = s;
      }
    }
  }
}

__global__ void relu_dropout_bwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ dy,
                                        int lddy, float* __restrict__ dp, int ldd, long M, int N,
                                        float inv_keep) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = M * (long)N;
  for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long m = idx / N;
    int n = (int)(idx - m * N);
    float yv = y[(size_t)m * ldy + n];
    dp[(size_t)m * ldd + n] = yv > 0.f ? dy[(size_t)m * lddy + n] * inv_keep : 0.f;
  }
}

// Forward / grad-input with only a few output tiles (the 17-query decoder rows: M = 544): one workgroup per CU
// exposes the full global-load latency on every k-step (30-124 us for 0.07-0.3 GFLOP).  Split the contraction
// over more workgroups (>= 4 k-steps each, <= ~256 workgroups) and accumulate with atomics into a zeroed output.
static int plan_small_splitk(int tiles, int K, int& kper) {
  const int ksteps = cdiv(K, BK);
  if (deterministic_mode()) { kper = ksteps * BK; return 1; }      // no float atomics on the output
  int s = 256 / (tiles > 0 ? tiles : 1);
  if (s > ksteps / 4) s = ksteps / 4;
  if (tiles > 64 || s < 2) { kper = ksteps * BK; return 1; }
  kper = cdiv(ksteps, s) * BK;
  return cdiv(K, kper);
}
static int zero_rows(float* p, int ld, long rows, int cols, hipStream_t st) {
  const hipError_t e = hipMemset2DAsync(p, (size_t)ld * sizeof(float), 0, (size_t)cols * sizeof(float), (size_t)rows, st);
  HOISDF_REQUIRE(e == hipSuccess, HOISDF_ERR_LAUNCH, "gemm: zeroing the split-k output failed: %s", hipGetErrorString(e));
  return 0;
}

static void plan_splitk(long M, int N, int K, int& splitk, int& kper) {
  int tiles = cdiv(N, BM) * cdiv(K, BN);
  int ksteps = cdiv(M, BK);
  // fewer, longer k-slices when the output is only a few tiles: every slice adds its whole tile with
  // device-scope atomics, and below ~1 slice per CU-slot that traffic outweighs the occupancy
  // (MI355X sweep, tools/microbench.py: 256x256 213 -> 162 us, 768x256 378 -> 306 us, >= 16 tiles unchanged)
  const int target = tiles <= 4 ? 512 : (tiles <= 12 ? 768 : 1024);
  int want = tiles >= target ? 1 : cdiv(target, tiles);
  splitk = want < 1 ? 1 : want;
  if (splitk > ksteps / 4) splitk = ksteps / 4 > 0 ? ksteps / 4 : 1;   // >= 4 k-tiles per split
  kper = cdiv(ksteps, splitk) * BK;
  splitk = cdiv(M, kper);
}

}  // namespace hoisdf

using namespace hoisdf;

extern "C" int hoisdf_linear_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias,
                                 float* y, int ldy, long M, int N, int K, int act, float drop_p,
                                 uint64_t seed, uint32_t* relu_bits, void* stream) {
  HOISDF_REQUIRE(M == 0 || (x && W && y), HOISDF_ERR_INVALID, "linear_fwd: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, HOISDF_ERR_INVALID,
                 "linear_fwd: bad sizes M=%ld N=%d K=%d ldx=%d ldw=%d ldy=%d", M, N, K, ldx, ldw, ldy);
  HOISDF_REQUIRE(drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID, "linear_fwd: drop_p=%f", drop_p);
  HOISDF_REQUIRE(M < (1L << 31), HOISDF_ERR_INVALID, "linear_fwd: M too large");
  if (M == 0) return HOISDF_OK;
  GemmArgs g{};
  g.A = x; g.B = W; g.C = y; g.bias = bias;
  g.M = (int)M; g.N = N; g.K = K; g.lda = ldx; g.ldb = ldw; g.ldc = ldy;
  g.act = act; g.drop_p = drop_p; g.inv_keep = 1.f / (1.f - drop_p); g.thresh = drop_threshold(drop_p); g.seed = seed;
  g.splitk = 1; g.k_per_split = ((K + BK - 1) / BK) * BK; g.atomic_out = 0;
  g.bits_out = relu_bits; g.ldbits = (N + 31) / 32;
  // (round 6: off by default for the FORWARD - float atomics in an unfixed order made the regression heads' outputs differ in the last
  // bit between two identical runs, and with them sample 0's MANO outputs between two batches that differ in the OTHER samples; what
  // still takes this kernel in a forward are the ragged N = 3 / 6 / 10 heads on a few hundred rows: microseconds either way.
  // HOISDF_FWD_SPLITK=1 brings it back for A/B runs)
  static int fwd_splitk = -1;
  if (fwd_splitk < 0) { const char* e = getenv("HOISDF_FWD_SPLITK"); fwd_splitk = (e && atoi(e) == 1) ? 1 : 0; }
  if (fwd_splitk && act == 0 && relu_bits == nullptr) {
    g.splitk = plan_small_splitk(cdiv(M, BM) * cdiv(N, BN), K, g.k_per_split);
    if (g.splitk > 1) {
      g.atomic_out = 1;
      if (int rc = zero_rows(y, ldy, M, N, as_stream(stream))) return rc;
    }
  }
  return launch_gemm<true, true>(g, as_stream(stream));
}

extern "C" int hoisdf_linear_bwd_input(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p,
                                       const float* W, int ldw, float* dx, int lddx, long M, int N, int K,
                                       int accumulate, void* stream) {
  HOISDF_REQUIRE(M == 0 || (dy && W && dx), HOISDF_ERR_INVALID, "linear_bwd_input: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldw >= K && lddx >= K && M < (1L << 31) &&
                     drop_p >= 0.f && drop_p < 1.f,
                 HOISDF_ERR_INVALID, "linear_bwd_input: bad sizes");
  if (M == 0) return HOISDF_OK;
  GemmArgs g{};
  // dx[m][k] = sum_n dy[m][n] W[n][k]: contraction index n; "B" is W read as [n][k] = [contract][out]
  g.A = dy; g.B = W; g.C = dx; g.bias = nullptr; g.abits = relu_bits; g.ldbits = (N + 31) / 32; g.ascale = 1.f / (1.f - drop_p);
  g.M = (int)M; g.N = K; g.K = N; g.lda = lddy; g.ldb = ldw; g.ldc = lddx;
  g.act = 0; g.drop_p = 0.f; g.inv_keep = 1.f; g.seed = 0;
  g.splitk = plan_small_splitk(cdiv(M, BM) * cdiv(K, BN), N, g.k_per_split);
  g.atomic_out = g.splitk > 1;
  g.beta = accumulate ? 1 : 0;
  if (g.atomic_out && !accumulate)          // accumulating: the atomics simply add onto the existing gradient
    if (int rc = zero_rows(dx, lddx, M, K, as_stream(stream))) return rc;
  return launch_gemm<true, false>(g, as_stream(stream));
}

extern "C" long hoisdf_linear_bwd_weight_workspace(long M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int splitk, kper;
  plan_splitk(M, N, K, splitk, kper);
  if (splitk <= 1) return 0;
  return (long)splitk * ((long)N * K + N);
}

extern "C" int hoisdf_linear_bwd_weight(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p,
                                        const float* x, int ldx, float* dW, int lddw, float* db, long M, int N,
                                        int K, float* workspace, long workspace_floats, void* stream) {
  HOISDF_REQUIRE(dW && (M == 0 || (dy && x)), HOISDF_ERR_INVALID, "linear_bwd_weight: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K && M < (1L << 31) &&
                     drop_p >= 0.f && drop_p < 1.f,
                 HOISDF_ERR_INVALID, "linear_bwd_weight: bad sizes");
  if (M == 0) return HOISDF_OK;
  hipStream_t st = as_stream(stream);
  GemmArgs g{};
  // dW[n][k] = sum_m dy[m][n] x[m][k]: out rows n, out cols k, contraction m
  g.A = dy; g.B = x; g.bias = nullptr; g.abits = relu_bits; g.ldbits = (N + 31) / 32; g.ascale = 1.f / (1.f - drop_p);
  g.M = N; g.N = K; g.K = (int)M; g.lda = lddy; g.ldb = ldx;
  g.act = 0; g.drop_p = 0.f; g.inv_keep = 1.f; g.seed = 0;
  int splitk, kper;
  plan_splitk(M, N, K, splitk, kper);
  g.splitk = splitk; g.k_per_split = kper;
  const long need = splitk > 1 ? (long)splitk * ((long)N * K + N) : 0;
  if (splitk == 1) {
    // one pass, direct stores (dW/db fully overwritten; db via atomics needs zero only if tiles_m... =1 split: plain)
    g.C = dW; g.ldc = lddw; g.atomic_out = 0; g.partial = 1; g.c_split_stride = 0;
    g.colsum = db; g.colsum_split_stride = 0;
    return launch_gemm<false, false>(g, st);
  }
  if (workspace && workspace_floats >= need) {
    HOISDF_REQUIRE(lddw == K, HOISDF_ERR_INVALID, "linear_bwd_weight: workspace mode needs a dense dW (lddw == K)");
    g.C = workspace; g.ldc = K; g.atomic_out = 0; g.partial = 1; g.c_split_stride = (long)N * K;
    g.colsum = db ? workspace + (size_t)splitk * N * K : nullptr; g.colsum_split_stride = N;
    int rc = launch_gemm<false, false>(g, st);
    if (rc) return rc;
    const long n = (long)N * K;
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(256), 0, st, workspace, (long)N * K, splitk, dW, n);
    if (db)
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, workspace + (size_t)splitk * N * K, (long)N,
                         splitk, db, (long)N);
    return check_launch("linear_bwd_weight reduce");
  }
  // workspace-less: atomics into caller-zeroed dW / db
  g.C = dW; g.ldc = lddw; g.atomic_out = 1; g.partial = 0; g.c_split_stride = 0;
  g.colsum = db; g.colsum_split_stride = 0;
  return launch_gemm<false, false>(g, st);
}

extern "C" int hoisdf_relu_dropout_bwd(const float* y, int ldy, const float* dy, int lddy, float* dpre,
                                       int ldd, long M, int N, float drop_p, void* stream) {
  HOISDF_REQUIRE(y && dy && dpre && M >= 0 && N > 0 && drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID,
                 "relu_dropout_bwd: bad arguments");
  if (M == 0) return HOISDF_OK;
  long total = M * (long)N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), y, ldy, dy,
                     lddy, dpre, ldd, M, N, 1.f / (1.f - drop_p));
  return check_launch("relu_dropout_bwd");
}
