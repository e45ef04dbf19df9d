// Split-precision GEMM family for the linear layers (opt-in, cfg.gemm_split): the three contractions of a linear layer on
// the 16-bit MFMA pipe (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate) with both operands split into f16 hi + lo parts and
// three products per contraction (hi.hi + hi.lo + lo.hi, f32 accumulation): ~21-22 significant bits per operand, the
// precision class of attention_split.hip.  reference: common/nets/layer.py:168-201 (MLP), common/nets/transformer.py:286-302
// (the encoder layers' in/out projections and feed-forward), main/model.py:181-244 (linear_sdfin / decoder MLPs).
//
// f16 keeps 11 + 11 bits in a hi + lo pair only while the lo part is a normal number; below that the pair degrades to an
// absolute resolution of 2^-25.  Every operand is therefore moved up by an exact power of two before the split and the
// factor is taken out of the f32 result in the epilogue:
//   * an operand whose rows are k-contiguous in memory (x, W in the forward; dy in grad-input) gets one scale PER ROW
//     (row maximum -> [2^14, 2^15)), found by the converting wave itself;
//   * an operand that has to be transposed because the contraction runs over its rows (W in grad-input; dy and x in
//     grad-weight) gets one scale for the whole tensor from an amax pre-pass (the scale cannot vary along the contraction).
// A conversion pass writes the hi / lo planes k-contiguous ([rows padded to 128][k padded to 32/64], zeros in the padding),
// applying the forward's ReLU / dropout sign bitmap to dy on the way and (grad-weight) emitting the bias gradient's
// per-tile column sums; the GEMM itself is ONE kernel for all three contractions: 128 x 128 tile, 4 waves as 2 x 2,
// each wave 64 x 64 = 2 x 2 MFMA tiles, 32-deep k-slabs of the four planes double-buffered in LDS (80 KB, 2 WG / CU).
// The epilogue is the f32 kernel's (bias, ReLU, dropout, sign bitmap, accumulate-into, LDS-transposed 128-byte stores).
// Grad-weight splits the contraction over workgroups into partial tiles + an ordered reduce (no atomics: deterministic).
#include "common.h"
#include <hip/hip_fp16.h>

namespace hoisdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TM = 128, TN = 128, KS = 32;
constexpr int RPS = 40;                 // halves per LDS row of a [128][32] slab (80 B: conflict-free ds_read_b128)
constexpr int PLANE = 128 * RPS;        // halves per plane per stage
constexpr int STAGE = 4 * PLANE;        // A hi, A lo, B hi, B lo: 40 KB
constexpr unsigned LDS_BYTES = 2u * STAGE * sizeof(_Float16);
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct SplitGemmArgs {
  const _Float16 *Ah, *Al, *Bh, *Bl;    // planes [rows padded to 128][Kp]
  const float *a_rinv, *b_rinv;         // 1 / scale per row (stride 1) or one scalar (stride 0)
  int a_rs, b_rs;
  float* C;
  const float* bias;
  uint32_t* bits_out;
  int M, N, Kp, ldc, ldbits;      // Kp: contraction length walked (a multiple of 32, zero padded in both operands)
  long lda, ldb;                  // plane row pitches in halves
  int act;
  float drop_p, inv_keep;
  uint32_t thresh;
  uint64_t seed;
  int splitk, k_per_split;
  long c_split_stride;
  int tiles_m, tiles_n, vecC, beta;
};

// power-of-two scale that brings amax into [2^14, 2^15); inv = 1 / scale (exact)
__device__ __forceinline__ float pow2_scale(float amax, float& inv) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  if (e == 0 || e == 255) { inv = 1.f; return 1.f; }            // zero / denormal / non-finite: leave alone
  int se = 127 + 14 - (e - 127);
  se = se < 4 ? 4 : (se > 250 ? 250 : se);
  inv = __uint_as_float((uint32_t)(254 - se) << 23);
  return __uint_as_float((uint32_t)se << 23);
}

__device__ __forceinline__ void split1(float e, _Float16& hi, _Float16& lo) {
  hi = (_Float16)e;
  lo = (_Float16)(e - (float)hi);
}
}  // namespace

// ---- amax of a [R][C] matrix (row stride ld) into a zeroed device word (float bits; non-negative floats order as uints).
// A wave walks whole rows (no index division; 1 KB coalesced reads), one atomic per block.
__global__ __launch_bounds__(256) void gs_amax_kernel(const float* __restrict__ src, long ld, long R, int C, int vec,
                                                      uint32_t* __restrict__ out) {
  __shared__ float s_m[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m0 = 0.f, m1 = 0.f;
  for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
    const float* s = src + r * ld;
    if (vec) {
      int c = lane * 4;
      for (; c + 256 < C; c += 512) {
        const float4 v = *reinterpret_cast<const float4*>(s + c);
        const float4 w = *reinterpret_cast<const float4*>(s + c + 256);
        m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(w.x), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w))));
      }
      if (c < C) {
        const float4 v = *reinterpret_cast<const float4*>(s + c);
        m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
    } else {
      for (int c = lane; c < C; c += 64) m0 = fmaxf(m0, fabsf(s[c]));
    }
  }
  const float m = wave_max(fmaxf(m0, m1));
  if (lane == 0) s_m[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (t > 0.f) atomicMax(out, __float_as_uint(t));
  }
}

// ---- plain conversion: rows k-contiguous in memory.  One wave per row: pass 1 finds the row maximum (the row stays in
// L1 / L2 for pass 2), pass 2 scales, splits and writes 8-byte hi / lo pieces.  Rows >= R and columns >= K are zero.
__global__ __launch_bounds__(256) void gs_convert_rows_kernel(const float* __restrict__ src, long ld, int R, int K, int Rp,
                                                              int Kp, const uint32_t* __restrict__ bits, int ldbits,
                                                              float ascale, int vec, _Float16* __restrict__ hi,
                                                              _Float16* __restrict__ lo, float* __restrict__ rinv) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Rp) return;
  _Float16* oh = hi + (size_t)row * Kp;
  _Float16* ol = lo + (size_t)row * Kp;
  if (row >= R) {
    for (int c = lane * 8; c < Kp; c += 512) {
      *reinterpret_cast<uint4*>(oh + c) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(ol + c) = make_uint4(0u, 0u, 0u, 0u);
    }
    if (lane == 0) rinv[row] = 1.f;
    return;
  }
  const float* s = src + (size_t)row * ld;
  const uint32_t* bw = bits ? bits + (size_t)row * ldbits : nullptr;
  float amax = 0.f;
  if (vec) {
    for (int c = lane * 4; c < K; c += 256) {
      float4 v = *reinterpret_cast<const float4*>(s + c);
      if (bw) {
        const uint32_t nib = bw[c >> 5] >> (c & 31);
        v.x = (nib & 1u) ? v.x : 0.f; v.y = (nib & 2u) ? v.y : 0.f; v.z = (nib & 4u) ? v.z : 0.f; v.w = (nib & 8u) ? v.w : 0.f;
      }
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  } else {
    for (int c = lane; c < K; c += 64) {
      float v = s[c];
      if (bw && !((bw[c >> 5] >> (c & 31)) & 1u)) v = 0.f;
      amax = fmaxf(amax, fabsf(v));
    }
  }
  amax = wave_max(amax) * ascale;
  float inv;
  const float sc = pow2_scale(amax, inv) * ascale;           // ascale (1 / keep) folded into the multiplier: dy * ascale is
  if (lane == 0) rinv[row] = inv;                            // rounded once more in the f32 path, not here
  if (vec) {
    for (int c = lane * 4; c < Kp; c += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < K) {
        v = *reinterpret_cast<const float4*>(s + c);
        if (bw) {
          const uint32_t nib = bw[c >> 5] >> (c & 31);
          v.x = (nib & 1u) ? v.x : 0.f; v.y = (nib & 2u) ? v.y : 0.f; v.z = (nib & 4u) ? v.z : 0.f; v.w = (nib & 8u) ? v.w : 0.f;
        }
      }
      _Float16 h0, h1, h2, h3, l0, l1, l2, l3;
      split1(v.x * sc, h0, l0); split1(v.y * sc, h1, l1); split1(v.z * sc, h2, l2); split1(v.w * sc, h3, l3);
      *reinterpret_cast<f16x4*>(oh + c) = f16x4{h0, h1, h2, h3};
      *reinterpret_cast<f16x4*>(ol + c) = f16x4{l0, l1, l2, l3};
    }
  } else {
    for (int c = lane; c < Kp; c += 64) {
      float v = c < K ? s[c] : 0.f;
      if (bw && c < K && !((bw[c >> 5] >> (c & 31)) & 1u)) v = 0.f;
      _Float16 h, l;
      split1(v * sc, h, l);
      oh[c] = h;
      ol[c] = l;
    }
  }
}

// ---- transposing conversion: src [M][C] (C contiguous) -> planes [Cp][Mp] (M contiguous), one scale for the tensor from
// the amax word.  One block per 64 x 64 tile through LDS (coalesced 64-byte reads, 32-byte writes).  Optionally the
// per-tile column sums of the (masked, unscaled) source: colsum_part[m-tile][C], summed in tile order afterwards.
__global__ __launch_bounds__(256) void gs_convert_trn_kernel(const float* __restrict__ src, long ld, int M, int C, int Mp,
                                                             const uint32_t* __restrict__ amax_bits,
                                                             const uint32_t* __restrict__ bits, int ldbits, float ascale,
                                                             int vec, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                             float* __restrict__ sinv, float* __restrict__ colsum_part) {
  constexpr int TP = 72;
  __shared__ __attribute__((aligned(16))) _Float16 tile[2][64 * TP];
  __shared__ float red[64][65];
  const int tid = threadIdx.x;
  const int r = tid >> 2, dc = (tid & 3) * 16;
  const int m = blockIdx.x * 64 + r, c0 = blockIdx.y * 64 + dc;
  float inv;
  const float sc = pow2_scale(__uint_as_float(amax_bits[0]) * ascale, inv) * ascale;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) sinv[0] = inv;
  float e[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) e[i] = 0.f;
  if (m < M) {
    const float* s = src + (size_t)m * ld;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + 4 * i;
      if (vec && c + 3 < C) {
        const float4 v = *reinterpret_cast<const float4*>(s + c);
        e[4 * i] = v.x; e[4 * i + 1] = v.y; e[4 * i + 2] = v.z; e[4 * i + 3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < C) e[4 * i + j] = s[c + j];
      }
    }
    if (bits) {
      // 16 consecutive columns starting at a multiple of 16: inside one 32-bit word
      const uint32_t w = c0 < C ? bits[(size_t)m * ldbits + (c0 >> 5)] >> (c0 & 31) : 0u;
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i] = ((w >> i) & 1u) ? e[i] : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    _Float16 h, l;
    split1(e[i] * sc, h, l);
    tile[0][(dc + i) * TP + r] = h;
    tile[1][(dc + i) * TP + r] = l;
    if (colsum_part) red[r][dc + i] = e[i] * ascale;
  }
  __syncthreads();
  const int d = tid >> 2, rc = (tid & 3) * 16;
  _Float16* oh = hi + (size_t)(blockIdx.y * 64 + d) * Mp + blockIdx.x * 64 + rc;
  _Float16* ol = lo + (size_t)(blockIdx.y * 64 + d) * Mp + blockIdx.x * 64 + rc;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    *reinterpret_cast<uint4*>(oh + 8 * i) = *reinterpret_cast<const uint4*>(&tile[0][d * TP + rc + 8 * i]);
    *reinterpret_cast<uint4*>(ol + 8 * i) = *reinterpret_cast<const uint4*>(&tile[1][d * TP + rc + 8 * i]);
  }
  if (colsum_part && tid < 64) {
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) s += red[j][tid];
    const int c = blockIdx.y * 64 + tid;
    if (c < C) colsum_part[(size_t)blockIdx.x * C + c] = s;
  }
}

// out[c] = sum over tiles t of part[t][c] in a fixed order: block = 16 columns x 16 interleaved tile groups
__global__ __launch_bounds__(256) void gs_colsum_reduce_kernel(const float* __restrict__ part, int ntile, int C,
                                                               float* __restrict__ out) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    int t = grp;
    for (; t + 16 < ntile; t += 32) {
      s0 += part[(size_t)t * C + c];
      s1 += part[(size_t)(t + 16) * C + c];
    }
    if (t < ntile) s0 += part[(size_t)t * C + c];
  }
  red[grp][cl] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j][threadIdx.x];
    out[c] = s;
  }
}

// out[i] = sum_s part[s * stride + i]  (n a multiple of 4 or handled by the tail)
__global__ __launch_bounds__(256) void gs_reduce_partials_kernel(const float* __restrict__ part, long stride, int splits,
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

// ============================================================================================
// C[M][N] = epilogue( (A . B^T) / (sa[m] sb[n]) ),  A planes [Mp][Kp], B planes [Np][Kp]
// ============================================================================================
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(SplitGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 31, h = lane >> 5;

  const int ntile = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  int split, t;
  if (g.splitk > 1) {
    split = (bid & 7) + 8 * (bid / (8 * ntile));       // every tile of one k-slice on the same XCD (shared L2), as gemm.hip
    t = (bid >> 3) % ntile;
    if (split >= g.splitk) return;
  } else {
    split = 0;
    t = xcd_remap(bid, ntile);
  }
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.Kp, kbeg + g.k_per_split);
  const int nk = (kend - kbeg) / KS;


  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: thread -> (row tid >> 1, 16 halves at (tid & 1) * 16) of each of the four planes
  const size_t goffA = (size_t)(m0 + (tid >> 1)) * g.lda + kbeg + (tid & 1) * 16;
  const size_t goffB = (size_t)(n0 + (tid >> 1)) * g.ldb + kbeg + (tid & 1) * 16;
  const _Float16 *pah = g.Ah + goffA, *pal = g.Al + goffA, *pbh = g.Bh + goffB, *pbl = g.Bl + goffB;
  const int soff = (tid >> 1) * RPS + (tid & 1) * 16;
  // Two register sets: the slab loaded during iteration kt is stored to LDS at the end of iteration kt + 1 and consumed in
  // kt + 2, so a global load has two iterations (~1500 MFMA cycles per wave, two waves per SIMD) to land; with a distance
  // of one the 768 MFMA cycles of a slab did not cover the HBM latency (measured 195 TF-equivalent).
  uint4 p0, p1, p2, p3, p4, p5, p6, p7, q0, q1, q2, q3, q4, q5, q6, q7;      // register sets "p" and "q" (named scalars:
                                                                             // an indexed struct went to scratch memory)
#define GS_GLOAD(x, kt)                                                                                     \
  do {                                                                                                      \
    const int o_ = (kt) * KS;                                                                               \
    x##0 = *reinterpret_cast<const uint4*>(pah + o_); x##1 = *reinterpret_cast<const uint4*>(pah + o_ + 8); \
    x##2 = *reinterpret_cast<const uint4*>(pal + o_); x##3 = *reinterpret_cast<const uint4*>(pal + o_ + 8); \
    x##4 = *reinterpret_cast<const uint4*>(pbh + o_); x##5 = *reinterpret_cast<const uint4*>(pbh + o_ + 8); \
    x##6 = *reinterpret_cast<const uint4*>(pbl + o_); x##7 = *reinterpret_cast<const uint4*>(pbl + o_ + 8); \
  } while (0)
#define GS_SSTORE(buf, x)                                                                               \
  do {                                                                                                  \
    _Float16* p_ = (buf) + soff;                                                                        \
    *reinterpret_cast<uint4*>(p_) = x##0;             *reinterpret_cast<uint4*>(p_ + 8) = x##1;             \
    *reinterpret_cast<uint4*>(p_ + PLANE) = x##2;     *reinterpret_cast<uint4*>(p_ + PLANE + 8) = x##3;     \
    *reinterpret_cast<uint4*>(p_ + 2 * PLANE) = x##4; *reinterpret_cast<uint4*>(p_ + 2 * PLANE + 8) = x##5; \
    *reinterpret_cast<uint4*>(p_ + 3 * PLANE) = x##6; *reinterpret_cast<uint4*>(p_ + 3 * PLANE + 8) = x##7; \
  } while (0)
  const int arow = (wm * 64 + c) * RPS + 8 * h;
  const int brow = (wn * 64 + c) * RPS + 8 * h;
  auto compute = [&](const _Float16* Ah) {
    const _Float16* Al = Ah + PLANE;
    const _Float16* Bh = Ah + 2 * PLANE;
    const _Float16* Bl = Ah + 3 * PLANE;
#pragma unroll
    for (int ks = 0; ks < KS / 16; ++ks) {
      const f16x8 a0 = *reinterpret_cast<const f16x8*>(Ah + arow + 16 * ks);
      const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ah + arow + 32 * RPS + 16 * ks);
      const f16x8 a0l = *reinterpret_cast<const f16x8*>(Al + arow + 16 * ks);
      const f16x8 a1l = *reinterpret_cast<const f16x8*>(Al + arow + 32 * RPS + 16 * ks);
      const f16x8 b0 = *reinterpret_cast<const f16x8*>(Bh + brow + 16 * ks);
      const f16x8 b1 = *reinterpret_cast<const f16x8*>(Bh + brow + 32 * RPS + 16 * ks);
      const f16x8 b0l = *reinterpret_cast<const f16x8*>(Bl + brow + 16 * ks);
      const f16x8 b1l = *reinterpret_cast<const f16x8*>(Bl + brow + 32 * RPS + 16 * ks);
      acc[0][0] = MF16(a0l, b0, acc[0][0]);       // small terms first
      acc[0][1] = MF16(a0l, b1, acc[0][1]);
      acc[1][0] = MF16(a1l, b0, acc[1][0]);
      acc[1][1] = MF16(a1l, b1, acc[1][1]);
      acc[0][0] = MF16(a0, b0l, acc[0][0]);
      acc[0][1] = MF16(a0, b1l, acc[0][1]);
      acc[1][0] = MF16(a1, b0l, acc[1][0]);
      acc[1][1] = MF16(a1, b1l, acc[1][1]);
      acc[0][0] = MF16(a0, b0, acc[0][0]);
      acc[0][1] = MF16(a0, b1, acc[0][1]);
      acc[1][0] = MF16(a1, b0, acc[1][0]);
      acc[1][1] = MF16(a1, b1, acc[1][1]);
    }
  };
  if (nk > 0) {
    GS_GLOAD(p, 0);
    GS_SSTORE(lds, p);
    if (nk > 1) GS_GLOAD(q, 1);
  }
  __syncthreads();
  // iteration kt (even): load slab kt + 2 into p, compute stage 0, park q (slab kt + 1) in stage 1; odd: mirrored
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 2 < nk) GS_GLOAD(p, kt + 2);
    compute(lds);
    if (kt + 1 < nk) GS_SSTORE(lds + STAGE, q);
    __syncthreads();
    if (kt + 1 < nk) {
      if (kt + 3 < nk) GS_GLOAD(q, kt + 3);
      compute(lds + STAGE);
      if (kt + 2 < nk) GS_SSTORE(lds, p);
      __syncthreads();
    }
  }

  // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
  // the A-row scales of the tile go through the (now idle) staging buffer, behind the 16 KB the store path below uses: a
  // static array would push 2 x 80 KB of dynamic LDS over the CU's 160 KB and halve the occupancy
  float* s_ra = reinterpret_cast<float*>(lds) + 4 * 32 * 32;
  if (tid < TM) s_ra[tid] = g.a_rinv[(size_t)(m0 + tid) * g.a_rs];      // planes are padded: m0 + tid < Mp always
  __syncthreads();
  float* Cb = g.C + (size_t)split * g.c_split_stride;
  const int rbase = m0 + wm * 64 + 4 * h;
  const int cbase = n0 + wn * 64 + c;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = cbase + j * 32;
    const float sb = g.b_rinv[(size_t)col * g.b_rs];                         // col < Np always (padded planes)
    const float bv = (g.bias != nullptr && split == 0 && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sa = s_ra[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        float v = acc[i][j][r] * (sa * sb) + bv;
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
  const bool full = (m0 + TM <= g.M) && (n0 + TN <= g.N);
  if (full && g.vecC) {
    // through LDS: one 32x32 block per wave at a time in a wave-private 4 KB slice, read back row-wise so that one
    // global_store_dwordx4 covers 8 complete 128-byte row segments (same scheme as gemm.hip)
    float* w = reinterpret_cast<float*>(lds) + wave * (32 * 32);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c] = acc[i][j][r];
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
          if (row < g.M && col < g.N) {
            float* cp = Cb + (size_t)row * g.ldc + col;
            *cp = g.beta ? *cp + acc[i][j][r] : acc[i][j][r];
          }
        }
  }
  if (g.bits_out) {
    uint32_t w0 = 0u, w1 = 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = i * 32 + (r & 3) + 8 * (r >> 2);
        const unsigned long long b0 = __ballot(acc[i][0][r] > 0.f);
        const unsigned long long b1 = __ballot(acc[i][1][r] > 0.f);
        if (lane == rl) { w0 = (uint32_t)b0; w1 = (uint32_t)b1; }
        if (lane == rl + 4) { w0 = (uint32_t)(b0 >> 32); w1 = (uint32_t)(b1 >> 32); }
      }
    const int row = m0 + wm * 64 + lane;
    const int wcol = (n0 + wn * 64) >> 5;
    const int nvalid = g.N - (n0 + wn * 64);
    if (row < g.M) {
      if (nvalid > 0) g.bits_out[(size_t)row * g.ldbits + wcol] = nvalid >= 32 ? w0 : (w0 & ((1u << nvalid) - 1u));
      if (nvalid > 32) g.bits_out[(size_t)row * g.ldbits + wcol + 1] = nvalid >= 64 ? w1 : (w1 & ((1u << (nvalid - 32)) - 1u));
    }
  }
}

namespace {
inline long up(long v, long m) { return (v + m - 1) / m * m; }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// workspace carving (256-byte aligned pieces)
struct Carver {
  char* base; size_t off;
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};
struct Planes { _Float16 *hi, *lo; float* rinv; long rows_p, kp; int rs; };

Planes take_rows(Carver& c, long R, long K) {
  Planes p;
  p.rows_p = up(R, 128); p.kp = up(K, KS); p.rs = 1;
  p.hi = c.take<_Float16>((size_t)p.rows_p * p.kp);
  p.lo = c.take<_Float16>((size_t)p.rows_p * p.kp);
  p.rinv = c.take<float>((size_t)p.rows_p);
  return p;
}
// transposed operand: rows = the source's columns (padded to 128), contraction = the source's rows (padded to 64)
Planes take_trn(Carver& c, long Msrc, long Csrc) {
  Planes p;
  p.rows_p = up(Csrc, 128); p.kp = up(Msrc, 64); p.rs = 0;
  p.hi = c.take<_Float16>((size_t)p.rows_p * p.kp);
  p.lo = c.take<_Float16>((size_t)p.rows_p * p.kp);
  p.rinv = c.take<float>(64);           // [0] = 1 / scale, [1] = amax bits
  return p;
}

int convert_rows(const float* src, long ld, long R, int K, const uint32_t* bits, float ascale, const Planes& p, hipStream_t st) {
  const int vec = al16(src) && (ld % 4 == 0) && (K % 4 == 0);
  hipLaunchKernelGGL(gs_convert_rows_kernel, dim3((unsigned)(p.rows_p / 4)), dim3(256), 0, st, src, ld, (int)R, K,
                     (int)p.rows_p, (int)p.kp, bits, (K + 31) / 32, ascale, vec, p.hi, p.lo, p.rinv);
  return check_launch("gemm_split convert_rows");
}
int convert_trn(const float* src, long ld, long M, int C, const uint32_t* bits, float ascale, const Planes& p,
                float* colsum_part, hipStream_t st) {
  uint32_t* amax = reinterpret_cast<uint32_t*>(p.rinv + 1);
  if (hipMemsetAsync(amax, 0, sizeof(uint32_t), st) != hipSuccess) {
    set_error("gemm_split: memset failed");
    return HOISDF_ERR_LAUNCH;
  }
  const int vec = al16(src) && (ld % 4 == 0) && (C % 4 == 0);
  long nb = (M + 3) / 4;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(gs_amax_kernel, dim3((unsigned)nb), dim3(256), 0, st, src, ld, M, C, vec, amax);
  hipLaunchKernelGGL(gs_convert_trn_kernel, dim3((unsigned)(p.kp / 64), (unsigned)(p.rows_p / 64)), dim3(256), 0, st, src,
                     ld, (int)M, C, (int)p.kp, amax, bits, (C + 31) / 32, ascale, vec, p.hi, p.lo, p.rinv, colsum_part);
  return check_launch("gemm_split convert_trn");
}

int launch_split(SplitGemmArgs g, const Planes& A, const Planes& B, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_BYTES) != hipSuccess) {
      set_error("gemm_split: cannot raise the dynamic LDS limit to %u bytes", LDS_BYTES);
      return HOISDF_ERR_LAUNCH;
    }
    attr_set = true;
  }
  g.Ah = A.hi; g.Al = A.lo; g.a_rinv = A.rinv; g.a_rs = A.rs;
  g.Bh = B.hi; g.Bl = B.lo; g.b_rinv = B.rinv; g.b_rs = B.rs;
  g.lda = A.kp; g.ldb = B.kp;
  g.Kp = (int)(A.kp < B.kp ? A.kp : B.kp);        // both cover the contraction; the longer one's tail is zero padding
  g.tiles_m = cdiv(g.M, TM);
  g.tiles_n = cdiv(g.N, TN);
  g.vecC = al16(g.C) && (g.ldc % 4 == 0) && (g.c_split_stride % 4 == 0);
  const int ntile = g.tiles_m * g.tiles_n;
  const int nwg = g.splitk > 1 ? ntile * 8 * cdiv(g.splitk, 8) : ntile;
  hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)nwg), dim3(256), LDS_BYTES, st, g);
  return check_launch("gemm_split");
}

// contraction slices for grad-weight: ~1024 workgroups, >= 8 slabs each
void plan_split(long Kp, int tiles, int& splitk, int& kper) {
  const int slabs = (int)(Kp / KS);
  int want = tiles >= 1024 ? 1 : cdiv(1024, tiles);
  if (want > slabs / 8) want = slabs / 8 > 0 ? slabs / 8 : 1;
  kper = cdiv(slabs, want) * KS;
  splitk = cdiv(Kp, kper);
}
}  // namespace

}  // namespace hoisdf

using namespace hoisdf;

extern "C" long hoisdf_linear_split_workspace(long M, int N, int K, int which) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  Carver c{nullptr, 0};
  if (which == 0) {                    // forward: x rows, W rows
    take_rows(c, M, K);
    take_rows(c, N, K);
  } else if (which == 1) {             // grad-input: dy rows, W^T
    take_rows(c, M, N);
    take_trn(c, N, K);
  } else {                             // grad-weight: dy^T, x^T, partial tiles, bias-gradient partials
    Planes a = take_trn(c, M, N);
    take_trn(c, M, K);
    int splitk, kper;
    plan_split(a.kp, cdiv(N, TM) * cdiv(K, TN), splitk, kper);
    if (splitk > 1) c.take<float>((size_t)splitk * N * K);
    c.take<float>((size_t)(a.kp / 64) * N);
  }
  return (long)((c.off + 255) & ~(size_t)255);
}

extern "C" int hoisdf_linear_fwd_split(const float* x, int ldx, const float* W, int ldw, const float* bias, float* y,
                                       int ldy, long M, int N, int K, int act, float drop_p, uint64_t seed,
                                       uint32_t* relu_bits, void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(M == 0 || (x && W && y), HOISDF_ERR_INVALID, "linear_fwd_split: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N && M < (1L << 31), HOISDF_ERR_INVALID,
                 "linear_fwd_split: bad sizes M=%ld N=%d K=%d ldx=%d ldw=%d ldy=%d", M, N, K, ldx, ldw, ldy);
  HOISDF_REQUIRE(drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID, "linear_fwd_split: drop_p=%f", drop_p);
  if (M == 0) return HOISDF_OK;
  HOISDF_REQUIRE(workspace && workspace_bytes >= hoisdf_linear_split_workspace(M, N, K, 0), HOISDF_ERR_INVALID,
                 "linear_fwd_split: workspace too small");
  hipStream_t st = as_stream(stream);
  Carver c{static_cast<char*>(workspace), 0};
  Planes A = take_rows(c, M, K), B = take_rows(c, N, K);
  if (int rc = convert_rows(x, ldx, M, K, nullptr, 1.f, A, st)) return rc;
  if (int rc = convert_rows(W, ldw, N, K, nullptr, 1.f, B, st)) return rc;
  SplitGemmArgs g{};
  g.C = y; g.ldc = ldy; g.bias = bias; g.M = (int)M; g.N = N;
  g.act = act; g.drop_p = drop_p; g.inv_keep = 1.f / (1.f - drop_p); g.thresh = drop_threshold(drop_p); g.seed = seed;
  g.bits_out = relu_bits; g.ldbits = (N + 31) / 32;
  g.splitk = 1; g.k_per_split = (int)A.kp;
  return launch_split(g, A, B, st);
}

extern "C" int hoisdf_linear_bwd_input_split(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p,
                                             const float* W, int ldw, float* dx, int lddx, long M, int N, int K,
                                             int accumulate, void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(M == 0 || (dy && W && dx), HOISDF_ERR_INVALID, "linear_bwd_input_split: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldw >= K && lddx >= K && M < (1L << 31) && drop_p >= 0.f &&
                     drop_p < 1.f,
                 HOISDF_ERR_INVALID, "linear_bwd_input_split: bad sizes");
  if (M == 0) return HOISDF_OK;
  HOISDF_REQUIRE(workspace && workspace_bytes >= hoisdf_linear_split_workspace(M, N, K, 1), HOISDF_ERR_INVALID,
                 "linear_bwd_input_split: workspace too small");
  hipStream_t st = as_stream(stream);
  Carver c{static_cast<char*>(workspace), 0};
  // dx[m][k] = sum_n dy[m][n] W[n][k]: A = dy rows (contraction n contiguous), B = W^T ([k][n])
  Planes A = take_rows(c, M, N), B = take_trn(c, N, K);
  if (int rc = convert_rows(dy, lddy, M, N, relu_bits, 1.f / (1.f - drop_p), A, st)) return rc;
  if (int rc = convert_trn(W, ldw, N, K, nullptr, 1.f, B, nullptr, st)) return rc;
  SplitGemmArgs g{};
  g.C = dx; g.ldc = lddx; g.M = (int)M; g.N = K;
  g.inv_keep = 1.f;
  g.splitk = 1; g.k_per_split = (int)B.kp;
  g.beta = accumulate ? 1 : 0;
  return launch_split(g, A, B, st);
}

extern "C" int hoisdf_linear_bwd_weight_split(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p,
                                              const float* x, int ldx, float* dW, int lddw, float* db, long M, int N, int K,
                                              void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(dW && (M == 0 || (dy && x)), HOISDF_ERR_INVALID, "linear_bwd_weight_split: null pointer");
  HOISDF_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw == K && M < (1L << 31) && drop_p >= 0.f &&
                     drop_p < 1.f,
                 HOISDF_ERR_INVALID, "linear_bwd_weight_split: bad sizes (a dense dW, lddw == K, is required)");
  HOISDF_REQUIRE(workspace && workspace_bytes >= hoisdf_linear_split_workspace(M, N, K, 2), HOISDF_ERR_INVALID,
                 "linear_bwd_weight_split: workspace too small");
  hipStream_t st = as_stream(stream);
  Carver c{static_cast<char*>(workspace), 0};
  // dW[n][k] = sum_m dy[m][n] x[m][k]: A = dy^T ([n][m]), B = x^T ([k][m])
  Planes A = take_trn(c, M, N), B = take_trn(c, M, K);
  int splitk, kper;
  plan_split(A.kp, cdiv(N, TM) * cdiv(K, TN), splitk, kper);
  float* part = splitk > 1 ? c.take<float>((size_t)splitk * N * K) : nullptr;
  float* cpart = c.take<float>((size_t)(A.kp / 64) * N);
  if (int rc = convert_trn(dy, lddy, M, N, relu_bits, 1.f / (1.f - drop_p), A, db ? cpart : nullptr, st)) return rc;
  if (int rc = convert_trn(x, ldx, M, K, nullptr, 1.f, B, nullptr, st)) return rc;
  SplitGemmArgs g{};
  g.M = N; g.N = K; g.inv_keep = 1.f;
  g.splitk = splitk; g.k_per_split = kper;
  if (splitk > 1) { g.C = part; g.ldc = K; g.c_split_stride = (long)N * K; }
  else { g.C = dW; g.ldc = lddw; }
  if (int rc = launch_split(g, A, B, st)) return rc;
  if (splitk > 1) {
    const long n = (long)N * K;
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gs_reduce_partials_kernel, dim3(blocks), dim3(256), 0, st, part, n, splitk, dW, n);
  }
  if (db) hipLaunchKernelGGL(gs_colsum_reduce_kernel, dim3((unsigned)cdiv(N, 16)), dim3(256), 0, st, cpart, (int)(A.kp / 64), N, db);
  return check_launch("linear_bwd_weight_split reduce");
}
