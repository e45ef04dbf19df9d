// Split-precision GEMM family for the linear layers (opt-in, cfg.gemm_split): the three contractions of a linear layer on
// the 16-bit MFMA pipe (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate) with both operands split into f16 hi + lo parts and
// three products per contraction (hi.hi + hi.lo + lo.hi, f32 accumulation): ~21-22 significant bits per operand, the
// precision class of attention_split.hip.  reference: common/nets/layer.py:168-201 (MLP), common/nets/transformer.py:286-302
// (the encoder layers' in/out projections and feed-forward), main/model.py:181-244 (linear_sdfin / decoder MLPs).
//
// f16 keeps 11 + 11 bits in a hi + lo pair only while the lo part is a normal number; below that the pair degrades to an
// absolute resolution of 2^-25.  Every operand is therefore moved up by an exact power of two before the split and the
// factor is taken out of the f32 result in the epilogue:
//   * forward / grad-input: the activation operand (x, dy) is read as f32 and split on its way into LDS, one scale PER ROW
//     (row maximum -> [2^14, 2^15)) from a one-read pre-pass (gs_row_scale_kernel) - no hi / lo copy of an activation goes
//     through HBM; the weights (tiny) are pre-converted to hi / lo planes, per row (forward) or transposed with one
//     scale (grad-input).  The ReLU / dropout sign bitmap of the forward is applied to dy during the conversion.
//   * grad-weight: the contraction runs over the rows of both operands, so both are written once as TRANSPOSED planes;
//     source row m of x is multiplied by its row scale sx[m], row m of dy by p / sx[m] with p = min_m sx[m] sd[m] (the
//     product of the two factors is constant along the contraction, every row's magnitude stays proportional to its
//     contribution to dW).  That pass also emits the bias gradient's per-tile column sums.
// ONE GEMM kernel (gemm_split_kc_kernel): 128 x 128 tile, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles, 32-deep
// k-slabs of the four planes double-buffered in LDS (64 KB, 2 WG / CU), raw buffer loads two slabs ahead.  The epilogue
// is the f32 kernel's (bias, ReLU, dropout, sign bitmap, accumulate-into, LDS-transposed 128-byte stores).  Grad-weight
// splits the contraction over workgroups into partial tiles + an ordered reduce (no atomics: deterministic).
// Measured (MI355X, tools/mb_gsplit.py, pre-passes included): forward / grad-input 155-215 TF-equivalent against 88-108
// for the f32 kernel; grad-weight 110-127 against 61-103 when min(N, K) >= 512, but NO gain for the 256-wide transformer
// shapes (the two conversion passes cost what the contraction saves), which therefore stay on the f32 kernel.
#include <stdlib.h>

#include "common.h"
#include <hip/hip_fp16.h>

namespace hoisdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TM = 128, TN = 128, KS = 32;
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// power-of-two scale that brings amax into [2^14, 2^15); inv = 1 / scale (exact)
__device__ __forceinline__ float pow2_scale(float amax, float& inv) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  if (e == 255) { inv = 1.f; return 1.f; }                      // non-finite: leave alone
  // zero (or denormal) rows: a LARGE scale, so that such a row never bounds the common factor of grad-weight (a row of
  // zeros must not cost the other rows their bits); 0 * 2^100 is still 0
  if (e == 0) { inv = __uint_as_float((254u - 227u) << 23); return __uint_as_float(227u << 23); }
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

// ---- transposing conversion: src [M][C] (C contiguous) -> planes [Cp][Mp] (M contiguous); every source row m (= the
// contraction index of the GEMM that consumes the planes) is multiplied by a power of two: one scalar for the tensor
// (weights) or a per-row factor whose product over the two operands of the contraction is constant (grad-weight).  One block per 64 x 64 tile through LDS (coalesced 64-byte reads, 32-byte writes).  Optionally the
// per-tile column sums of the (masked, unscaled) source: colsum_part[m-tile][C], summed in tile order afterwards.
__global__ __launch_bounds__(256) void gs_convert_trn_kernel(const float* __restrict__ src, long ld, int M, int C, int Mp,
                                                             const float* __restrict__ row_scale,
                                                             const float* __restrict__ num, int invert,
                                                             const uint32_t* __restrict__ bits, int ldbits, float ascale,
                                                             int vec, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                             float* __restrict__ colsum_part) {
  constexpr int TP = 72;
  __shared__ __attribute__((aligned(16))) _Float16 tile[2][64 * TP];
  __shared__ float red[64][65];
  const int tid = threadIdx.x;
  const int r = tid >> 2, dc = (tid & 3) * 16;
  const int m = blockIdx.x * 64 + r, c0 = blockIdx.y * 64 + dc;
  // multiplier of source row m (powers of two): row_scale[m], num[0] / row_scale[m] (invert) or the scalar num[0]
  float sc = ascale;
  if (m < M) sc *= row_scale ? (invert ? num[0] / row_scale[m] : row_scale[m]) : num[0];
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

// ---- per-row scales of a k-contiguous f32 operand (the in-kernel conversion needs them before it reads a row):
// scale[r] = the power of two that brings max |row r| (masked, times ascale) into [2^14, 2^15).  One wave per row, ONE
// read of the operand and R floats written - the GEMM converts the f32 rows itself, no hi / lo planes go through HBM.
__global__ __launch_bounds__(256) void gs_row_scale_kernel(const float* __restrict__ src, long ld, int R, int K,
                                                           const uint32_t* __restrict__ bits, int ldbits, float ascale,
                                                           int vec, float* __restrict__ scale) {
  // a wave takes FOUR rows at a time: four independent loads in flight per lane (a 256-wide row is a single float4 per
  // lane - one row per wave ran at 3.2 TB/s)
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (row0 >= R) return;
  const float* s[4];
  const uint32_t* bw[4];
  float amax[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = min(row0 + j, R - 1);
    s[j] = src + (size_t)r * ld;
    bw[j] = bits ? bits + (size_t)r * ldbits : nullptr;
  }
  if (vec) {
    for (int c = lane * 4; c < K; c += 256) {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(s[j] + c);
      if (bits) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t nib = bw[j][c >> 5] >> (c & 31);
          v[j].x = (nib & 1u) ? v[j].x : 0.f; v[j].y = (nib & 2u) ? v[j].y : 0.f;
          v[j].z = (nib & 4u) ? v[j].z : 0.f; v[j].w = (nib & 8u) ? v[j].w : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        amax[j] = fmaxf(amax[j], fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
    }
  } else {
    for (int c = lane; c < K; c += 64) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = s[j][c];
        if (bits && !((bw[j][c >> 5] >> (c & 31)) & 1u)) v = 0.f;
        amax[j] = fmaxf(amax[j], fabsf(v));
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float m = wave_max(amax[j]) * ascale;
    float inv;
    const float sc = pow2_scale(m, inv);
    if (lane == 0 && row0 + j < R) scale[row0 + j] = sc;
  }
}
// scalar form for the operands whose contraction runs over their rows: scale[0] from the tensor amax word
__global__ void gs_scalar_scale_kernel(const uint32_t* __restrict__ amax_bits, float ascale, float* __restrict__ scale) {
  float inv;
  scale[0] = pow2_scale(__uint_as_float(amax_bits[0]) * ascale, inv);
  scale[2] = inv;
}
// grad-weight: min over m of sx[m] * sd[m] (positive powers of two: ordered as uints) into out[0] (preset to +inf bits);
// finalize: out[1] = 1 / out[0]
__global__ __launch_bounds__(256) void gs_min_product_kernel(const float* __restrict__ sx, const float* __restrict__ sd,
                                                             long M, uint32_t* __restrict__ out) {
  float m = 3.0e38f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long)gridDim.x * 256) {
    const float p = fminf(fmaxf(sx[i] * sd[i], 1.0e-30f), 1.0e30f);
    m = fminf(m, p);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMin(out, __float_as_uint(m));
}
__global__ void gs_min_product_finish_kernel(float* __restrict__ v) {
  // round the clamped minimum down to a power of two (it is one unless a clamp hit) and publish its inverse
  const float p = __uint_as_float(__float_as_uint(v[0]) & 0x7f800000u);
  v[0] = p;
  v[1] = __uint_as_float((254u << 23) - __float_as_uint(p));
}

// ============================================================================================
// C[M][N] = epilogue( (A . B^T) / (sa sb) )
// LDS image of a [128 rows][32 k] plane (8 KB, unpadded): row R lives in slot (R & 3) * 32 + (R >> 2), its four 16-byte
// k-chunks at chunk ^ (R & 3).  With that permutation the MFMA fragment reads (16 consecutive rows, one chunk) are spread
// over all 64 banks and the staging writes (row = tid >> 1, two chunks) are at most 2-way conflicted.
// ============================================================================================
constexpr int PLANE_B = 128 * 64;                 // bytes per plane per stage
constexpr int STAGE_B = 4 * PLANE_B;              // A hi, A lo, B hi, B lo: 32 KB
constexpr unsigned LDS2_BYTES = 2u * STAGE_B;     // double buffered: 64 KB (2 workgroups per CU)

struct FusedArgs {
  const float* A;                           // f32 operand A [M][lda] (forward / grad-input), or
  const _Float16 *Ah, *Al;                  // pre-converted planes of A [Mp][lda halves] (grad-weight)
  long lda;
  const _Float16 *Bh, *Bl; long ldb;        // planes of B [Np][ldb halves]
  const float* a_scale; int a_rs;           // scale of A: per row (stride 1) or scalar (stride 0; planes: the product p)
  const float* b_rinv; int b_rs;            // 1 / scale of B's rows (stride 1) or scalar (stride 0)
  const uint32_t* abits; int ldbits; float ascale;     // sign bitmap of dy [rows][ceil(cols / 32)], 1 / keep
  float* C; int ldc;
  const float* bias;
  uint32_t* bits_out; int ldbits_out;
  int M, N, K;                              // output rows, output columns, contraction length
  int act; float drop_p, inv_keep; uint32_t thresh; uint64_t seed;
  int splitk, k_per_split; long c_split_stride;
  int tiles_m, tiles_n, vecC, beta;
};

__device__ __forceinline__ uint32_t pack2(_Float16 a, _Float16 b) {
  return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}
// 8 scaled values -> 16 bytes of hi, 16 bytes of lo
__device__ __forceinline__ void split8(const float (&e)[8], uint4& hi, uint4& lo) {
  _Float16 h[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) split1(e[i], h[i], l[i]);
  hi = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
  lo = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
}
__device__ __forceinline__ int lds_off(int R, int chunk) {            // byte offset inside a plane
  return (((R & 3) * 32 + (R >> 2)) << 6) + ((chunk ^ (R & 3)) << 4);
}

// ---- epilogue shared by the split GEMM kernels (C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  PER_ROW_A: the A scale is per output row (k-contiguous A), else scalar sA;
// PLANE_B: 1 / scale of the B rows from g.b_rinv, else the scalar sB.  All waves are past the main loop's last barrier.
template <bool PER_ROW_A, bool PLANE_B>
__device__ __forceinline__ void split_epilogue(const FusedArgs& g, f32x16 (&acc)[2][2], char* lds, int m0, int n0, int split,
                                               float sA, float sB) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 31, h = lane >> 5;
  // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
  // the A-row factors of the tile go through the (now idle) staging buffer, behind the 16 KB the store path below uses
  float* s_ra = reinterpret_cast<float*>(lds) + 4 * 32 * 32;
  if (tid < TM) {
    const float sc = PER_ROW_A ? g.a_scale[(size_t)min(m0 + tid, g.M - 1) * g.a_rs] : sA;
    s_ra[tid] = __uint_as_float((254u << 23) - __float_as_uint(sc));        // 1 / (a power of two)
  }
  __syncthreads();
  float* Cb = g.C + (size_t)split * g.c_split_stride;
  const int rbase = m0 + wm * 64 + 4 * h;
  const int cbase = n0 + wn * 64 + c;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = cbase + j * 32;
    const float sb = PLANE_B ? g.b_rinv[(size_t)col * g.b_rs] : __uint_as_float((254u << 23) - __float_as_uint(sB));
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
      if (nvalid > 0) g.bits_out[(size_t)row * g.ldbits_out + wcol] = nvalid >= 32 ? w0 : (w0 & ((1u << nvalid) - 1u));
      if (nvalid > 32) g.bits_out[(size_t)row * g.ldbits_out + wcol + 1] = nvalid >= 64 ? w1 : (w1 & ((1u << (nvalid - 32)) - 1u));
    }
  }
}

// ---- forward / grad-input: A f32 k-contiguous (converted here), B = weight planes.  Staging loads are raw buffer loads
// (fixed per-thread VGPR offset + a scalar k offset: no address arithmetic in the loop, out-of-buffer reads return 0) into
// TWO register sets, so that the slab consumed in iteration kt + 2 is requested in iteration kt: a load has two iterations
// (~3000 cycles with two waves per SIMD) to land.  With a distance of one the 768 MFMA cycles of a slab did not cover the
// load latency under load (measured: 4900 cycles per slab iteration).  A third set (distance 3) does not fit: 3 x 33
// staging + 64 accumulator + 64 fragment registers spill at the 256-register budget of two waves per SIMD.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool MASK, bool APLANES, int WGS>
__global__ __launch_bounds__(256, WGS) void gemm_split_kc_kernel(FusedArgs g) {
  // WGS = 2: two LDS stages (64 KB), one barrier per slab.  WGS = 3: ONE stage (32 KB) and <= 168 VGPRs, a second barrier
  // per slab, three workgroups per CU to cover the load latency of the short (K = 256) contractions.
  constexpr int S1 = WGS == 3 ? 0 : STAGE_B;
  extern __shared__ __attribute__((aligned(16))) char lds[];
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
  const int kend = min(g.K, kbeg + g.k_per_split) - kbeg;      // contraction range of this slice, relative to kbeg
  const int nk = (kend + KS - 1) / KS;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int srow = tid >> 1, shalf = tid & 1;
  const int arow = min(srow, g.M - 1 - m0);                   // rows past M: clamp (their products reach unstored rows only)
  const int rows_left = g.M - m0;
  // buffer descriptors over this tile's row panels (wave-uniform), sized so that reads past the tensor return 0
  // (plane operands are padded to whole tiles; kbeg shifts the base so that the scalar k offsets start at 0)
  const long a_bytes = APLANES ? (long)TM * g.lda * 2 - kbeg * 2 : ((long)(min(rows_left, TM) - 1) * g.lda + g.K) * 4;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      APLANES ? (void*)const_cast<_Float16*>(g.Ah + (size_t)m0 * g.lda + kbeg) : (void*)const_cast<float*>(g.A + (size_t)m0 * g.lda),
      0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ral = __builtin_amdgcn_make_buffer_rsrc(
      APLANES ? (void*)const_cast<_Float16*>(g.Al + (size_t)m0 * g.lda + kbeg) : nullptr, 0, APLANES ? (int)a_bytes : 0, 0x00020000);
  const int b_bytes = (int)((long)TN * g.ldb * 2 - kbeg * 2);
  const __amdgpu_buffer_rsrc_t rbh = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(g.Bh + (size_t)n0 * g.ldb + kbeg), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbl = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(g.Bl + (size_t)n0 * g.ldb + kbeg), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint32_t*>(MASK ? g.abits + (size_t)m0 * g.ldbits : nullptr), 0,
      MASK ? (int)((long)min(rows_left, TM) * g.ldbits * 4) : 0, 0x00020000);
  const int voA = APLANES ? (int)((long)srow * g.lda * 2) + shalf * 32 : (int)((long)arow * g.lda * 4) + shalf * 64;
  const int voB = (int)((long)srow * g.ldb * 2) + shalf * 32;
  const int voM = arow * g.ldbits * 4;
  const float mulA = APLANES ? 1.f : (MASK ? g.ascale : 1.f) * g.a_scale[(size_t)(m0 + arow) * g.a_rs];

  u32x4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;
  uint32_t pm = 0xffffffffu, qm = 0xffffffffu;
#define KC_LOAD(x, kt)                                                            \
  do {                                                                            \
    const int ka_ = (kt) * (KS * (APLANES ? 2 : 4)), kb_ = (kt) * (KS * 2);       \
    if (APLANES) {                                                                \
      x##a0 = __builtin_amdgcn_raw_buffer_load_b128(ra, voA, ka_, 0);             \
      x##a1 = __builtin_amdgcn_raw_buffer_load_b128(ra, voA + 16, ka_, 0);        \
      x##a2 = __builtin_amdgcn_raw_buffer_load_b128(ral, voA, ka_, 0);            \
      x##a3 = __builtin_amdgcn_raw_buffer_load_b128(ral, voA + 16, ka_, 0);       \
    } else {                                                                      \
      x##a0 = __builtin_amdgcn_raw_buffer_load_b128(ra, voA, ka_, 0);             \
      x##a1 = __builtin_amdgcn_raw_buffer_load_b128(ra, voA + 16, ka_, 0);        \
      x##a2 = __builtin_amdgcn_raw_buffer_load_b128(ra, voA + 32, ka_, 0);        \
      x##a3 = __builtin_amdgcn_raw_buffer_load_b128(ra, voA + 48, ka_, 0);        \
    }                                                                             \
    x##b0 = __builtin_amdgcn_raw_buffer_load_b128(rbh, voB, kb_, 0);              \
    x##b1 = __builtin_amdgcn_raw_buffer_load_b128(rbh, voB + 16, kb_, 0);         \
    x##b2 = __builtin_amdgcn_raw_buffer_load_b128(rbl, voB, kb_, 0);              \
    x##b3 = __builtin_amdgcn_raw_buffer_load_b128(rbl, voB + 16, kb_, 0);         \
    if (MASK) x##m = __builtin_amdgcn_raw_buffer_load_b32(rm, voM, (kt) * 4, 0);  \
  } while (0)
  auto half8 = [&](char* st, const u32x4& u, const u32x4& w, uint32_t mb, int q, int krem) {
    float e[8] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w),
                  __uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MASK) e[i] = ((mb >> (8 * q + i)) & 1u) ? e[i] : 0.f;
      if (krem < 16) e[i] = (8 * q + i < krem) ? e[i] : 0.f;      // the k tail of the last slab (a neighbouring row's data)
      e[i] *= mulA;
    }
    uint4 hi, lo;
    split8(e, hi, lo);
    const int o = lds_off(srow, 2 * shalf + q);
    *reinterpret_cast<uint4*>(st + o) = hi;
    *reinterpret_cast<uint4*>(st + PLANE_B + o) = lo;
  };
#define KC_STORE(st, x, kt)                                                                     \
  do {                                                                                          \
    const int krem_ = kend - (kt) * KS - shalf * 16;                                            \
    const uint32_t mb_ = MASK ? (x##m >> (shalf * 16)) : 0xffffu;                               \
    const int o0_ = lds_off(srow, 2 * shalf), o1_ = lds_off(srow, 2 * shalf + 1);              \
    if (APLANES) {                                                                              \
      *reinterpret_cast<u32x4*>((st) + o0_) = x##a0;                                            \
      *reinterpret_cast<u32x4*>((st) + o1_) = x##a1;                                            \
      *reinterpret_cast<u32x4*>((st) + PLANE_B + o0_) = x##a2;                                  \
      *reinterpret_cast<u32x4*>((st) + PLANE_B + o1_) = x##a3;                                  \
    } else {                                                                                    \
      half8((st), x##a0, x##a1, mb_, 0, krem_);                                                 \
      half8((st), x##a2, x##a3, mb_, 1, krem_);                                                 \
    }                                                                                           \
    *reinterpret_cast<u32x4*>((st) + 2 * PLANE_B + o0_) = x##b0;                                \
    *reinterpret_cast<u32x4*>((st) + 2 * PLANE_B + o1_) = x##b1;                                \
    *reinterpret_cast<u32x4*>((st) + 3 * PLANE_B + o0_) = x##b2;                                \
    *reinterpret_cast<u32x4*>((st) + 3 * PLANE_B + o1_) = x##b3;                                \
  } while (0)

  const int aoff = (((c & 3) * 32 + wm * 16 + (c >> 2)) << 6);
  const int boff = (((c & 3) * 32 + wn * 16 + (c >> 2)) << 6);
  const int ch0 = ((0 + h) ^ (c & 3)) << 4, ch1 = ((2 + h) ^ (c & 3)) << 4;
  auto compute = [&](const char* st) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ks == 0 ? ch0 : ch1;
      const f16x8 a0 = *reinterpret_cast<const f16x8*>(st + aoff + ch);
      const f16x8 a1 = *reinterpret_cast<const f16x8*>(st + aoff + 512 + ch);
      const f16x8 a0l = *reinterpret_cast<const f16x8*>(st + PLANE_B + aoff + ch);
      const f16x8 a1l = *reinterpret_cast<const f16x8*>(st + PLANE_B + aoff + 512 + ch);
      const f16x8 b0 = *reinterpret_cast<const f16x8*>(st + 2 * PLANE_B + boff + ch);
      const f16x8 b1 = *reinterpret_cast<const f16x8*>(st + 2 * PLANE_B + boff + 512 + ch);
      const f16x8 b0l = *reinterpret_cast<const f16x8*>(st + 3 * PLANE_B + boff + ch);
      const f16x8 b1l = *reinterpret_cast<const f16x8*>(st + 3 * PLANE_B + boff + 512 + ch);
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

  // The prefetch loads are UNCONDITIONAL (past the last slab they re-request it; the data is never stored): behind a
  // branch the compiler has to assume the shorter load queue at the merge point and waits for the just-issued loads too.
  const int last = nk - 1;
  KC_LOAD(p, 0);
  KC_LOAD(q, min(1, last));           // both requests out before the first wait
  KC_STORE(lds, p, 0);
  __syncthreads();
  // iteration kt (even): request slab kt + 2 into p, compute stage 0, park q (slab kt + 1) in stage 1; odd: mirrored
  for (int kt = 0; kt < nk; kt += 2) {
    KC_LOAD(p, min(kt + 2, last));
    compute(lds);
    if (WGS == 3) __syncthreads();
    if (kt + 1 < nk) KC_STORE(lds + S1, q, kt + 1);
    __syncthreads();
    if (kt + 1 < nk) {
      KC_LOAD(q, min(kt + 3, last));
      compute(lds + S1);
      if (WGS == 3) __syncthreads();
      if (kt + 2 < nk) KC_STORE(lds, p, kt + 2);
      __syncthreads();
    }
  }
#undef KC_LOAD
#undef KC_STORE
  split_epilogue<!APLANES, true>(g, acc, lds, m0, n0, split, APLANES ? g.a_scale[0] : 1.f, 1.f);
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
struct Planes { _Float16 *hi, *lo; float* rinv; long rows_p, kp; };

Planes take_rows(Carver& c, long R, long K) {
  Planes p;
  p.rows_p = up(R, 128); p.kp = up(K, KS);
  p.hi = c.take<_Float16>((size_t)p.rows_p * p.kp);
  p.lo = c.take<_Float16>((size_t)p.rows_p * p.kp);
  p.rinv = c.take<float>((size_t)p.rows_p);
  return p;
}
// transposed operand: rows = the source's columns (padded to 128), contraction = the source's rows (padded to 64)
Planes take_trn(Carver& c, long Msrc, long Csrc) {
  Planes p;
  p.rows_p = up(Csrc, 128); p.kp = up(Msrc, 64);
  p.hi = c.take<_Float16>((size_t)p.rows_p * p.kp);
  p.lo = c.take<_Float16>((size_t)p.rows_p * p.kp);
  p.rinv = c.take<float>(64);           // scalar slots: [8] scale, [9] amax word, [10] 1 / scale (convert_trn_scalar)
  return p;
}

int convert_rows(const float* src, long ld, long R, int K, const uint32_t* bits, float ascale, const Planes& p, hipStream_t st) {
  const int vec = al16(src) && (ld % 4 == 0) && (K % 4 == 0);
  hipLaunchKernelGGL(gs_convert_rows_kernel, dim3((unsigned)(p.rows_p / 4)), dim3(256), 0, st, src, ld, (int)R, K,
                     (int)p.rows_p, (int)p.kp, bits, (K + 31) / 32, ascale, vec, p.hi, p.lo, p.rinv);
  return check_launch("gemm_split convert_rows");
}
int tensor_scale(const float* src, long ld, long R, int C, float ascale, float* scal, hipStream_t st);
// planes of src^T with per-source-row multipliers (see gs_convert_trn_kernel)
int convert_trn(const float* src, long ld, long M, int C, const float* row_scale, const float* num, int invert,
                const uint32_t* bits, float ascale, const Planes& p, float* colsum_part, hipStream_t st) {
  const int vec = al16(src) && (ld % 4 == 0) && (C % 4 == 0);
  hipLaunchKernelGGL(gs_convert_trn_kernel, dim3((unsigned)(p.kp / 64), (unsigned)(p.rows_p / 64)), dim3(256), 0, st, src,
                     ld, (int)M, C, (int)p.kp, row_scale, num, invert, bits, (C + 31) / 32, ascale, vec, p.hi, p.lo,
                     colsum_part);
  return check_launch("gemm_split convert_trn");
}
// weights: one scale for the tensor; p.rinv[8] = scale, p.rinv[10] = 1 / scale (what the epilogue reads)
int convert_trn_scalar(const float* src, long ld, long M, int C, const Planes& p, hipStream_t st) {
  if (int rc = tensor_scale(src, ld, M, C, 1.f, p.rinv + 8, st)) return rc;
  return convert_trn(src, ld, M, C, nullptr, p.rinv + 8, 0, nullptr, 1.f, p, nullptr, st);
}

template <typename K>
bool raise_lds(K kernel) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)LDS2_BYTES) == hipSuccess;
}

// KIND 0: A f32 k-contiguous (forward, grad-input); 1: A pre-converted planes (grad-weight)
template <int KIND>
int launch_fused(FusedArgs g, hipStream_t st) {
  static bool attr_set = false;
  static int wgs = 0;             // 0: rule below, 2 / 3: forced (HOISDF_GSPLIT_WG, experiments)
  if (!attr_set) {
    if (const char* e = getenv("HOISDF_GSPLIT_WG")) wgs = atoi(e) == 3 ? 3 : (atoi(e) == 2 ? 2 : 0);
    bool ok;
    if (KIND == 0) ok = raise_lds(gemm_split_kc_kernel<false, false, 2>) && raise_lds(gemm_split_kc_kernel<true, false, 2>);
    else ok = raise_lds(gemm_split_kc_kernel<false, true, 2>);
    if (!ok) {
      set_error("gemm_split: cannot raise the dynamic LDS limit to %u bytes", LDS2_BYTES);
      return HOISDF_ERR_LAUNCH;
    }
    attr_set = true;
  }
  g.tiles_m = cdiv(g.M, TM);
  g.tiles_n = cdiv(g.N, TN);
  g.vecC = al16(g.C) && (g.ldc % 4 == 0) && (g.c_split_stride % 4 == 0);
  const int ntile = g.tiles_m * g.tiles_n;
  const int nwg = g.splitk > 1 ? ntile * 8 * cdiv(g.splitk, 8) : ntile;
  const dim3 grid((unsigned)nwg), block(256);
  // three workgroups per CU (single LDS stage, 168 VGPRs): measured +5...+9 % on the unmasked K = 256 forwards
  // (65536x768x256 151 -> 138 us, x1024x256 190 -> 181 us), -1...-20 % with the sign bitmap or K >= 512 (spills, two
  // barriers per slab) - tools/mb_gsplit.py with HOISDF_GSPLIT_WG=3
  const bool three = wgs == 3 ? g.K <= 512 : (wgs == 0 && KIND == 0 && !g.abits && g.K <= 256);
  const unsigned lb = three ? (unsigned)STAGE_B : LDS2_BYTES;
  if (KIND == 0) {
    if (three) {
      if (g.abits) hipLaunchKernelGGL((gemm_split_kc_kernel<true, false, 3>), grid, block, lb, st, g);
      else hipLaunchKernelGGL((gemm_split_kc_kernel<false, false, 3>), grid, block, lb, st, g);
    } else {
      if (g.abits) hipLaunchKernelGGL((gemm_split_kc_kernel<true, false, 2>), grid, block, lb, st, g);
      else hipLaunchKernelGGL((gemm_split_kc_kernel<false, false, 2>), grid, block, lb, st, g);
    }
  } else {
    hipLaunchKernelGGL((gemm_split_kc_kernel<false, true, 2>), grid, block, LDS2_BYTES, st, g);
  }
  return check_launch("gemm_split");
}

int row_scales(const float* src, long ld, long R, int K, const uint32_t* bits, float ascale, float* scale, hipStream_t st) {
  const int vec = al16(src) && (ld % 4 == 0) && (K % 4 == 0);
  hipLaunchKernelGGL(gs_row_scale_kernel, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, st, src, ld, (int)R, K, bits,
                     (K + 31) / 32, ascale, vec, scale);
  return check_launch("gemm_split row scales");
}
// scal[0] <- scale of the whole tensor, scal[1] = amax word (scratch)
int tensor_scale(const float* src, long ld, long R, int C, float ascale, float* scal, hipStream_t st) {
  uint32_t* amax = reinterpret_cast<uint32_t*>(scal + 1);
  if (hipMemsetAsync(amax, 0, sizeof(uint32_t), st) != hipSuccess) {
    set_error("gemm_split: memset failed");
    return HOISDF_ERR_LAUNCH;
  }
  const int vec = al16(src) && (ld % 4 == 0) && (C % 4 == 0);
  long nb = (R + 3) / 4;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(gs_amax_kernel, dim3((unsigned)nb), dim3(256), 0, st, src, ld, R, C, vec, amax);
  hipLaunchKernelGGL(gs_scalar_scale_kernel, dim3(1), dim3(1), 0, st, amax, ascale, scal);
  return check_launch("gemm_split tensor scale");
}

// contraction slices for grad-weight: ~1024 workgroups, >= 8 slabs each
void plan_split(long Kc, int tiles, int& splitk, int& kper) {
  const int slabs = cdiv(Kc, KS);
  int want = tiles >= 1024 ? 1 : cdiv(1024, tiles);
  if (want > slabs / 8) want = slabs / 8 > 0 ? slabs / 8 : 1;
  kper = cdiv(slabs, want) * KS;
  splitk = cdiv(Kc, kper);
}
}  // namespace

}  // namespace hoisdf

using namespace hoisdf;

extern "C" long hoisdf_linear_split_workspace(long M, int N, int K, int which) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  Carver c{nullptr, 0};
  if (which == 0) {                    // forward: row scales of x, W planes
    c.take<float>((size_t)M);
    take_rows(c, N, K);
  } else if (which == 1) {             // grad-input: row scales of dy, W^T planes
    c.take<float>((size_t)M);
    take_trn(c, N, K);
  } else {                             // grad-weight: row scales of x and dy, scalars, dy^T and x^T planes, partials
    c.take<float>((size_t)M);
    c.take<float>((size_t)M);
    c.take<float>(64);
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
  float* xs = c.take<float>((size_t)M);
  Planes B = take_rows(c, N, K);
  if (int rc = row_scales(x, ldx, M, K, nullptr, 1.f, xs, st)) return rc;
  if (int rc = convert_rows(W, ldw, N, K, nullptr, 1.f, B, st)) return rc;
  FusedArgs g{};
  g.A = x; g.lda = ldx; g.a_scale = xs; g.a_rs = 1;
  g.Bh = B.hi; g.Bl = B.lo; g.ldb = B.kp; g.b_rinv = B.rinv; g.b_rs = 1;
  g.C = y; g.ldc = ldy; g.bias = bias; g.M = (int)M; g.N = N; g.K = K;
  g.act = act; g.drop_p = drop_p; g.inv_keep = 1.f / (1.f - drop_p); g.thresh = drop_threshold(drop_p); g.seed = seed;
  g.bits_out = relu_bits; g.ldbits_out = (N + 31) / 32;
  g.splitk = 1; g.k_per_split = (int)up(K, KS);
  return launch_fused<0>(g, st);
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
  // dx[m][k] = sum_n dy[m][n] W[n][k]: A = dy rows (contraction n contiguous), B = W^T planes ([k][n])
  float* ds = c.take<float>((size_t)M);
  Planes B = take_trn(c, N, K);
  const float ascale = 1.f / (1.f - drop_p);
  // (row maxima over the UNMASKED dy: an upper bound is all the scale needs, and the pass stays a plain streaming read)
  if (int rc = row_scales(dy, lddy, M, N, nullptr, ascale, ds, st)) return rc;
  if (int rc = convert_trn_scalar(W, ldw, N, K, B, st)) return rc;
  FusedArgs g{};
  g.A = dy; g.lda = lddy; g.a_scale = ds; g.a_rs = 1; g.abits = relu_bits; g.ldbits = (N + 31) / 32; g.ascale = ascale;
  g.Bh = B.hi; g.Bl = B.lo; g.ldb = B.kp; g.b_rinv = B.rinv + 10; g.b_rs = 0;
  g.C = dx; g.ldc = lddx; g.M = (int)M; g.N = K; g.K = N;
  g.inv_keep = 1.f;
  g.splitk = 1; g.k_per_split = (int)up(N, KS);
  g.beta = accumulate ? 1 : 0;
  return launch_fused<0>(g, st);
}

extern "C" int hoisdf_linear_bwd_weight_split(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p,
                                              const float* x, int ldx, float* dW, int lddw, float* db, long M, int N, int K,
                                              const float* x_row_scale, const float* dy_row_scale, void* workspace,
                                              long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(dW && (M == 0 || (dy && x)), HOISDF_ERR_INVALID, "linear_bwd_weight_split: null pointer");
  HOISDF_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw == K && M < (1L << 31) && drop_p >= 0.f &&
                     drop_p < 1.f,
                 HOISDF_ERR_INVALID, "linear_bwd_weight_split: bad sizes (a dense dW, lddw == K, is required)");
  HOISDF_REQUIRE(workspace && workspace_bytes >= hoisdf_linear_split_workspace(M, N, K, 2), HOISDF_ERR_INVALID,
                 "linear_bwd_weight_split: workspace too small");
  hipStream_t st = as_stream(stream);
  Carver c{static_cast<char*>(workspace), 0};
  // dW[n][k] = sum_m dy[m][n] x[m][k] = (1 / p) sum_m (dy[m][n] p / sx[m]) (x[m][k] sx[m]),  p = min_m sx[m] sd[m]:
  // x rows are normalised by their own scale sx[m]; dy rows get p / sx[m], which keeps them below the row-normalised
  // dy sd[m] (<= 2^15) and leaves every row with a magnitude proportional to its contribution to dW.  Both operands are
  // written once as transposed hi / lo planes; the GEMM is the plane x plane form of the forward kernel, split along m.
  float* sx = c.take<float>((size_t)M);
  float* sd = c.take<float>((size_t)M);
  float* scal = c.take<float>(64);               // [0] p, [1] 1 / p, [2] 1.0
  Planes A = take_trn(c, M, N), B = take_trn(c, M, K);
  int splitk, kper;
  plan_split(A.kp, cdiv(N, TM) * cdiv(K, TN), splitk, kper);
  float* part = splitk > 1 ? c.take<float>((size_t)splitk * N * K) : nullptr;
  float* cpart = c.take<float>((size_t)(A.kp / 64) * N);
  const float ascale = 1.f / (1.f - drop_p);
  if (!x_row_scale) {
    if (int rc = row_scales(x, ldx, M, K, nullptr, 1.f, sx, st)) return rc;
    x_row_scale = sx;
  }
  if (!dy_row_scale) {
    if (int rc = row_scales(dy, lddy, M, N, relu_bits, ascale, sd, st)) return rc;
    dy_row_scale = sd;
  }
  const uint32_t init[3] = {0x7f7fffffu, 0u, 0x3f800000u};
  // (a host array copied by hipMemcpyAsync would not be stream-ordered with a pageable source: three 4-byte memsets)
  if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(scal), init[0], 1, st) != hipSuccess ||
      hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(scal + 2), init[2], 1, st) != hipSuccess) {
    set_error("linear_bwd_weight_split: memset failed");
    return HOISDF_ERR_LAUNCH;
  }
  long nb = (M + 255) / 256;
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(gs_min_product_kernel, dim3((unsigned)nb), dim3(256), 0, st, x_row_scale, dy_row_scale, M,
                     reinterpret_cast<uint32_t*>(scal));
  hipLaunchKernelGGL(gs_min_product_finish_kernel, dim3(1), dim3(1), 0, st, scal);
  if (int rc = convert_trn(dy, lddy, M, N, x_row_scale, scal, 1, relu_bits, ascale, A, db ? cpart : nullptr, st)) return rc;
  if (int rc = convert_trn(x, ldx, M, K, x_row_scale, nullptr, 0, nullptr, 1.f, B, nullptr, st)) return rc;
  FusedArgs g{};
  g.Ah = A.hi; g.Al = A.lo; g.lda = A.kp; g.a_scale = scal; g.a_rs = 0;
  g.Bh = B.hi; g.Bl = B.lo; g.ldb = B.kp; g.b_rinv = scal + 2; g.b_rs = 0;
  g.M = N; g.N = K; g.K = (int)A.kp; g.inv_keep = 1.f;
  g.splitk = splitk; g.k_per_split = kper;
  if (splitk > 1) { g.C = part; g.ldc = K; g.c_split_stride = (long)N * K; }
  else { g.C = dW; g.ldc = lddw; }
  if (int rc = launch_fused<1>(g, st)) return rc;
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
