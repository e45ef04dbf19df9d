// fp32 linear layers EMULATED on the bf16 MFMA pipe ("bf16x3"): forward and grad-input of common/nets/layer.py:168-201
// (MLP), common/nets/transformer.py:286-302 (in / out projections, feed-forward), main/model.py:56-90 (input MLPs, heads).
//
// Every f32 operand is split EXACTLY into three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 significand bits; bf16 has the f32
// exponent range, so - unlike an f16 hi / lo pair - nothing has to be scaled and nothing is lost: x0 = bf16(x),
// x1 = bf16(x - x0), x2 = bf16(x - x0 - x1), every subtraction exact).  A product x y is accumulated in f32 from six
// v_mfma_f32_32x32x16_bf16 products, x0y0 + x0y1 + x1y0 + x1y1 + x0y2 + x2y0 (each bf16 x bf16 product is exact in f32); the
// three dropped terms are <= 2^-24 |x y|, below the rounding of an f32 fused multiply-add.  Against fp64 the result has the
// error of an f32 GEMM (measured next to the exact-f32 MFMA kernel: tools/ubench/gemm_emu_lab.hip, tests/test_gpu_emu.py),
// while the bf16 pipe runs 16 x the f32 MFMA rate: 2.67 x after six products.
//
//   A (activations x, or dy) is read as f32, k-contiguous, and split on its way into LDS (thread = tile row; the forward's
//   ReLU / dropout sign bitmap and 1 / keep are applied to dy before the split).  B (the weight) is pre-split ONCE per weight
//   update into a "slab image": for column tile tn (128 output columns), slab s (16 k), plane p, k-chunk c (8 k), row r the
//   16 bytes at ((((tn * nslab + s) * 3 + p) * 2 + c) * 128 + r) * 16 - exactly the LDS image of the slab, so staging it is
//   three fully coalesced 16-byte loads and three ds_write_b128 per thread (hoisdf_linear_emu_prepare; transposed for grad-input).
//   LDS image of a plane slab: [chunk][row][16 B]: the MFMA fragment read (32 consecutive rows of one chunk per half-wave,
//   ds_read_b128) and the staging write (consecutive rows) are both bank-conflict free without padding.
// Tile 256 x 128, 4 waves as 2 x 2, wave tile 128 x 64 = 4 x 2 MFMA blocks (128 accumulators), 16-deep slabs double-buffered
// in LDS (72 KB), two workgroups per CU (<= 256 VGPRs); one barrier per slab; the next slab is converted / parked and the one
// after it requested at the top of every slab.  Epilogue = gemm.hip's (bias, ReLU, dropout, 1-bit sign map, accumulate-into,
// LDS-transposed 16-byte stores).
#include <stdlib.h>

#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>

#include "common.h"

namespace hoisdf {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int TM = 256, TN = 128, KS = 16, NT = 256;
constexpr int WN = TN / 2, NJ = WN / 32;
constexpr int A_U4 = 3 * 2 * TM, B_U4 = 3 * 2 * TN, STAGE_U4 = A_U4 + B_U4;
constexpr int NB = B_U4 / NT;
constexpr unsigned LDS_BYTES = 2u * STAGE_U4 * 16u;        // 73 728

// exact three-way split (native ext vectors only: arrays of HIP's uint4 / float4 structs end up in scratch)
#define SPLIT1(x, i)                             \
  do {                                           \
    const __bf16 a_ = (__bf16)(x);               \
    const float r1_ = (x) - (float)a_;           \
    const __bf16 b_ = (__bf16)r1_;               \
    const float r2_ = r1_ - (float)b_;           \
    p0[i] = a_; p1[i] = b_; p2[i] = (__bf16)r2_; \
  } while (0)
__device__ __forceinline__ void split3x8(const float4 u, const float4 w, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
  SPLIT1(u.x, 0); SPLIT1(u.y, 1); SPLIT1(u.z, 2); SPLIT1(u.w, 3);
  SPLIT1(w.x, 4); SPLIT1(w.y, 5); SPLIT1(w.z, 6); SPLIT1(w.w, 7);
}

struct EmuArgs {
  const float* A; long lda;                 // [M][lda] f32, k-contiguous
  const u32x4* Bimg;                        // slab image of the weight operand (rows = output columns)
  float* C; int ldc;
  const float* bias;
  const uint32_t* abits; int ldbits; float ascale;      // sign bitmap of A ([M][ceil(K / 32)]) and 1 / keep (grad-input)
  uint32_t* bits_out; int ldbits_out;
  int M, N, K;                              // output rows, output columns, contraction length
  int act; float drop_p, inv_keep; uint32_t thresh; uint64_t seed;
  int tiles_m, tiles_n, vecC, beta;
  QkvPlanes qkv;                            // .on: the output tile goes into attention planes instead of C (common.h)
  // f16x2 form: row magnitudes of A (common.h: one word per row, bits of max |A[row][:]| or an upper bound), the image's {scale, 1 / scale}
  const uint32_t* a_amax; const float* b_scale;
  uint32_t* amax_out;                       // row magnitudes of C (any form; zero on entry; null = not wanted)
  uint32_t* head_out; int head_L, head_nb;  // head magnitudes of C (common.h: word[(col / 64) * head_nb + row / head_L]; null = not wanted)
};
}  // namespace

// ---- weight -> slab image.  transpose = 0: image row n, contraction k = W[n][k] (forward);  1: image row k, contraction
// n = W[n][k] (grad-input: dx = dy . W).  One thread per (tile, slab, chunk, row): 8 source values -> 3 x 16 bytes.
__device__ __forceinline__ void emu_prep_weight_unit(const float* __restrict__ W, int ldw, int R, int Kc, int transpose, int nslab,
                                                     long idx, u32x4* __restrict__ img) {
  const int r = (int)(idx % TN);
  const int c = (int)((idx / TN) % 2);
  const int s = (int)((idx / (2 * TN)) % nslab);
  const int tn = (int)(idx / ((long)2 * TN * nslab));
  const int row = tn * TN + r;
  const int k0 = s * KS + c * 8;
  float e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k0 + i;
    float v = 0.f;
    if (row < R && k < Kc) v = transpose ? W[(size_t)k * ldw + row] : W[(size_t)row * ldw + k];
    e[i] = v;
  }
  bf16x8 p0, p1, p2;
  split3x8(make_float4(e[0], e[1], e[2], e[3]), make_float4(e[4], e[5], e[6], e[7]), p0, p1, p2);
  const size_t base = ((size_t)(tn * nslab + s) * 3) * 2 * TN;
  img[base + (0 * 2 + c) * TN + r] = __builtin_bit_cast(u32x4, p0);
  img[base + (1 * 2 + c) * TN + r] = __builtin_bit_cast(u32x4, p1);
  img[base + (2 * 2 + c) * TN + r] = __builtin_bit_cast(u32x4, p2);
}

__global__ __launch_bounds__(256) void emu_prep_weight_kernel(const float* __restrict__ W, int ldw, int R, int Kc, int transpose,
                                                              int nslab, long total, u32x4* __restrict__ img) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx < total) emu_prep_weight_unit(W, ldw, R, Kc, transpose, nslab, idx, img);
}

// many images in one launch (all weights of a model after an optimizer step): the block finds its item in the table by its
// first-block offsets (ascending)
__global__ __launch_bounds__(256) void emu_prep_weight_batch_kernel(const hoisdf_emu_prep_item* __restrict__ items, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_block <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const hoisdf_emu_prep_item it = items[lo];
  const int R = it.transpose ? it.K : it.N, Kc = it.transpose ? it.N : it.K;
  const int nslab = ((Kc + KS - 1) / KS);
  const long total = (long)((R + TN - 1) / TN) * nslab * 2 * TN;
  const long idx = ((long)blockIdx.x - it.first_block) * 256 + threadIdx.x;
  if (idx < total) emu_prep_weight_unit(it.W, it.ldw, R, Kc, it.transpose, nslab, idx, static_cast<u32x4*>(it.image));
}

// C-tile epilogue shared by all main-loop forms (tile TM_ x TN_, 2 x 2 waves, wave tile 128 x 32 NJ_): bias, ReLU, dropout, 1-bit
// sign map, accumulate-into, LDS-transposed 16-byte stores
template <int TM_, int TN_, int NJ_>
__device__ __forceinline__ void emu_epilogue(const EmuArgs& g, f32x16 (&acc)[4][NJ_], u32x4* lds, int m0, int n0, int wm, int wn, int wave,
                                             int lane, int l31, int kh, float post_scale, const float* row_post = nullptr) {
  constexpr int WN_ = TN_ / 2;
  static_assert(WN_ == NJ_ * 32, "wave tile");
  if (row_post) {                              // (f16x2) one factor per output ROW: the row's operand scale, the weight's, 1 / keep - from LDS
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ps = *reinterpret_cast<const f32x4*>(row_post + wm * 128 + i * 32 + 8 * q + 4 * kh);
#pragma unroll
        for (int j = 0; j < NJ_; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] *= ps[e];
      }
  } else if (post_scale != 1.f) {                     // (grad-input, rotated form) 1 / keep of the forward's dropout, once per element; (f16x2) the operand scales
#pragma unroll
    for (int j = 0; j < NJ_; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= post_scale;
  }
  // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); all waves are
  // past the main loop's last barrier, the staging buffer is free
  const int rbase = m0 + wm * 128 + 4 * kh;
  const int cbase = n0 + wn * WN_ + l31;
#pragma unroll
  for (int j = 0; j < NJ_; ++j) {
    const int col = cbase + j * 32;
    const float bv = (g.bias != nullptr && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bv;
        if (g.act == 1) v = fmaxf(v, 0.f);
        acc[i][j][r] = v;
      }
  }
  if (g.drop_p > 0.f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
        const uint32_t rk = drop_rowkey(g.seed, (uint32_t)row);
#pragma unroll
        for (int j = 0; j < NJ_; ++j) acc[i][j][r] *= drop_scale(rk, (uint32_t)(cbase + j * 32), g.thresh, g.inv_keep);
      }
  }
  if (g.head_out) {
    // head magnitudes (common.h): each 64-column group of the wave's 128-row sub-tile -> the word(s) of the sample(s) its rows belong to
    // (one sample when head_L % 128 == 0; otherwise every sample the 128 rows touch gets the whole sub-tile's maximum: an upper bound)
    const int row0 = m0 + wm * 128;
    if (row0 < g.M) {
      const int b0 = row0 / g.head_L, b1 = (min(row0 + 127, g.M - 1)) / g.head_L;
#pragma unroll
      for (int hh = 0; hh < NJ_ / 2; ++hh) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, __builtin_fabsf(acc[i][2 * hh + j][r]));
        uint32_t mb = group_max_u32<64>(__builtin_bit_cast(uint32_t, m));
        const int grp = (n0 + wn * WN_ + hh * 64) >> 6;
        if (lane == 0 && n0 + wn * WN_ + hh * 64 < g.N)
          for (int b = b0; b <= b1; ++b) atomicMax(g.head_out + (size_t)grp * g.head_nb + b, mb);
      }
    }
  }
  if (g.qkv.on) {
    // attention-plane output (common.h QkvPlanes): every 128 x 64 part of the wave's sub-tile is 128 consecutive tokens of one sample x
    // one head of one part; per 32-row block through the wave-private LDS slice: row planes as 8 lanes x 16 bytes per token and piece,
    // transposed value planes as one d per lane, 8 consecutive tokens (16 bytes) per store
    constexpr int ES = 64 + 4;
    float* w = reinterpret_cast<float*>(lds) + wave * (32 * ES);
    const int row0 = m0 + wm * 128, cw = n0 + wn * WN_;
    if (row0 + 128 > g.M || cw + WN_ > g.N) return;           // (never: the launcher takes whole wave tiles only)
    const int b = row0 / g.qkv.L, s0 = row0 - b * g.qkv.L;
#pragma unroll
    for (int hh = 0; hh < NJ_ / 2; ++hh) {
      const int colg = g.qkv.col0 + cw + hh * 64;
      const int part = colg / g.qkv.E, head = (colg - part * g.qkv.E) >> 6;
      const size_t bh = (size_t)b * g.qkv.H + head;
      const float sc = part == 0 ? g.qkv.qscale : 1.f;
      __bf16* const r0 = static_cast<__bf16*>(g.qkv.r[part][0]);
      __bf16* const r1 = static_cast<__bf16*>(g.qkv.r[part][1]);
      __bf16* const r2 = static_cast<__bf16*>(g.qkv.r[part][2]);
      const bool trn = part == 2 && g.qkv.vt[0] != nullptr;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2) + 4 * kh) * ES + j * 32 + l31] = acc[i][2 * hh + j][r] * sc;
        if (r0) {
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int rr = p * 8 + (lane >> 3), cc = (lane & 7) * 8;
            const float4 u = *reinterpret_cast<const float4*>(w + rr * ES + cc);
            const float4 v = *reinterpret_cast<const float4*>(w + rr * ES + cc + 4);
            bf16x8 p0, p1, p2;
            split3x8(u, v, p0, p1, p2);
            const size_t o = ((size_t)bh * g.qkv.Lp + s0 + i * 32 + rr) * 64 + cc;
            *reinterpret_cast<bf16x8*>(r0 + o) = p0;
            *reinterpret_cast<bf16x8*>(r1 + o) = p1;
            *reinterpret_cast<bf16x8*>(r2 + o) = p2;
          }
        }
        if (trn) {
          __bf16* const t0 = static_cast<__bf16*>(g.qkv.vt[0]);
          __bf16* const t1 = static_cast<__bf16*>(g.qkv.vt[1]);
          __bf16* const t2 = static_cast<__bf16*>(g.qkv.vt[2]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float* c0 = w + (8 * q) * ES + lane;
            const float4 u = make_float4(c0[0], c0[ES], c0[2 * ES], c0[3 * ES]);
            const float4 v = make_float4(c0[4 * ES], c0[5 * ES], c0[6 * ES], c0[7 * ES]);
            bf16x8 p0, p1, p2;
            split3x8(u, v, p0, p1, p2);
            const size_t o = ((size_t)bh * 64 + lane) * g.qkv.Lp + s0 + i * 32 + 8 * q;
            *reinterpret_cast<bf16x8*>(t0 + o) = p0;
            *reinterpret_cast<bf16x8*>(t1 + o) = p1;
            *reinterpret_cast<bf16x8*>(t2 + o) = p2;
          }
        }
      }
    }
    return;
  }
  const bool full = (m0 + TM_ <= g.M) && (n0 + TN_ <= g.N);
  const bool stream_c = !g.beta && (long)g.M * g.N >= (16L << 20);
  if (full && g.vecC) {
    // one row of blocks (32 x WN_) per wave at a time through a wave-private LDS slice, read back row-wise: one
    // global_store_dwordx4 covers complete 256-byte row segments
    constexpr int ES = WN_ + 4, LPR = WN_ / 4, RPI = 64 / LPR;
    float* w = reinterpret_cast<float*>(lds) + wave * (32 * ES);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < NJ_; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2) + 4 * kh) * ES + j * 32 + l31] = acc[i][j][r];
      if (g.amax_out) {
        // row magnitudes of C (common.h): lane l folds the (l >> 5) half of row l & 31 of the 32-row block parked in its wave's LDS slice
        // (WN_ / 8 16-byte reads + as many v_max3 with |.| operands), the two halves meet through one exchange, lanes 0-31 publish
        // (the first form - 4 DPP steps per stored 16-byte piece - cost 700 issue slots per wave tile, this one ~120)
        const float* rp = w + l31 * ES + kh * (WN_ / 2);
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < WN_ / 8; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(rp + 4 * q);
          m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t.x)), __builtin_fmaxf(__builtin_fabsf(t.y), __builtin_fmaxf(__builtin_fabsf(t.z), __builtin_fabsf(t.w))));
        }
        uint32_t mb = __builtin_bit_cast(uint32_t, m);
        mb = max(mb, (uint32_t)__shfl_xor((int)mb, 32, 64));
        if (kh == 0) atomicMax(g.amax_out + (m0 + wm * 128 + i * 32 + l31), mb);
      }
#pragma unroll
      for (int p = 0; p < 32 / RPI; ++p) {
        const int rr = p * RPI + lane / LPR, cc = (lane % LPR) * 4;
        float4 v = *reinterpret_cast<const float4*>(w + rr * ES + cc);
        float4* cp = reinterpret_cast<float4*>(g.C + (size_t)(m0 + wm * 128 + i * 32 + rr) * g.ldc + n0 + wn * WN_ + cc);
        if (g.beta) {
          const float4 old = *cp;
          v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
        }
        // an output too large to stay in the L2s (>= 64 MB) is streamed past them: -0.12 ms per train step, two same-box pairs
        typedef float v4f_ __attribute__((ext_vector_type(4)));
#ifdef H2_ABL_NOSTORE                 /* (tools/ablate_h2.sh: how much of the kernel is the output's way to HBM?  never true at run time) */
        if (g.seed != 0x5eed5eed5eedull) continue;
#endif
        if (stream_c) __builtin_nontemporal_store(v4f_{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_*>(cp));
        else *cp = v;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NJ_; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2), col = cbase + j * 32;
          if (row < g.M && col < g.N) {
            float* cp = g.C + (size_t)row * g.ldc + col;
            *cp = g.beta ? *cp + acc[i][j][r] : acc[i][j][r];
          }
        }
    if (g.amax_out) {                         // (edge tiles / unaligned C: per row over the 32 lanes that hold its columns)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
          uint32_t mb = 0u;
#pragma unroll
          for (int j = 0; j < NJ_; ++j) if (cbase + j * 32 < g.N) mb = max(mb, mag_bits(acc[i][j][r]));
          mb = group_max_u32<32>(mb);
          if (l31 == 0 && row < g.M) atomicMax(g.amax_out + row, mb);
        }
    }
  }
  if (g.bits_out) {
    // lanes 0-31 hold 32 consecutive columns of one row, lanes 32-63 of the row 4 below: one ballot is two mask words.
    // Each lane collects the words of "its" rows (lane and lane + 64 of the wave's 128-row sub-tile) and writes them once.
    uint32_t wd[2][NJ_];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int j = 0; j < NJ_; ++j) wd[hh][j] = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (i & 1) * 32 + (r & 3) + 8 * (r >> 2);      // row within a 64-row half, as held by lanes 0-31
#pragma unroll
        for (int j = 0; j < NJ_; ++j) {
          const unsigned long long q = __ballot(acc[i][j][r] > 0.f);
          if (lane == rl) wd[i >> 1][j] = (uint32_t)q;
          if (lane == rl + 4) wd[i >> 1][j] = (uint32_t)(q >> 32);
        }
      }
    const int wcol = (n0 + wn * WN_) >> 5;
    const int nvalid = g.N - (n0 + wn * WN_);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int row = m0 + wm * 128 + hh * 64 + lane;
      if (row < g.M) {
#pragma unroll
        for (int j = 0; j < NJ_; ++j) {
          const int nv = nvalid - 32 * j;
          if (nv > 0) g.bits_out[(size_t)row * g.ldbits_out + wcol + j] = nv >= 32 ? wd[hh][j] : (wd[hh][j] & ((1u << nv) - 1u));
        }
      }
    }
  }
}

template <bool MASK>
__global__ __launch_bounds__(NT, 2) void emu_kc_kernel(EmuArgs g) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int t = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int nslab = (g.K + KS - 1) / KS;
  const int last = nslab - 1;

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: thread = row of the A tile (rows past M re-read the last row: their products only reach rows that are never
  // stored); B image pieces tid + 256 q.  All loads are unconditional (a load behind a branch makes hipcc assume the shorter
  // queue at the merge and wait for everything): the k tail is handled by clamping the address and zeroing the value.
  const int arow_i = min(m0 + tid, g.M - 1);
  const float* arow = g.A + (size_t)arow_i * g.lda;
  const uint32_t* mrow = MASK ? g.abits + (size_t)arow_i * g.ldbits : nullptr;
  const u32x4* bsrc = g.Bimg + (size_t)tn * nslab * B_U4 + tid;
  const int kmax4 = g.K - 4;                       // K is a multiple of 4 (checked by the host)
  float4 ra[4];
  u32x4 rb[NB];
  uint32_t rm = 0xffffffffu;
#define LOAD_SLAB(sl)                                                                                                  \
  do {                                                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                      \
        ra[q] = *reinterpret_cast<const float4*>(arow + min((sl) * KS + q * 4, kmax4));                                \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) rb[q] = bsrc[(size_t)(sl) * B_U4 + q * NT];                          \
    if (MASK) rm = mrow[(sl) >> 1];                                                                                    \
  } while (0)
#define STORE_SLAB(st, sl)                                                                                             \
  do {                                                                                                                 \
    const int krem_ = g.K - (sl) * KS;                      /* valid k in this slab (>= 16 except in the last one) */  \
    const uint32_t mb_ = MASK ? (rm >> (((sl) & 1) * 16)) : 0xffffu;                                                   \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                    \
      float4 v_ = ra[q];                                                                                               \
      if (MASK) {                                                                                                      \
        v_.x = ((mb_ >> (4 * q + 0)) & 1u) ? v_.x * g.ascale : 0.f;                                                    \
        v_.y = ((mb_ >> (4 * q + 1)) & 1u) ? v_.y * g.ascale : 0.f;                                                    \
        v_.z = ((mb_ >> (4 * q + 2)) & 1u) ? v_.z * g.ascale : 0.f;                                                    \
        v_.w = ((mb_ >> (4 * q + 3)) & 1u) ? v_.w * g.ascale : 0.f;                                                    \
      }                                                                                                                \
      if (4 * q >= krem_) v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                        \
      ra[q] = v_;                                                                                                      \
    }                                                                                                                  \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                                    \
      bf16x8 p0, p1, p2;                                                                                               \
      split3x8(ra[2 * c], ra[2 * c + 1], p0, p1, p2);                                                                  \
      (st)[(0 * 2 + c) * TM + tid] = __builtin_bit_cast(u32x4, p0);                                                    \
      (st)[(1 * 2 + c) * TM + tid] = __builtin_bit_cast(u32x4, p1);                                                    \
      (st)[(2 * 2 + c) * TM + tid] = __builtin_bit_cast(u32x4, p2);                                                    \
    }                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) (st)[A_U4 + tid + q * NT] = rb[q];                                   \
  } while (0)

  LOAD_SLAB(0);
  STORE_SLAB(lds, 0);
  LOAD_SLAB(min(1, last));
  __syncthreads();

  for (int s = 0; s < nslab; ++s) {
    const u32x4* st = lds + (s & 1) * STAGE_U4;
    u32x4* nx = lds + ((s + 1) & 1) * STAGE_U4;
    const u32x4* sa = st + wm * 128 + l31;
    const u32x4* sb = st + A_U4 + wn * WN + l31;
    bf16x8 b0[NJ], b1[NJ], b2[NJ], a[4];
#define RD_B(dst, p) _Pragma("unroll") for (int j = 0; j < NJ; ++j) dst[j] = __builtin_bit_cast(bf16x8, sb[((p) * 2 + kh) * TN + j * 32])
#define RD_A(p) _Pragma("unroll") for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, sa[((p) * 2 + kh) * TM + i * 32])
#define MM1(bx) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFB(a[i], bx[j], acc[i][j])
    RD_B(b0, 0); RD_A(2);
    // the next slab: registers -> the other stage (masked, converted); the slab after it -> registers (a whole slab of MFMAs
    // to land).  Pinned here: hipcc otherwise sinks the loads to the end of the loop body.  Past the end the last slab is
    // simply staged again into the stage nobody reads any more.
    STORE_SLAB(nx, min(s + 1, last));
    LOAD_SLAB(min(s + 2, last));
    __builtin_amdgcn_sched_barrier(0);
    MM1(b0);                                   // x2 y0            (small terms first)
    RD_B(b1, 1); RD_A(1);
    MM1(b1); MM1(b0);                          // x1 y1, x1 y0
    RD_B(b2, 2); RD_A(0);
    MM1(b2); MM1(b1); MM1(b0);                 // x0 y2, x0 y1, x0 y0
    __syncthreads();
  }
#undef LOAD_SLAB
#undef STORE_SLAB
#undef RD_A
#undef RD_B
#undef MM1

  emu_epilogue<TM, TN, NJ>(g, acc, lds, m0, n0, wm, wn, wave, lane, l31, kh, 1.f);
}


// ---- main loop, second form ("rotated", hand-interleaved, line-coalesced activation loads): same tile, product order and
// epilogue as emu_kc_kernel (forward results are bit identical), but
//  * the six product groups of a slab are rotated by half a slab against the barrier: a phase = [x0 y2, x0 y1, x0 y0 of slab
//    s - 1 | x2 y0, x1 y1, x1 y0 of slab s], so the 24 MFMAs right behind the barrier take fragments that were read BEFORE it and
//    every fragment read of slab s is issued 4 ... 24 MFMAs ahead of its first use;
//  * the staging of slab s + 1 (16 f32 per thread -> three bf16 planes -> the other stage, plus the weight image pieces) is
//    cut into units of <= 5 VALU / one LDS or global instruction, and every unit is pinned behind ONE MFMA of the same wave:
//    on this part a wave's VALU work hides under its OWN MFMAs only (profiles/r03_mfma_valu_overlap.txt), and left alone hipcc
//    emits [convert + write everything | 48 MFMAs] - with one LDS array it even has to (every fragment read may alias the
//    staging writes that precede it in program order), which is why the two stages are two distinct __shared__ objects here;
//  * the activation tile is loaded with FOUR LANES PER ROW (lane = row l / 4 of a 16-row group, 16-byte quad l % 4 of the
//    slab's 64 bytes; four such items per thread): a wave instruction touches 16 cache lines instead of 64.  With thread =
//    row (the first form) every global_load_dwordx4 asks the vector memory pipe for 64 different lines, 16 bytes of each: the
//    ablations of round 4 (profiles/r04_kc2_ablation.txt) showed the loop running at 285 - 338 TF without staging and at
//    130 - 170 with the loads and LDS writes but WITHOUT any conversion arithmetic - the address / tag path, not the VALU,
//    was the limiter.  A quad converts to 8 bytes per plane (ds_write_b64); rows of the second k-chunk are stored with
//    bit 2 of the row flipped so that a 16-lane group's 4 rows x 2 chunks x 2 halves cover all 32 banks once.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef EMU_ABL
#define EMU_ABL 0
#endif
template <bool MASK, bool KTAIL>
__global__ __launch_bounds__(NT, 2) void emu_kc2_kernel(EmuArgs g) {
  __shared__ __attribute__((aligned(16))) u32x4 st0[STAGE_U4];
  __shared__ __attribute__((aligned(16))) u32x4 st1[STAGE_U4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int t = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int nslab = (g.K + KS - 1) / KS;
  const int last = nslab - 1;

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging roles: item i (0..3) of this thread = row i * 64 + wave * 16 + lane / 4 of the tile, quad qd = lane % 4 of the slab
  // (k = 4 qd .. 4 qd + 3; chunk c = qd / 2, half qd % 2 of the chunk's 16 bytes).  Rows past M read as zero (their products
  // only reach rows that are never stored).
  const int rl = lane >> 2, qd = lane & 3, cq = qd >> 1;
  // buffer descriptors (wave-uniform) over the tile's row panel, its sign-bitmap rows and the column tile's weight image: the
  // loads are buffer_load (32-bit per-lane offset in ONE register + scalar slab offset) - with flat addressing hipcc keeps a
  // 64-bit address pair per item alive across the loop and spills
  // (four descriptors each, one per item = 64-row quarter of the tile: the per-lane offset is the SAME register for all four,
  // and rows past M read as zero through the quarter's record count - no clamping, no per-item offset registers)
  const int rows_in = min(TM, g.M - m0);
  __amdgpu_buffer_rsrc_t rsa[4], rsm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rows_q = max(min(rows_in - i * 64, 64), 0);     // valid rows of this quarter
#ifdef H2_ABL_NOLOADA                 /* (tools/ablate_h2.sh: every tile reads the FIRST tile's rows - the same loads, served by the caches) */
    rsa[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + ((size_t)i * 64) * g.lda), 0,
                                               rows_q > 0 ? (int)((((long)rows_q - 1) * g.lda + g.K) * 4) : 0, 0x00020000);
#else
    rsa[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + ((size_t)m0 + i * 64) * g.lda), 0,
                                               rows_q > 0 ? (int)((((long)rows_q - 1) * g.lda + g.K) * 4) : 0, 0x00020000);
#endif
    rsm[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(MASK ? g.abits + ((size_t)m0 + i * 64) * g.ldbits : nullptr), 0,
                                               MASK ? (int)((long)rows_q * g.ldbits * 4) : 0, 0x00020000);
  }
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(g.Bimg + (size_t)tn * nslab * B_U4), 0, nslab * B_U4 * 16, 0x00020000);
  const int aoff = (int)((((long)wave * 16 + rl) * g.lda + 4 * qd) * 4);
  const int moff = (int)(((long)wave * 16 + rl) * g.ldbits * 4);
  // 8-byte LDS slot of item 0, plane 0: unit (chunk cq, row ^ (cq << 2)), half qd & 1; item i adds 128 slots, plane p 2 * 2 * TM.
  // ds_write_b64 is served in contiguous 16-lane groups over 32 banks (128 bytes): a group = 4 rows x 4 quads; flipping bit 2 of
  // the row for the second chunk puts its 4 rows x 16 bytes into the other half of the bank row (PMC: SQ_LDS_BANK_CONFLICT back at
  // the first form's level; with bit 3 flipped the two chunks collided, + 25 % LDS cycles)
  const int wslot = 2 * (cq * TM + ((wave * 16 + rl) ^ (cq << 2))) + (qd & 1);
  const int aread = (wm * 128 + l31) ^ (kh << 2);              // fragment rows follow the same row flip (chunk = kh)
  const int kq = g.K - 4 * qd;                                   // quad valid in slab sl iff sl * 16 < kq
  f32x2 rp[8], fu[8];                                            // the thread's 4 quads of the slab being staged, as pairs (+ unpacked planes)
  uint32_t t0[8], t1[8], t2[8];                                  // their three bf16 planes (two values per register)
  u32x4 rb[NB];
  uint32_t rm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, mb = 0xfu;
  bool kin = true;
#define NOP_ ((void)0)
// ---- staging units.  LDGA(i, sl): the quad of item i in slab sl -> rp[2 i], rp[2 i + 1] (+ its sign-bitmap word)
#if EMU_ABL == 2 || EMU_ABL == 4
#define LDGA(i, sl) NOP_
#define LDGB(q, sl) NOP_
#elif EMU_ABL == 5 || EMU_ABL == 7          /* no weight-image loads (7: nor their stage writes) */
#define LDGA(i, sl)                                                                                                    \
  do {                                                                                                                 \
    const int k0_ = min((sl), last) * KS;                                                                              \
    const f32x4 v_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa[i], aoff, k0_ * 4, 0));       \
    rp[2 * (i)] = f32x2{v_[0], v_[1]}; rp[2 * (i) + 1] = f32x2{v_[2], v_[3]};                                          \
    if (MASK) rm[i] = __builtin_amdgcn_raw_buffer_load_b32(rsm[i], moff, (k0_ >> 5) * 4, 0);                           \
  } while (0)
#define LDGB(q, sl) NOP_
#elif EMU_ABL == 6                          /* no activation loads */
#define LDGA(i, sl) NOP_
#define LDGB(q, sl) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rsb, (tid + (q) * NT) * 16, min((sl), last) * (B_U4 * 16), 0)
#else
#define LDGA(i, sl)                                                                                                    \
  do {                                                                                                                 \
    const int k0_ = min((sl), last) * KS;                      /* (uniform) a pad slab re-reads the last one */        \
    const f32x4 v_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa[i], aoff, k0_ * 4, 0));       \
    rp[2 * (i)] = f32x2{v_[0], v_[1]}; rp[2 * (i) + 1] = f32x2{v_[2], v_[3]};                                          \
    if (MASK) rm[i] = __builtin_amdgcn_raw_buffer_load_b32(rsm[i], moff, (k0_ >> 5) * 4, 0);                           \
  } while (0)
#define LDGB(q, sl) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rsb, (tid + (q) * NT) * 16, min((sl), last) * (B_U4 * 16), 0)
#endif
// the conversion of pair p = 2 i + j (item i, half j of its quad) in four units (U1 .. U4) or two (CV1 = U1 + U2, CV2 = U3 + U4):
//   U1  sign bitmap (bit -> all-ones / zero word -> and; the 1 / keep factor is applied once, in the epilogue) and k tail;
//       first plane = bf16(v) (v_cvt_pk_bf16_f32), unpacked again    U2  first residual (one packed subtract)
//   U3  second plane, unpacked                                        U4  second residual, third plane
// UI(i, sl): per-item scalars of slab sl (the quad's four sign bits; is the quad inside K)
#define UI(i, sl) do { if (MASK) mb = rm[i] >> ((((sl) & 1) << 4) + 4 * qd); if (KTAIL) kin = (sl) * KS < kq; } while (0)
#if EMU_ABL == 1
#define U1(p) do { t0[p] = __builtin_bit_cast(uint32_t, rp[p].x); t1[p] = __builtin_bit_cast(uint32_t, rp[p].y); t2[p] = t0[p]; } while (0)
#define U2(p) NOP_
#define U3(p) NOP_
#define U4(p) NOP_
#else
#define PK_SUB(d, a, b) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b))
#define U1(p)                                                                                                          \
  do {                                                                                                                 \
    f32x2 v_ = rp[p];                                                                                                  \
    if (MASK) {           /* bit -> all-ones / zero word (v_bfe_i32) -> v_and: written as asm, hipcc (ROCm 7.2) turns the plain   */ \
      float xa_, xb_;     /* expression into compare + select, and MISCOMPILES the two-element form (the y lane reads x)        */ \
      asm("v_and_b32 %0, %1, %2" : "=v"(xa_) : "v"(__builtin_amdgcn_sbfe((int)mb, 2 * ((p) & 1), 1)), "v"(v_.x));       \
      asm("v_and_b32 %0, %1, %2" : "=v"(xb_) : "v"(__builtin_amdgcn_sbfe((int)mb, 2 * ((p) & 1) + 1, 1)), "v"(v_.y));   \
      v_ = f32x2{xa_, xb_};                                                                                            \
    }                                                                                                                  \
    if (KTAIL) { if (!kin) v_ = f32x2{0.f, 0.f}; }             /* K is a multiple of 4: a quad is all in or all out */ \
    const uint32_t h_ = __builtin_bit_cast(uint32_t, __builtin_convertvector(v_, bf16x2));                             \
    t0[p] = h_; rp[p] = v_;                                                                                            \
    fu[p] = f32x2{__builtin_bit_cast(float, h_ << 16), __builtin_bit_cast(float, h_ & 0xffff0000u)};                   \
  } while (0)
#define U2(p) PK_SUB(rp[p], rp[p], fu[p])
#define U3(p)                                                                                                          \
  do {                                                                                                                 \
    const uint32_t h_ = __builtin_bit_cast(uint32_t, __builtin_convertvector(rp[p], bf16x2));                          \
    t1[p] = h_;                                                                                                        \
    fu[p] = f32x2{__builtin_bit_cast(float, h_ << 16), __builtin_bit_cast(float, h_ & 0xffff0000u)};                   \
  } while (0)
#define U4(p) do { f32x2 w_; PK_SUB(w_, rp[p], fu[p]); t2[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(w_, bf16x2)); } while (0)
#endif
// STA(st, i, tp, pl): the 8 bytes of plane pl (tp = t0 / t1 / t2) of item i -> the stage.  STB(st, q): weight image piece q.
#if EMU_ABL == 2 || EMU_ABL == 3
#define STA(st, i, tp, pl) asm volatile("" :: "v"(tp[2 * (i)]), "v"(tp[2 * (i) + 1]))
#define STB(st, q) asm volatile("" :: "v"(rb[q]))
#elif EMU_ABL == 7
#define STA(st, i, tp, pl) reinterpret_cast<u32x2*>(st)[wslot + (i) * 128 + (pl) * 4 * TM] = u32x2{tp[2 * (i)], tp[2 * (i) + 1]}
#define STB(st, q) asm volatile("" :: "v"(rb[q]))
#else
#define STA(st, i, tp, pl) reinterpret_cast<u32x2*>(st)[wslot + (i) * 128 + (pl) * 4 * TM] = u32x2{tp[2 * (i)], tp[2 * (i) + 1]}
#define STB(st, q) (st)[A_U4 + tid + (q) * NT] = rb[q]
#endif
#define LDA(st, p, i) __builtin_bit_cast(bf16x8, (st)[aread + ((p) * 2 + kh) * TM + (i) * 32])
#define LDB(st, p, j) __builtin_bit_cast(bf16x8, (st)[A_U4 + wn * WN + l31 + ((p) * 2 + kh) * TN + (j) * 32])
#define SB() __builtin_amdgcn_sched_barrier(0)
// one MFMA + the unit that hides under it
#define M1(ax, bx, i, j, work) do { acc[i][j] = MFB(ax[i], bx[j], acc[i][j]); work; SB(); } while (0)
#define MM(ax, bx) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFB(ax[i], bx[j], acc[i][j])
// phase s: on entry aX = x0 fragments of slab s - 1, b2 / b1 / bC = its weight fragments (planes 2, 1, 0), rp / rb / rm = slab
// s + 1 as loaded; cur = the stage that holds slab s, nxt = the stage that receives slab s + 1.  On exit aY, b2, b1, bN = slab s.
// The slot table (which unit hides under which MFMA) is generated: tools/gen/kc2_phase.py -> kc2_phase.inc.
#include "kc2_phase.inc"
#define SYNC() do { SB(); __syncthreads(); SB(); } while (0)

  // the slab count is rounded up to an even number (a pad slab stages zeros for A: its products add exact zeros), so the phases
  // after the head come in pairs plus one and the two register assignments never have to merge
  const int nslab2 = (nslab + 1) & ~1;
  bf16x8 aP[4], aQ[4], bP[NJ], bQ[NJ], b1[NJ], b2[NJ];
  // prologue (left to the compiler): slab 0 -> st0, slab 1 -> registers; the first half of slab 0; slab 1 -> st1, slab 2 -> registers
#define STAGE_ALL(st, sl)                                                                                              \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
      UI(i, sl);                                                                                                       \
      U1(2 * i); U2(2 * i); U3(2 * i); U4(2 * i); U1(2 * i + 1); U2(2 * i + 1); U3(2 * i + 1); U4(2 * i + 1);          \
      STA(st, i, t0, 0); STA(st, i, t1, 1); STA(st, i, t2, 2);                                                         \
    }                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) STB(st, q);                                                         \
  } while (0)
#define LOAD_ALL(sl)                                                                                                   \
  do {                                                                                                                 \
    /* issue order pinned to a phase's: the vmcnt waits inside the loop are counted against BOTH histories that reach   */ \
    /* the loop head (left free, hipcc put item 0 second to last here and the odd phases waited vmcnt(1) at slot 2)    */ \
    SB(); LDGA(0, sl); SB();                                                                                           \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) { LDGB(q, sl); SB(); }                                              \
    _Pragma("unroll") for (int i = 1; i < 4; ++i) { LDGA(i, sl); SB(); }                                               \
  } while (0)
  LOAD_ALL(0);
  STAGE_ALL(st0, 0);
  LOAD_ALL(1);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) { aQ[i] = LDA(st0, 2, i); aP[i] = LDA(st0, 1, i); }
#pragma unroll
  for (int j = 0; j < NJ; ++j) { bP[j] = LDB(st0, 0, j); b1[j] = LDB(st0, 1, j); b2[j] = LDB(st0, 2, j); }
  MM(aQ, bP);                                                  // x2 y0 of slab 0
#pragma unroll
  for (int i = 0; i < 4; ++i) aQ[i] = LDA(st0, 0, i);
  MM(aP, b1); MM(aP, bP);                                      // x1 y1, x1 y0
  STAGE_ALL(st1, 1);
  LOAD_ALL(2);
  SYNC();                                                      // aQ = x0 fragments of slab 0, bP / b1 / b2 its weight fragments
  for (int s = 1; s + 1 < nslab2; s += 2) {
    PHASE(st1, st0, s, aQ, aP, bP, bQ);
    SYNC();
    PHASE(st0, st1, s + 1, aP, aQ, bQ, bP);
    SYNC();
  }
  PHASE(st1, st0, nslab2 - 1, aQ, aP, bP, bQ);
  SB();
  MM(aP, b2); MM(aP, b1); MM(aP, bQ);
  __syncthreads();
#undef LDGA
#undef LDGB
#undef UI
#undef U1
#undef U2
#undef U3
#undef U4
#undef STA
#undef STB
#undef LDA
#undef LDB
#undef SB
#undef M1
#undef MM
#undef NOP_
#undef PHASE
#undef SYNC
#undef STAGE_ALL
#undef LOAD_ALL
  emu_epilogue<TM, TN, NJ>(g, acc, st0, m0, n0, wm, wn, wave, lane, l31, kh, MASK ? g.ascale : 1.f);
}


// ============================================================================================================================
// f16x2 form ("h2"): the same contractions from TWO f16 pieces per operand and THREE products.
//   x s = hi + lo + r,  hi = f16(x s), lo = f16(x s - hi), |r| <= max(2^-22 |x s|, 2^-25) (on average 2^-24 |x s|): with the operand
//   scaled by a power of two s so that max |x s| lies in [2^13, 2^14) every element within 2^-16 of the largest keeps 22 bits; below
//   that lo is an f16 subnormal: an ABSOLUTE error of 2^-38 max |x|.  x y is accumulated in f32 from lo hi + hi lo + hi hi (each f16 x f16 product
//   is exact in f32); the dropped lo lo term is <= 2^-22 |x y|, on average 2^-26 |x y| with a random sign - the rounding of an f32
//   multiply-add.  Half the MFMA work of the bf16x3 form at the same matrix-pipe rate; the price is the scale: the largest
//   magnitude of every operand has to be known when its contraction is launched (weights: found while the image is built;
//   activations / gradients: magnitude words written by the producing kernel's epilogue, or by hoisdf's own magnitude pass).
// Tile 256 x 256 x 16, 4 waves as 2 x 2, wave tile 128 x 128 = 4 x 4 MFMA blocks (256 accumulators in AGPRs, one workgroup per CU):
// with half the MFMA work per byte the 256 x 128 tile of emu_kc2_kernel would ask the vector memory pipe for ~100 GB/s per CU.
// Staging, LDS layout and the rotated, pinned phase are emu_kc2_kernel's (tools/gen/h2_phase.py -> h2_phase.inc).
// Weight image: column tile tn (256 columns), slab s (16 k), plane p (hi, lo), chunk c (8 k), row r: 16 bytes at
// ((((tn * nslab + s) * 2 + p) * 2 + c) * 256 + r) * 16; behind the last tile a 128-byte trailer: 16 magnitude words, then {s, 1 / s}.
// ============================================================================================================================
namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define MFH(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
constexpr int HTM = 256, HTN = 256;
constexpr int HA_U4 = 2 * 2 * HTM, HB_U4 = 2 * 2 * HTN;      // 16-byte units of the activation stage / of one 256-row image block per slab
constexpr int H_TRAILER = 128;

// (the power-of-two operand scale from the magnitude words and the 256-thread block maximum live in common.h: the attention kernels use them too)
__device__ __forceinline__ uint32_t h2_exp(uint32_t amax_bits) { return mag_exp(amax_bits); }
__device__ __forceinline__ float h2_scale(uint32_t amax_bits) { return mag_scale(amax_bits); }
__device__ __forceinline__ float h2_inv_scale(uint32_t amax_bits) { return mag_inv_scale(amax_bits); }
}  // namespace

// row magnitudes (common.h) of a row-major f32 matrix measured by the library: one wave per row, eight rows in flight per wave, plain
// stores (every row is written: no zeroing needed).  K % 4 == 0, rows 16-byte aligned.
__global__ __launch_bounds__(256) void emu_rowmag_kernel(const float* __restrict__ x, long ld, long M, int K, uint32_t* __restrict__ words) {
  constexpr int RPW = 8;
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  const int k4 = K >> 2;
  uint32_t m[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    m[r] = 0u;
    if (row0 + r < M)
      for (int c = lane; c < k4; c += 64) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (row0 + r) * ld + 4 * c);
        m[r] = max(m[r], max(max(v[0] & 0x7fffffffu, v[1] & 0x7fffffffu), max(v[2] & 0x7fffffffu, v[3] & 0x7fffffffu)));
      }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const uint32_t mm = group_max_u32<64>(m[r]);
    if (lane == 0 && row0 + r < M) words[row0 + r] = mm;
  }
}
// head magnitudes (common.h) of a row-major f32 matrix: groups of 64 columns x samples of L rows -> words[group * nb + sample] (zero on
// entry).  One wave per 16 consecutive rows; a 16-lane group owns a column group (ncols = 64 groups: (groups + 3) / 4 sweeps).
__global__ __launch_bounds__(256) void emu_headmag_kernel(const float* __restrict__ x, long ld, long M, int groups, int L, int nb,
                                                          uint32_t* __restrict__ words) {
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  for (int g0 = 0; g0 < groups; g0 += 4) {
    const int grp = g0 + (lane >> 4);
    uint32_t m = 0u; long cur = -1;
    for (int r = 0; r < 16; ++r) {
      const long row = row0 + r;
      if (row >= M) break;
      const long b = row / L;
      if (b != cur) {
        if (cur >= 0 && grp < groups) { const uint32_t mm = group_max_u32<16>(m); if ((lane & 15) == 0) atomicMax(words + (size_t)grp * nb + cur, mm); }
        cur = b; m = 0u;
      }
      if (grp < groups) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + row * ld + grp * 64 + 4 * (lane & 15));
        m = max(m, max(max(v[0] & 0x7fffffffu, v[1] & 0x7fffffffu), max(v[2] & 0x7fffffffu, v[3] & 0x7fffffffu)));
      }
    }
    if (cur >= 0 && grp < groups) { const uint32_t mm = group_max_u32<16>(m); if ((lane & 15) == 0) atomicMax(words + (size_t)grp * nb + cur, mm); }
  }
}

// ---- weight -> f16x2 image.  Pass 1: 16 magnitude words per weight into the trailer; pass 2: scale, split, write (+ {s, 1 / s}).
__host__ __device__ __forceinline__ uint32_t* h2_trailer(void* image, int R, int Kc) {
  return reinterpret_cast<uint32_t*>(static_cast<char*>(image) + (size_t)((R + HTN - 1) / HTN) * ((Kc + KS - 1) / KS) * HB_U4 * 16);
}
__device__ __forceinline__ void h2_weight_amax_unit(const float* __restrict__ W, int ldw, int N, int K, int part, uint32_t* trailer, uint32_t* red4) {
  uint32_t m = 0u;
  const long n = (long)N * K;
  for (long i = (long)part * 256 + threadIdx.x; i < n; i += 16 * 256) {
    const long r = i / K;
    m = max(m, __builtin_bit_cast(uint32_t, W[r * ldw + (i - r * K)]) & 0x7fffffffu);
  }
  m = block_max_u32(m, red4);
  if (threadIdx.x == 0) trailer[part] = m;
}
__device__ __forceinline__ void h2_prep_weight_unit(const float* __restrict__ W, int ldw, int R, int Kc, int transpose, int nslab,
                                                    long idx, u32x4* __restrict__ img) {
  uint32_t* tr = h2_trailer(img, R, Kc);
  uint32_t am = 0u;
#pragma unroll
  for (int i = 0; i < 16; ++i) am = max(am, tr[i]);
  const float sc = h2_scale(am);
  if (idx == 0) {                                  // (the whole trailer is defined: images compare equal byte for byte)
    reinterpret_cast<float*>(tr)[16] = sc; reinterpret_cast<float*>(tr)[17] = h2_inv_scale(am);
#pragma unroll
    for (int i = 18; i < H_TRAILER / 4; ++i) tr[i] = 0u;
  }
  const int r = (int)(idx % HTN);
  const int c = (int)((idx / HTN) % 2);
  const int s = (int)((idx / (2 * HTN)) % nslab);
  const int tn = (int)(idx / ((long)2 * HTN * nslab));
  const int row = tn * HTN + r;
  const int k0 = s * KS + c * 8;
  f16x8 hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k0 + i;
    float v = 0.f;
    if (row < R && k < Kc) v = transpose ? W[(size_t)k * ldw + row] : W[(size_t)row * ldw + k];
    v *= sc;
    hi[i] = (_Float16)v;
    lo[i] = (_Float16)(v - (float)hi[i]);
  }
  const size_t base = ((size_t)(tn * nslab + s) * 2) * 2 * HTN;
  img[base + (0 * 2 + c) * HTN + r] = __builtin_bit_cast(u32x4, hi);
  img[base + (1 * 2 + c) * HTN + r] = __builtin_bit_cast(u32x4, lo);
}
__global__ __launch_bounds__(256) void h2_weight_amax_kernel(const float* __restrict__ W, int ldw, int N, int K, int R, int Kc, void* image) {
  __shared__ uint32_t red4[4];
  h2_weight_amax_unit(W, ldw, N, K, blockIdx.x, h2_trailer(image, R, Kc), red4);
}
__global__ __launch_bounds__(256) void h2_weight_amax_batch_kernel(const hoisdf_emu_prep_item* __restrict__ items) {
  __shared__ uint32_t red4[4];
  const hoisdf_emu_prep_item it = items[blockIdx.x >> 4];
  const int R = it.transpose ? it.K : it.N, Kc = it.transpose ? it.N : it.K;
  h2_weight_amax_unit(it.W, it.ldw, it.N, it.K, blockIdx.x & 15, h2_trailer(it.image, R, Kc), red4);
}
__global__ __launch_bounds__(256) void h2_prep_weight_kernel(const float* __restrict__ W, int ldw, int R, int Kc, int transpose, int nslab,
                                                             long total, u32x4* __restrict__ img) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx < total) h2_prep_weight_unit(W, ldw, R, Kc, transpose, nslab, idx, img);
}
__global__ __launch_bounds__(256) void h2_prep_weight_batch_kernel(const hoisdf_emu_prep_item* __restrict__ items, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_block <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const hoisdf_emu_prep_item it = items[lo];
  const int R = it.transpose ? it.K : it.N, Kc = it.transpose ? it.N : it.K;
  const int nslab = ((Kc + KS - 1) / KS);
  const long total = (long)((R + HTN - 1) / HTN) * nslab * 2 * HTN;
  const long idx = ((long)blockIdx.x - it.first_block) * 256 + threadIdx.x;
  if (idx < total) h2_prep_weight_unit(it.W, it.ldw, R, Kc, it.transpose, nslab, idx, static_cast<u32x4*>(it.image));
}

// NJ = 4: tile 256 x 256 (one workgroup per CU); NJ = 2: tile 256 x 128, wave tile 128 x 64, two workgroups per CU (few or narrow
// tiles: the prologue / epilogue of one workgroup under the main loop of the other) - it reads one half of the image's 256-row blocks
template <bool MASK, bool KTAIL, int NJ>
__global__ __launch_bounds__(NT, NJ == 4 ? 1 : 2) void emu_h2_kernel(EmuArgs g) {
  constexpr int TN_ = 64 * NJ;                                        // tile width
  constexpr int BST = 2 * 2 * TN_;                                    // 16-byte units of the weight operand per stage
  constexpr int STG = HA_U4 + BST;
  constexpr int EPI = (4 * 32 * (TN_ / 2 + 4) * 4 + 64) / 16;         // the epilogue's four transposition slices + a few words
  __shared__ __attribute__((aligned(16))) u32x4 st0[EPI > STG ? EPI : STG];
  __shared__ __attribute__((aligned(16))) u32x4 st1[STG];
  __shared__ __attribute__((aligned(16))) float rpost[HTM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int t = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int m0 = tm * HTM, n0 = tn * TN_;
  const int nslab = (g.K + KS - 1) / KS;
  const int last = nslab - 1;
  const int rl = lane >> 2, qd = lane & 3, cq = qd >> 1;

  // operand scales: ONE PER ROW of A from its row magnitudes (common.h; a row's rounding depends on that row alone), the weight's from
  // the image trailer.  The staging thread keeps the scales of its four rows; every output row's factor (1 / row scale, 1 / weight
  // scale, 1 / keep) waits in LDS for the epilogue.  f16 conversions saturate (a word below the row's true maximum clips, no Inf).
  // (the five words are REQUESTED here, ahead of the first slab's loads, and used behind them: no load round trip of its own per tile)
  f16_saturate_on();
  const uint32_t wpost = m0 + tid < g.M ? g.a_amax[m0 + tid] : 0u;
  uint32_t wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + i * 64 + wave * 16 + rl;
    wrow[i] = row < g.M ? g.a_amax[row] : 0u;
  }
  float sAr[4];

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging roles, descriptors and LDS slots: emu_kc2_kernel's (item i = row i * 64 + wave * 16 + lane / 4, quad lane % 4)
  const int rows_in = min(HTM, g.M - m0);
  __amdgpu_buffer_rsrc_t rsa[4], rsm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rows_q = max(min(rows_in - i * 64, 64), 0);
#ifdef H2_ABL_NOLOADA                 /* (tools/ablate_h2.sh: every tile reads the FIRST tile's rows - the same loads, served by the caches) */
    rsa[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + ((size_t)i * 64) * g.lda), 0,
                                               rows_q > 0 ? (int)((((long)rows_q - 1) * g.lda + g.K) * 4) : 0, 0x00020000);
#else
    rsa[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + ((size_t)m0 + i * 64) * g.lda), 0,
                                               rows_q > 0 ? (int)((((long)rows_q - 1) * g.lda + g.K) * 4) : 0, 0x00020000);
#endif
    rsm[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(MASK ? g.abits + ((size_t)m0 + i * 64) * g.ldbits : nullptr), 0,
                                               MASK ? (int)((long)rows_q * g.ldbits * 4) : 0, 0x00020000);
  }
  // the image is laid out in 256-row blocks (HB_U4 units per slab): a 128-wide tile reads rows (tn & 1) * 128 .. + 127 of its block
  const int tb = NJ == 4 ? tn : tn >> 1, r0 = NJ == 4 ? 0 : (tn & 1) * 128;
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(g.Bimg + (size_t)tb * nslab * HB_U4), 0, nslab * HB_U4 * 16, 0x00020000);
  const int boff = NJ == 4 ? tid * 16 : ((tid >> 7) * HTN + r0 + (tid & 127)) * 16;     // piece q: + q * (NJ == 4 ? NT : 2 * HTN) units
  const int aoff = (int)((((long)wave * 16 + rl) * g.lda + 4 * qd) * 4);
  const int moff = (int)(((long)wave * 16 + rl) * g.ldbits * 4);
  const int wslot = 2 * (cq * HTM + ((wave * 16 + rl) ^ (cq << 2))) + (qd & 1);
  const int aread = (wm * 128 + l31) ^ (kh << 2);
  const int kq = g.K - 4 * qd;
  f32x2 rp[8], fu[8];
  uint32_t t0[8], t1[8];
  u32x4 rb[NJ];
  uint32_t rm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, mb = 0xfu;
  bool kin = true;
#define NOP_ ((void)0)
#define HLDGA(i, sl)                                                                                                   \
  do {                                                                                                                 \
    const int k0_ = min((sl), last) * KS;                      /* (uniform) a pad slab re-reads the last one */        \
    const f32x4 v_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa[i], aoff, k0_ * 4, 0));       \
    rp[2 * (i)] = f32x2{v_[0], v_[1]}; rp[2 * (i) + 1] = f32x2{v_[2], v_[3]};                                          \
    if (MASK) rm[i] = __builtin_amdgcn_raw_buffer_load_b32(rsm[i], moff, (k0_ >> 5) * 4, 0);                           \
  } while (0)
#define HLDGB(q, sl) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rsb, boff + (q) * ((NJ == 4 ? NT : 2 * HTN) * 16), min((sl), last) * (HB_U4 * 16), 0)
#define HUI(i, sl) do { if (MASK) mb = rm[i] >> ((((sl) & 1) << 4) + 4 * qd); if (KTAIL) kin = (sl) * KS < kq; } while (0)
#define PK_SUB(d, a, b) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b))
// pair p = 2 i + j of item i: HC1 sign bitmap / k tail / scale, hi plane (v_cvt_pk_f16_f32, round to nearest) and its f32 image;
// HC2 residual, lo plane
#define HC1(p)                                                                                                         \
  do {                                                                                                                 \
    f32x2 v_ = rp[p];                                                                                                  \
    if (MASK) {           /* (asm: see emu_kc2_kernel's U1) */                                                         \
      float xa_, xb_;                                                                                                  \
      asm("v_and_b32 %0, %1, %2" : "=v"(xa_) : "v"(__builtin_amdgcn_sbfe((int)mb, 2 * ((p) & 1), 1)), "v"(v_.x));       \
      asm("v_and_b32 %0, %1, %2" : "=v"(xb_) : "v"(__builtin_amdgcn_sbfe((int)mb, 2 * ((p) & 1) + 1, 1)), "v"(v_.y));   \
      v_ = f32x2{xa_, xb_};                                                                                            \
    }                                                                                                                  \
    if (KTAIL) { if (!kin) v_ = f32x2{0.f, 0.f}; }                                                                     \
    v_ *= sAr[(p) >> 1];                                                                                               \
    const f16x2 h_ = __builtin_convertvector(v_, f16x2);                                                               \
    t0[p] = __builtin_bit_cast(uint32_t, h_); rp[p] = v_;                                                              \
    fu[p] = __builtin_convertvector(h_, f32x2);                                                                        \
  } while (0)
#define HC2(p) do { f32x2 w_; PK_SUB(w_, rp[p], fu[p]); t1[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(w_, f16x2)); } while (0)
#define HSTA(st, i, tp, pl) reinterpret_cast<u32x2*>(st)[wslot + (i) * 128 + (pl) * 4 * HTM] = u32x2{tp[2 * (i)], tp[2 * (i) + 1]}
#define HSTB(st, q) (st)[HA_U4 + tid + (q) * NT] = rb[q]
#define HLA(st, p, i) __builtin_bit_cast(f16x8, (st)[aread + ((p) * 2 + kh) * HTM + (i) * 32])
#define HLB(st, p, j) __builtin_bit_cast(f16x8, (st)[HA_U4 + wn * (TN_ / 2) + l31 + ((p) * 2 + kh) * TN_ + (j) * 32])
#define SB() __builtin_amdgcn_sched_barrier(0)
#define HM1(ax, bx, i, j, work) do { acc[i][j] = MFH(ax[i], bx[j], acc[i][j]); work; SB(); } while (0)
#define HMM(ax, bx) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFH(ax[i], bx[j], acc[i][j])
#include "h2_phase.inc"
#define SYNC() do { SB(); __syncthreads(); SB(); } while (0)
#define HSTAGE_ALL(st, sl)                                                                                             \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
      HUI(i, sl);                                                                                                      \
      HC1(2 * i); HC2(2 * i); HC1(2 * i + 1); HC2(2 * i + 1);                                                          \
      HSTA(st, i, t0, 0); HSTA(st, i, t1, 1);                                                                          \
    }                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < NJ; ++q) HSTB(st, q);                                                        \
  } while (0)
#define HLOAD_ALL(sl)                                                                                                  \
  do {                                                                                                                 \
    /* issue order pinned to a phase's (B pieces, then the items): the vmcnt waits inside the loop count BOTH histories */ \
    SB();                                                                                                              \
    _Pragma("unroll") for (int q = 0; q < NJ; ++q) { HLDGB(q, sl); SB(); }                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { HLDGA(i, sl); SB(); }                                              \
  } while (0)

  const int nslab2 = (nslab + 1) & ~1;
  f16x8 aH[4], aL[4], bP[NJ], bQ[NJ], bL[NJ];
  HLOAD_ALL(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) sAr[i] = h2_scale(wrow[i]);
  HSTAGE_ALL(st0, 0);
  HLOAD_ALL(1);
  rpost[tid] = h2_inv_scale(wpost) * g.b_scale[1] * (MASK ? g.ascale : 1.f);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) { aL[i] = HLA(st0, 1, i); aH[i] = HLA(st0, 0, i); }
#pragma unroll
  for (int j = 0; j < NJ; ++j) { bP[j] = HLB(st0, 0, j); bL[j] = HLB(st0, 1, j); }
  HMM(aL, bP); HMM(aH, bL);                                    // lo hi, hi lo of slab 0
  HSTAGE_ALL(st1, 1);
  HLOAD_ALL(2);
  SYNC();                                                      // aH / bP = hi fragments of slab 0
#define HPH(cur, nxt, s, bC, bN) do { if constexpr (NJ == 4) { HPHASE(cur, nxt, s, bC, bN); } else { HPHASE2(cur, nxt, s, bC, bN); } } while (0)
  for (int s = 1; s + 1 < nslab2; s += 2) {
    HPH(st1, st0, s, bP, bQ);
    SYNC();
    HPH(st0, st1, s + 1, bQ, bP);
    SYNC();
  }
  HPH(st1, st0, nslab2 - 1, bP, bQ);
  SB();
  HMM(aH, bQ);                                                 // hi hi of the last slab
  __syncthreads();
#undef HLDGA
#undef HLDGB
#undef HUI
#undef PK_SUB
#undef HC1
#undef HC2
#undef HSTA
#undef HSTB
#undef HLA
#undef HLB
#undef SB
#undef HM1
#undef HMM
#undef NOP_
#undef HPHASE
#undef HPHASE2
#undef HPH
#undef SYNC
#undef HSTAGE_ALL
#undef HLOAD_ALL
  emu_epilogue<HTM, TN_, NJ>(g, acc, st0, m0, n0, wm, wn, wave, lane, l31, kh, 1.f, rpost);
}


// ============================================================================================================================
// grad-weight: dW[n][k] = sum_m dy_eff[m][n] x[m][k], db[n] = sum_m dy_eff[m][n].  The contraction runs over the ROWS of both
// operands, so each needs its planes transposed ([column][8 consecutive m]): a staging thread loads a 4-column x 8-row patch
// (8 float4, lanes along the columns: 1 KB contiguous per row), transposes it in registers and writes, per column, the three
// 16-byte pieces of that column's chunk - no transposed copy of an activation ever goes through HBM.
// Tile 256 (n) x 256 (k): one 4 x 8 patch per thread covers both operands of a 16-row slab (threads 0-127: dy, 128-255: x);
// 4 waves as 2 x 2, wave tile 128 x 128 = 4 x 4 MFMA blocks, 256 accumulators (AGPRs), one workgroup per CU; the rows are
// split over the workgroups (every slice of a tile on one XCD) into partial tiles + an ordered reduce: no atomics.
// ============================================================================================================================
namespace {
constexpr int DT = 256;                                  // tile edge (both n and k)

// a value the compiler cannot prove wave-uniform (it depends on tid < 128, which is uniform per wave) into scalar registers
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

struct DwArgs {
  const float* dy; long lddy;
  const float* x; long ldx;
  const uint32_t* bits; int ldbits; float ascale;
  float* C; long c_split_stride;             // partial tiles [split][N][K] (or dW itself when splitk == 1)
  float* colsum; long colsum_split_stride;   // partial bias gradients [split][N] (or db), may be null
  int M, N, K;
  int splitk, m_per_split, tiles_n, tiles_k;
  const uint32_t* dy_amax; const uint32_t* x_amax;      // f16x2 form: row magnitudes (common.h) of dy and x, M words each
};
}  // namespace

// DTK = tile width along k: 256 (wave tile 128 x 128, 256 accumulators, one workgroup per CU) or 128 (wave tile 128 x 64, two
// workgroups per CU: the conversion phase of one overlaps the MFMAs of the other; x patches on wave 2 only, wave 3 stages nothing)
template <bool MASK, int DTK>
__global__ __launch_bounds__(NT, DTK == 256 ? 1 : 2) void emu_dw_kernel(DwArgs g) {
  constexpr int NJ = DTK / 64;                         // 32-column blocks per wave along k
  constexpr int WK = DTK / 2;                          // wave tile width along k
  constexpr int A_U4 = 3 * 2 * DT, STAGE = A_U4 + 3 * 2 * DTK;
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int ntile = g.tiles_n * g.tiles_k;
  const int bid = blockIdx.x;
  const int split = (bid & 7) + 8 * (bid / (8 * ntile));      // every slice of one tile on the same XCD (shared L2)
  const int t = (bid >> 3) % ntile;
  if (split >= g.splitk) return;
  const int tn = t / g.tiles_k, tk = t - tn * g.tiles_k;
  const int n0 = tn * DT, k0 = tk * DTK;
  const int mbeg = split * g.m_per_split;
  const int mend = min(g.M, mbeg + g.m_per_split);
  const int nslab = (mend - mbeg + KS - 1) / KS;
  const int last = nslab - 1;

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging patch of this thread: operand (wave-uniform), column group cg (4 columns), chunk c (8 rows of the slab)
  const bool isA = tid < 128;
  const bool stager = DTK == 256 || tid < 192;          // (wave-uniform)
  const int cg = (isA || DTK == 256) ? (tid & 63) : (tid & 31), c = (isA || DTK == 256) ? ((tid >> 6) & 1) : ((tid >> 5) & 1);
  const int col0 = (isA ? n0 : k0) + 4 * cg;
  const int ncol = isA ? g.N : g.K;                      // multiples of 4 (checked by the host): a patch column group is all in or out
  const bool col_ok = stager && col0 < ncol;
  // addresses: a wave-uniform row base (scalar registers: operand pointer + slab row * leading dimension + e rows) plus ONE
  // per-thread byte offset that never changes (chunk rows + column group) - no vector address arithmetic in the slab loop
  const long ld = uni64(isA ? g.lddy : g.ldx);
  // (x addressed relative to dy: pointer arithmetic on a kernel argument keeps the global address space, an integer round trip
  // would turn the loads into flat ones)
  const char* opbase = reinterpret_cast<const char*>(g.dy) +
                       (long)uni64(isA ? 0ul : (uint64_t)(reinterpret_cast<const char*>(g.x) - reinterpret_cast<const char*>(g.dy)));
  const uint32_t voff = (uint32_t)(((long)c * 8 * ld + (col_ok ? col0 : 0)) * 4);
  const char* bitbase = reinterpret_cast<const char*>(g.bits);
  const uint32_t boff = (uint32_t)(((long)c * 8 * g.ldbits + ((col_ok ? col0 : 0) >> 5)) * 4);
  const float* src = (isA ? g.dy : g.x) + (col_ok ? col0 : 0);
  const uint32_t* bsrc = (MASK && isA) ? g.bits + ((col_ok ? col0 : 0) >> 5) : nullptr;
  const int bsh = col0 & 31;
  const bool do_colsum = isA && g.colsum != nullptr && tk == 0;
  float4 rv[8];
  uint32_t rm[8];
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#define DW_LOAD(sl)                                                                                                   \
  do {                                                                                                                \
    const int ms_ = mbeg + (sl) * KS;                        /* (uniform) first row of the slab */                    \
    if (ms_ + KS <= g.M) {                                                                                            \
      const char* sb_ = opbase + (size_t)ms_ * ld * 4;                                                                \
      const char* mb2_ = bitbase + (size_t)ms_ * g.ldbits * 4;                                                        \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                 \
        rv[e] = *reinterpret_cast<const float4*>(sb_ + (size_t)e * ld * 4 + voff);                                    \
        if (MASK) rm[e] = isA ? *reinterpret_cast<const uint32_t*>(mb2_ + (size_t)e * g.ldbits * 4 + boff) : 0xffffffffu; \
      }                                                                                                               \
    } else {                                                 /* the slab that crosses the end of the operands */      \
      const int mb_ = ms_ + c * 8;                                                                                    \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                 \
        const int m_ = min(mb_ + e, g.M - 1);                                                                         \
        rv[e] = *reinterpret_cast<const float4*>(src + (size_t)m_ * ld);                                              \
        if (MASK) rm[e] = isA ? bsrc[(size_t)m_ * g.ldbits] : 0xffffffffu;                                            \
      }                                                                                                               \
    }                                                                                                                 \
  } while (0)
#define DW_STORE(st, sl)                                                                                              \
  do {                                                                                                                \
    const int mb_ = mbeg + (sl) * KS + c * 8;                                                                         \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                   \
      float4 v_ = rv[e];                                                                                              \
      if (MASK && isA) {                                                                                              \
        const uint32_t nib_ = rm[e] >> bsh;                                                                           \
        v_.x = (nib_ & 1u) ? v_.x * g.ascale : 0.f;                                                                   \
        v_.y = (nib_ & 2u) ? v_.y * g.ascale : 0.f;                                                                   \
        v_.z = (nib_ & 4u) ? v_.z * g.ascale : 0.f;                                                                   \
        v_.w = (nib_ & 8u) ? v_.w * g.ascale : 0.f;                                                                   \
      }                                                                                                               \
      rv[e] = v_;                                                                                                     \
    }                                                                                                                 \
    if (!col_ok || mbeg + (sl) * KS + KS > mend) {           /* (rare) rows past the slice, columns past the operand */ \
      _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                                   \
        if (mb_ + e >= mend || !col_ok) rv[e] = make_float4(0.f, 0.f, 0.f, 0.f);                                      \
    }                                                                                                                 \
    if (do_colsum) {                                                                                                  \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                 \
        csum.x += rv[e].x; csum.y += rv[e].y; csum.z += rv[e].z; csum.w += rv[e].w;                                   \
      }                                                                                                               \
    }                                                                                                                 \
    const int rs_ = isA ? DT : DTK;                          /* rows per (plane, chunk) region of this operand */      \
    u32x4* dst_ = (st) + (isA ? 0 : A_U4) + c * rs_ + 4 * cg;                                                         \
    bf16x8 p0, p1, p2;                                                                                                \
    split3x8(make_float4(rv[0].x, rv[1].x, rv[2].x, rv[3].x), make_float4(rv[4].x, rv[5].x, rv[6].x, rv[7].x), p0, p1, p2); \
    dst_[0] = __builtin_bit_cast(u32x4, p0); dst_[2 * rs_] = __builtin_bit_cast(u32x4, p1); dst_[4 * rs_] = __builtin_bit_cast(u32x4, p2); \
    split3x8(make_float4(rv[0].y, rv[1].y, rv[2].y, rv[3].y), make_float4(rv[4].y, rv[5].y, rv[6].y, rv[7].y), p0, p1, p2); \
    dst_[1] = __builtin_bit_cast(u32x4, p0); dst_[2 * rs_ + 1] = __builtin_bit_cast(u32x4, p1); dst_[4 * rs_ + 1] = __builtin_bit_cast(u32x4, p2); \
    split3x8(make_float4(rv[0].z, rv[1].z, rv[2].z, rv[3].z), make_float4(rv[4].z, rv[5].z, rv[6].z, rv[7].z), p0, p1, p2); \
    dst_[2] = __builtin_bit_cast(u32x4, p0); dst_[2 * rs_ + 2] = __builtin_bit_cast(u32x4, p1); dst_[4 * rs_ + 2] = __builtin_bit_cast(u32x4, p2); \
    split3x8(make_float4(rv[0].w, rv[1].w, rv[2].w, rv[3].w), make_float4(rv[4].w, rv[5].w, rv[6].w, rv[7].w), p0, p1, p2); \
    dst_[3] = __builtin_bit_cast(u32x4, p0); dst_[2 * rs_ + 3] = __builtin_bit_cast(u32x4, p1); dst_[4 * rs_ + 3] = __builtin_bit_cast(u32x4, p2); \
  } while (0)

  if (nslab > 0 && stager) {
    DW_LOAD(0);
    DW_STORE(lds, 0);
    DW_LOAD(min(1, last));
  }
  __syncthreads();

  for (int s = 0; s < nslab; ++s) {
    const u32x4* st = lds + (s & 1) * STAGE;
    u32x4* nx = lds + ((s + 1) & 1) * STAGE;
    const u32x4* sa = st + wm * 128 + l31;
    const u32x4* sb = st + A_U4 + wn * WK + l31;
    bf16x8 b0[NJ], b1[NJ], b2[NJ], a[4];
#define RD_B(dst, p) _Pragma("unroll") for (int j = 0; j < NJ; ++j) dst[j] = __builtin_bit_cast(bf16x8, sb[((p) * 2 + kh) * DTK + j * 32])
#define RD_A(p) _Pragma("unroll") for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, sa[((p) * 2 + kh) * DT + i * 32])
#define MM1(bx) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFB(a[i], bx[j], acc[i][j])
    // long phase first: 48 MFMAs queue up right behind the barrier, the conversion of the next slab follows them.  (Measured
    // on MI355X, tools/mb_emu.py: pinning only the global loads and letting hipcc spread the conversion over the MFMAs, or an
    // explicit sched_group_barrier pipeline of 1 MFMA + 6 VALU, are within 2 % of this form.)
    RD_B(b0, 0); RD_A(0); RD_B(b1, 1); RD_B(b2, 2);
    MM1(b2); MM1(b1); MM1(b0);                 // x0 y2, x0 y1, x0 y0
    __builtin_amdgcn_sched_barrier(0);
    RD_A(1);
    if (stager) {
      if (s + 1 < nslab) DW_STORE(nx, s + 1);
      DW_LOAD(min(s + 2, last));
    }
    __builtin_amdgcn_sched_barrier(0);
    MM1(b1); MM1(b0);                          // x1 y1, x1 y0
    RD_A(2);
    MM1(b0);                                   // x2 y0
    __syncthreads();
  }
#undef DW_LOAD
#undef DW_STORE
#undef RD_A
#undef RD_B
#undef MM1

  // bias gradient partial: the two chunk threads of a column group add up through LDS (all waves are past the last barrier)
  if (g.colsum != nullptr && tk == 0) {
    float* red = reinterpret_cast<float*>(lds);
    if (isA) *reinterpret_cast<float4*>(&red[c * DT + 4 * cg]) = csum;
    __syncthreads();
    if (tid < DT) {
      const int n = n0 + tid;
      if (n < g.N) g.colsum[(size_t)split * g.colsum_split_stride + n] = red[tid] + red[DT + tid];
    }
    __syncthreads();
  }

  // epilogue: one row of 32 x 32 blocks (32 x WK) at a time through the wave's private LDS slice
  float* Cb = g.C + (size_t)split * g.c_split_stride;
  const bool full = (n0 + DT <= g.N) && (k0 + DTK <= g.K) && (g.K % 4 == 0);
  constexpr int ES = WK + 4;
  constexpr int LPR = WK / 4, RPI = 64 / LPR;          // lanes per row (one float4 each), rows per wave instruction
  float* w = reinterpret_cast<float*>(lds) + wave * (32 * ES);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2) + 4 * kh) * ES + j * 32 + l31] = acc[i][j][r];
#pragma unroll
    for (int p = 0; p < 32 / RPI; ++p) {
      const int rr = p * RPI + lane / LPR, cc = (lane % LPR) * 4;
      const int row = n0 + wm * 128 + i * 32 + rr, col = k0 + wn * WK + cc;
      const float4 v = *reinterpret_cast<const float4*>(w + rr * ES + cc);
      if (full) {
        *reinterpret_cast<float4*>(Cb + (size_t)row * g.K + col) = v;
      } else if (row < g.N) {
        float* cp = Cb + (size_t)row * g.K + col;
        if (col + 0 < g.K) cp[0] = v.x;
        if (col + 1 < g.K) cp[1] = v.y;
        if (col + 2 < g.K) cp[2] = v.z;
        if (col + 3 < g.K) cp[3] = v.w;
      }
    }
  }
}

// ---- grad-weight, second form ("rotated", hand-interleaved; 256 x 256 tiles): same partial-tile plan, product order and
// epilogue as emu_dw_kernel<MASK, 256> (results are bit identical without a sign bitmap), with the main loop rebuilt the way
// emu_kc2_kernel's was - here it matters more, because this kernel runs ONE wave per SIMD (256 accumulators) and nothing else
// covers a wave's conversion phase:
//  * a phase = [x1 y1, x1 y0, x2 y0 of slab s - 1 | x0 y2, x0 y1, x0 y0 of slab s] between two barriers (96 MFMAs): the 48 MFMAs
//    behind the barrier run on fragments read before it, every fragment read is 12 ... 48 MFMAs ahead of its use;
//  * the staging of slab s + 1 (a 4-column x 8-row patch per thread: 16 row pairs x {first plane, residual, second plane,
//    residual + third plane}, 12 LDS writes) is pinned unit by unit behind the MFMAs of the same wave (tools/gen/dw2_phase.py);
//    the patch sits in two half sets (columns 0-1 / 2-3 of its rows, 8-byte loads): a half is requested again for slab s + 2
//    the moment its two columns of slab s + 1 are converted, >= 56 MFMAs ahead of its next use - no second patch set
//    (the arch-VGPR half of the register file holds the fragments, 96, and the staging state; the accumulators fill the AGPRs);
//  * loads are buffer loads through a per-slab descriptor [first row of the slab, end of the row slice): rows past the slice
//    and the pad slab read as zero without a single select, columns past the operand by an out-of-range offset;
//  * the LDS column of output index n is n ^ ((n >> 3) & 3): with the plain layout the 16-byte writes of a patch (four
//    adjacent columns per lane = a 64-byte lane stride) hit two of the 32 store banks groups 4-way; the fragment reads
//    (32 consecutive columns per half-wave) stay conflict-free under the swizzle;
//  * the 1 / keep factor of the sign bitmap is applied once to the finished tile / bias-gradient partial, the bitmap itself
//    as a bit-extended and.
namespace {
constexpr int DSTAGE = 3 * 2 * DT * 2;                   // 16-byte units per stage: three planes x two chunks x (256 dy + 256 x columns)
}
template <bool MASK, bool HASDB>
__global__ __launch_bounds__(NT, 1) void emu_dw2_kernel(DwArgs g) {
  constexpr int A_U4 = 3 * 2 * DT;
  __shared__ __attribute__((aligned(16))) u32x4 s0[DSTAGE];
  __shared__ __attribute__((aligned(16))) u32x4 s1[DSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int ntile = g.tiles_n * g.tiles_k;
  const int bid = blockIdx.x;
  const int split = (bid & 7) + 8 * (bid / (8 * ntile));      // every slice of one tile on the same XCD (shared L2)
  const int t = (bid >> 3) % ntile;
  if (split >= g.splitk) return;
  const int tn = t / g.tiles_k, tk = t - tn * g.tiles_k;
  const int n0 = tn * DT, k0 = tk * DT;
  const int mbeg = split * g.m_per_split;
  const int mend = min(g.M, mbeg + g.m_per_split);
  const int nslab = (mend - mbeg + KS - 1) / KS;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging role of the wave: waves 0 / 1 the dy patch of chunk 0 / 1 (rows 0-7 / 8-15 of the slab), waves 2 / 3 the x patch
  const bool isA = wave < 2;
  const int c = wave & 1, cg = lane;
  const int col0 = (isA ? n0 : k0) + 4 * cg;
  const bool col_ok = col0 < (isA ? g.N : g.K);             // N, K multiples of 4: a column group is all in or all out
  const long ld = uni64(isA ? g.lddy : g.ldx);
  const char* opbase = reinterpret_cast<const char*>(g.dy) +
                       (long)uni64(isA ? 0ul : (uint64_t)(reinterpret_cast<const char*>(g.x) - reinterpret_cast<const char*>(g.dy)));
  // row e of the patch: one per-lane offset register + e * (row stride), added at the load (a scalar operand of the add); a lane whose
  // columns lie past the operand starts 1 GB out of range and reads zeros
  const int voff0 = col_ok ? (int)(((long)c * 8 * ld + col0) * 4) : 0x40000000;
  const int boff0 = (MASK && isA && col_ok) ? (int)(((long)c * 8 * g.ldbits + (col0 >> 5)) * 4) : 0x40000000;
  const int ldb4 = (int)(ld * 4), ldm4 = g.ldbits * 4;
  const uint32_t notA = isA ? 0u : 0xffffffffu;              // x patches carry no bitmap
  const int bsh = col0 & 31;                                 // the patch's four sign bits within its bitmap word
  constexpr bool SWZ = true;
  const int sw = SWZ ? (cg >> 1) & 3 : 0;
  const int wbase = (isA ? 0 : A_U4) + c * DT + 4 * cg;       // unit of the patch's first column in plane 0 (column j: + (j ^ sw))
  const int rsw = SWZ ? (l31 >> 3) & 3 : 0;
  const int aread = (wm * 128 + l31) ^ rsw, bread = A_U4 + ((wn * 128 + l31) ^ rsw);
  f32x2 rvL[8], rvH[8];                                       // the patch: columns 0-1 / 2-3 of its eight rows
  uint32_t rm[8], mpk = 0xffffffffu;                          // bitmap words of the slab in flight; the 8 x 4 sign bits of the patch being converted
  uint32_t t0[4], t1[4], t2[4];
  f32x2 rp_, fu_;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#define NOP_ ((void)0)
// DLDG(hf, e, sl): half hf (columns 2 hf, 2 hf + 1) of row e of the patch of slab sl through the slab's descriptor [first row of the
// slab, end of the slice);
// DLDM(e, sl): its bitmap word
#define DSLAB(sl)                                                                                                      \
    const int ms_ = mbeg + (sl) * KS;                                                                                  \
    const int left_ = max(mend - ms_, 0);                     /* (uniform) rows of the slice from this slab on */
#define DLDG(hf, e, sl)                                                                                                \
  do {                                                                                                                 \
    DSLAB(sl)                                                                                                          \
    const __amdgpu_buffer_rsrc_t r_ = __builtin_amdgcn_make_buffer_rsrc(                                               \
        const_cast<char*>(opbase + (size_t)ms_ * ld * 4), 0, (int)min((long)left_ * ld * 4, 0x3fffffffL), 0x00020000); \
    const f32x2 v_ = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_, voff0 + (e) * ldb4 + (hf) * 8, 0, 0)); \
    if ((hf) == 0) rvL[e] = v_; else rvH[e] = v_;                                                                      \
  } while (0)
#define DLDM(e, sl)                                                                                                    \
  do {                                                                                                                 \
    if (MASK) {                                                                                                        \
      DSLAB(sl)                                                                                                        \
      const __amdgpu_buffer_rsrc_t b_ = __builtin_amdgcn_make_buffer_rsrc(                                             \
          const_cast<uint32_t*>(g.bits + (size_t)ms_ * g.ldbits), 0, (int)min((long)left_ * g.ldbits * 4, 0x3fffffffL), 0x00020000); \
      rm[e] = __builtin_amdgcn_raw_buffer_load_b32(b_, boff0 + (e) * ldm4, 0, 0);                                      \
    }                                                                                                                  \
  } while (0)
#define PK_SUB(d, a, b) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b))
// the conversion of rows 2 pr, 2 pr + 1 of column j of the patch: DU1 bitmap + first plane (column 0 first packs the two rows' four
// sign bits into mpk - nibble e = row e - which frees the word registers for the next slab's words), DU2 first residual, DU3 second
// plane, DU4 second residual + third plane
#define DU1(j, pr)                                                                                                     \
  do {                                                                                                                 \
    f32x2 v_ = (j) < 2 ? f32x2{rvL[2 * (pr)][(j) & 1], rvL[2 * (pr) + 1][(j) & 1]} : f32x2{rvH[2 * (pr)][(j) & 1], rvH[2 * (pr) + 1][(j) & 1]}; \
    if (MASK) {                                                                                                        \
      if ((j) == 0) {                                                                                                  \
        const uint32_t n0_ = ((rm[2 * (pr)] | notA) >> bsh) & 0xfu, n1_ = ((rm[2 * (pr) + 1] | notA) >> bsh) & 0xfu;   \
        mpk = ((pr) == 0 ? 0u : mpk) | (n0_ << (8 * (pr))) | (n1_ << (8 * (pr) + 4));                                  \
      }                                                                                                                \
      float xa_, xb_;     /* (asm: see emu_kc2_kernel's U1) */                                                         \
      asm("v_and_b32 %0, %1, %2" : "=v"(xa_) : "v"(__builtin_amdgcn_sbfe((int)mpk, 8 * (pr) + (j), 1)), "v"(v_.x));    \
      asm("v_and_b32 %0, %1, %2" : "=v"(xb_) : "v"(__builtin_amdgcn_sbfe((int)mpk, 8 * (pr) + 4 + (j), 1)), "v"(v_.y)); \
      v_ = f32x2{xa_, xb_};                                                                                            \
    }                                                                                                                  \
    if (HASDB) csum[j] += v_.x + v_.y;                                                                                 \
    const uint32_t h_ = __builtin_bit_cast(uint32_t, __builtin_convertvector(v_, bf16x2));                             \
    t0[pr] = h_; rp_ = v_;                                                                                             \
    fu_ = f32x2{__builtin_bit_cast(float, h_ << 16), __builtin_bit_cast(float, h_ & 0xffff0000u)};                     \
  } while (0)
#define DU2(j, pr) PK_SUB(rp_, rp_, fu_)
#define DU3(j, pr)                                                                                                     \
  do {                                                                                                                 \
    const uint32_t h_ = __builtin_bit_cast(uint32_t, __builtin_convertvector(rp_, bf16x2));                            \
    t1[pr] = h_;                                                                                                       \
    fu_ = f32x2{__builtin_bit_cast(float, h_ << 16), __builtin_bit_cast(float, h_ & 0xffff0000u)};                     \
  } while (0)
#define DU4(j, pr) do { f32x2 w_; PK_SUB(w_, rp_, fu_); t2[pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(w_, bf16x2)); } while (0)
#define DSTA(st, j, pl) (st)[wbase + ((j) ^ sw) + (pl) * 2 * DT] = ((pl) == 0 ? u32x4{t0[0], t0[1], t0[2], t0[3]} : (pl) == 1 ? u32x4{t1[0], t1[1], t1[2], t1[3]} : u32x4{t2[0], t2[1], t2[2], t2[3]})
#define DLA(st, p, i) __builtin_bit_cast(bf16x8, (st)[aread + ((p) * 2 + kh) * DT + (i) * 32])
#define DLB(st, p, j) __builtin_bit_cast(bf16x8, (st)[bread + ((p) * 2 + kh) * DT + (j) * 32])
#define SB() __builtin_amdgcn_sched_barrier(0)
#define M1(ax, bx, i, j, work) do { acc[i][j] = MFB(ax[i], bx[j], acc[i][j]); work; SB(); } while (0)
#define MM(ax, bx) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = MFB(ax[i], bx[j], acc[i][j])
#define SYNC() do { SB(); __syncthreads(); SB(); } while (0)
#include "dw2_phase.inc"
#define DSTAGE_ALL(st)                                                                                                 \
  do {                                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
      _Pragma("unroll") for (int pr = 0; pr < 4; ++pr) { DU1(j, pr); DU2(j, pr); DU3(j, pr); DU4(j, pr); }             \
      DSTA(st, j, 0); DSTA(st, j, 1); DSTA(st, j, 2);                                                                  \
    }                                                                                                                  \
  } while (0)
#define DLOAD_ALL(sl) _Pragma("unroll") for (int e = 0; e < 8; ++e) { DLDG(0, e, sl); DLDG(1, e, sl); }
#define DLOADM_ALL(sl) _Pragma("unroll") for (int e = 0; e < 8; ++e) DLDM(e, sl)

  // the slab count is rounded up to an even number (a pad slab reads zeros through its empty descriptor); phases after the head
  // come in pairs plus one.
  const int nslab2 = (max(nslab, 1) + 1) & ~1;
  bf16x8 aX[4], aY[4], aZ[4], bP[4], bQ[4], bR[4];
  DLOAD_ALL(0);
  DLOADM_ALL(0);
  DSTAGE_ALL(s0);
  DLOAD_ALL(1);
  DLOADM_ALL(1);
  __syncthreads();
  // head (left to the compiler): the first half of slab 0, slab 1 -> s1, slab 2 requested
#pragma unroll
  for (int i = 0; i < 4; ++i) { aZ[i] = DLA(s0, 0, i); aX[i] = DLA(s0, 1, i); aY[i] = DLA(s0, 2, i); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { bQ[j] = DLB(s0, 0, j); bP[j] = DLB(s0, 1, j); bR[j] = DLB(s0, 2, j); }
  MM(aZ, bR); MM(aZ, bP); MM(aZ, bQ);                         // x0 y2, x0 y1, x0 y0 of slab 0; bP = y1, bQ = y0 stay for the next phase
  DSTAGE_ALL(s1);
  DLOAD_ALL(2);
  DLOADM_ALL(2);
  SYNC();
  for (int s = 1; s + 1 < nslab2; s += 2) {
    DPHASE(s1, s0, s, aX, aY, aZ, bP, bQ, bR);
    SYNC();
    DPHASE(s0, s1, s + 1, aX, aY, aZ, bR, bQ, bP);
    SYNC();
  }
  DPHASE(s1, s0, nslab2 - 1, aX, aY, aZ, bP, bQ, bR);
  SB();
  MM(aX, bR); MM(aX, bQ); MM(aY, bQ);                         // x1 y1, x1 y0, x2 y0 of the last slab
  __syncthreads();
#undef DSLAB
#undef DLDG
#undef DLDM
#undef DLOADM_ALL
#undef PK_SUB
#undef DU1
#undef DU2
#undef DU3
#undef DU4
#undef DSTA
#undef DLA
#undef DLB
#undef SB
#undef M1
#undef MM
#undef SYNC
#undef DPHASE
#undef DSTAGE_ALL
#undef DLOAD_ALL
#undef NOP_
  const float post = MASK ? g.ascale : 1.f;
  u32x4* lds = s0;
  // bias gradient partial: the two chunk threads of a column group add up through LDS
  if (HASDB && tk == 0) {
    float* red = reinterpret_cast<float*>(lds);
    if (isA) *reinterpret_cast<f32x4*>(&red[c * DT + 4 * cg]) = csum * post;
    __syncthreads();
    if (tid < DT) {
      const int n = n0 + tid;
      if (n < g.N) g.colsum[(size_t)split * g.colsum_split_stride + n] = red[tid] + red[DT + tid];
    }
    __syncthreads();
  }
  // epilogue: one row of 32 x 32 blocks (32 x 128) at a time through the wave's private LDS slice
  float* Cb = g.C + (size_t)split * g.c_split_stride;
  const bool full = (n0 + DT <= g.N) && (k0 + DT <= g.K) && (g.K % 4 == 0);
  constexpr int WK = DT / 2, ES = WK + 4, LPR = WK / 4, RPI = 64 / LPR;
  float* w = reinterpret_cast<float*>(lds) + wave * (32 * ES);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2) + 4 * kh) * ES + j * 32 + l31] = MASK ? acc[i][j][r] * post : acc[i][j][r];
#pragma unroll
    for (int p = 0; p < 32 / RPI; ++p) {
      const int rr = p * RPI + lane / LPR, cc = (lane % LPR) * 4;
      const int row = n0 + wm * 128 + i * 32 + rr, col = k0 + wn * WK + cc;
      const float4 v = *reinterpret_cast<const float4*>(w + rr * ES + cc);
      if (full) {
        *reinterpret_cast<float4*>(Cb + (size_t)row * g.K + col) = v;
      } else if (row < g.N) {
        float* cp = Cb + (size_t)row * g.K + col;
        if (col + 0 < g.K) cp[0] = v.x;
        if (col + 1 < g.K) cp[1] = v.y;
        if (col + 2 < g.K) cp[2] = v.z;
        if (col + 3 < g.K) cp[3] = v.w;
      }
    }
  }
}

// ---- f16x2 form of the 256 x 256 grad-weight tile (see "f16x2 form" above): both f32 operands scaled by their own power of two and
// split into hi + lo f16 pieces in the staging registers, three MFMA products per slab (tools/gen/dw2h_phase.py -> dw2h_phase.inc),
// stage = two planes (32 KB), the bias gradient from the unscaled values, dW scaled back in the epilogue.  The largest magnitudes
// come as magnitude words (common.h) from whoever produced dy and x, or from emu_amax_launch.
template <bool MASK, bool HASDB>
__global__ __launch_bounds__(NT, 1) void emu_dw2h_kernel(DwArgs g) {
  constexpr int A_U4 = 2 * 2 * DT;                           // two planes x two chunks x 256 columns
  constexpr int STG = 2 * A_U4;                              // dy + x: 16-byte units per stage (32 KB)
  constexpr int EPI = (4 * 32 * (DT / 2 + 4) * 4 + 15) / 16; // the epilogue's four transposition slices
  __shared__ __attribute__((aligned(16))) u32x4 lds_all[2 * STG > EPI ? 2 * STG : EPI];
  __shared__ uint32_t red4[4];
  u32x4* const s0 = lds_all;
  u32x4* const s1 = lds_all + STG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int ntile = g.tiles_n * g.tiles_k;
  const int bid = blockIdx.x;
  const int split = (bid & 7) + 8 * (bid / (8 * ntile));      // every slice of one tile on the same XCD (shared L2)
  const int t = (bid >> 3) % ntile;
  if (split >= g.splitk) return;
  const int tn = t / g.tiles_k, tk = t - tn * g.tiles_k;
  const int n0 = tn * DT, k0 = tk * DT;
  const int mbeg = split * g.m_per_split;
  const int mend = min(g.M, mbeg + g.m_per_split);
  const int nslab = (mend - mbeg + KS - 1) / KS;
  // operand scales from the row magnitudes (common.h) of THIS slice's rows (the contraction runs over them, so one scale per operand
  // and slice; the partial tile leaves unscaled): dy s_dy and x s_x in [2^13, 2^14) at the slice's largest element
  f16_saturate_on();
  uint32_t am_dy = 0u, am_x = 0u;
  for (int i = mbeg + (int)threadIdx.x; i < mend; i += NT) { am_dy = max(am_dy, g.dy_amax[i]); am_x = max(am_x, g.x_amax[i]); }
  am_dy = block_max_u32(am_dy, red4);
  __syncthreads();
  am_x = block_max_u32(am_x, red4);

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging role of the wave: waves 0 / 1 the dy patch of chunk 0 / 1 (rows 0-7 / 8-15 of the slab), waves 2 / 3 the x patch
  const bool isA = wave < 2;
  const int c = wave & 1, cg = lane;
  const int col0 = (isA ? n0 : k0) + 4 * cg;
  const bool col_ok = col0 < (isA ? g.N : g.K);             // N, K multiples of 4: a column group is all in or all out
  const long ld = uni64(isA ? g.lddy : g.ldx);
  const char* opbase = reinterpret_cast<const char*>(g.dy) +
                       (long)uni64(isA ? 0ul : (uint64_t)(reinterpret_cast<const char*>(g.x) - reinterpret_cast<const char*>(g.dy)));
  // row e of the patch: one per-lane offset register + e * (row stride), added at the load (a scalar operand of the add); a lane whose
  // columns lie past the operand starts 1 GB out of range and reads zeros
  const int voff0 = col_ok ? (int)(((long)c * 8 * ld + col0) * 4) : 0x40000000;
  const int boff0 = (MASK && isA && col_ok) ? (int)(((long)c * 8 * g.ldbits + (col0 >> 5)) * 4) : 0x40000000;
  const int ldb4 = (int)(ld * 4), ldm4 = g.ldbits * 4;
  const uint32_t notA = isA ? 0u : 0xffffffffu;              // x patches carry no bitmap
  const int bsh = col0 & 31;                                 // the patch's four sign bits within its bitmap word
  constexpr bool SWZ = true;
  const int sw = SWZ ? (cg >> 1) & 3 : 0;
  const int wbase = (isA ? 0 : A_U4) + c * DT + 4 * cg;       // unit of the patch's first column in plane 0 (column j: + (j ^ sw))
  const int rsw = SWZ ? (l31 >> 3) & 3 : 0;
  const int aread = (wm * 128 + l31) ^ rsw, bread = A_U4 + ((wn * 128 + l31) ^ rsw);
  f32x2 rvL[8], rvH[8];                                       // the patch: columns 0-1 / 2-3 of its eight rows
  uint32_t rm[8], mpk = 0xffffffffu;                          // bitmap words of the slab in flight; the 8 x 4 sign bits of the patch being converted
  uint32_t t0[4], t1[4];
  const float sc = isA ? h2_scale(am_dy) : h2_scale(am_x);   // (wave-uniform)
  f32x2 rp_, fu_;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#define NOP_ ((void)0)
// DLDG(hf, e, sl): half hf (columns 2 hf, 2 hf + 1) of row e of the patch of slab sl through the slab's descriptor [first row of the
// slab, end of the slice);
// DLDM(e, sl): its bitmap word
#define DSLAB(sl)                                                                                                      \
    const int ms_ = mbeg + (sl) * KS;                                                                                  \
    const int left_ = max(mend - ms_, 0);                     /* (uniform) rows of the slice from this slab on */
#define DLDG(hf, e, sl)                                                                                                \
  do {                                                                                                                 \
    DSLAB(sl)                                                                                                          \
    const __amdgpu_buffer_rsrc_t r_ = __builtin_amdgcn_make_buffer_rsrc(                                               \
        const_cast<char*>(opbase + (size_t)ms_ * ld * 4), 0, (int)min((long)left_ * ld * 4, 0x3fffffffL), 0x00020000); \
    const f32x2 v_ = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_, voff0 + (e) * ldb4 + (hf) * 8, 0, 0)); \
    if ((hf) == 0) rvL[e] = v_; else rvH[e] = v_;                                                                      \
  } while (0)
#define DLDM(e, sl)                                                                                                    \
  do {                                                                                                                 \
    if (MASK) {                                                                                                        \
      DSLAB(sl)                                                                                                        \
      const __amdgpu_buffer_rsrc_t b_ = __builtin_amdgcn_make_buffer_rsrc(                                             \
          const_cast<uint32_t*>(g.bits + (size_t)ms_ * g.ldbits), 0, (int)min((long)left_ * g.ldbits * 4, 0x3fffffffL), 0x00020000); \
      rm[e] = __builtin_amdgcn_raw_buffer_load_b32(b_, boff0 + (e) * ldm4, 0, 0);                                      \
    }                                                                                                                  \
  } while (0)
#define PK_SUB(d, a, b) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b))
// the conversion of rows 2 pr, 2 pr + 1 of column j of the patch: DU1 bitmap + first plane (column 0 first packs the two rows' four
// sign bits into mpk - nibble e = row e - which frees the word registers for the next slab's words), DU2 first residual, DU3 second
// plane, DU4 second residual + third plane
#define DU1(j, pr)                                                                                                     \
  do {                                                                                                                 \
    f32x2 v_ = (j) < 2 ? f32x2{rvL[2 * (pr)][(j) & 1], rvL[2 * (pr) + 1][(j) & 1]} : f32x2{rvH[2 * (pr)][(j) & 1], rvH[2 * (pr) + 1][(j) & 1]}; \
    if (MASK) {                                                                                                        \
      if ((j) == 0) {                                                                                                  \
        const uint32_t n0_ = ((rm[2 * (pr)] | notA) >> bsh) & 0xfu, n1_ = ((rm[2 * (pr) + 1] | notA) >> bsh) & 0xfu;   \
        mpk = ((pr) == 0 ? 0u : mpk) | (n0_ << (8 * (pr))) | (n1_ << (8 * (pr) + 4));                                  \
      }                                                                                                                \
      float xa_, xb_;     /* (asm: see emu_kc2_kernel's U1) */                                                         \
      asm("v_and_b32 %0, %1, %2" : "=v"(xa_) : "v"(__builtin_amdgcn_sbfe((int)mpk, 8 * (pr) + (j), 1)), "v"(v_.x));    \
      asm("v_and_b32 %0, %1, %2" : "=v"(xb_) : "v"(__builtin_amdgcn_sbfe((int)mpk, 8 * (pr) + 4 + (j), 1)), "v"(v_.y)); \
      v_ = f32x2{xa_, xb_};                                                                                            \
    }                                                                                                                  \
    if (HASDB) csum[j] += v_.x + v_.y;                                                                                 \
    v_ *= sc;                                                                                                          \
    const f16x2 h_ = __builtin_convertvector(v_, f16x2);      /* v_cvt_pk_f16_f32, round to nearest */                 \
    t0[pr] = __builtin_bit_cast(uint32_t, h_); rp_ = v_;                                                               \
    fu_ = __builtin_convertvector(h_, f32x2);                                                                          \
  } while (0)
#define DU2(j, pr) do { f32x2 w_; PK_SUB(w_, rp_, fu_); t1[pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(w_, f16x2)); } while (0)
#define DSTA(st, j, pl) (st)[wbase + ((j) ^ sw) + (pl) * 2 * DT] = ((pl) == 0 ? u32x4{t0[0], t0[1], t0[2], t0[3]} : u32x4{t1[0], t1[1], t1[2], t1[3]})
#define DLA(st, p, i) __builtin_bit_cast(f16x8, (st)[aread + ((p) * 2 + kh) * DT + (i) * 32])
#define DLB(st, p, j) __builtin_bit_cast(f16x8, (st)[bread + ((p) * 2 + kh) * DT + (j) * 32])
#define SB() __builtin_amdgcn_sched_barrier(0)
#define M1(ax, bx, i, j, work) do { acc[i][j] = MFH(ax[i], bx[j], acc[i][j]); work; SB(); } while (0)
#define MM(ax, bx) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = MFH(ax[i], bx[j], acc[i][j])
#define SYNC() do { SB(); __syncthreads(); SB(); } while (0)
#include "dw2h_phase.inc"
#define DSTAGE_ALL(st)                                                                                                 \
  do {                                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
      _Pragma("unroll") for (int pr = 0; pr < 4; ++pr) { DU1(j, pr); DU2(j, pr); }                                     \
      DSTA(st, j, 0); DSTA(st, j, 1);                                                                                  \
    }                                                                                                                  \
  } while (0)
#define DLOAD_ALL(sl) _Pragma("unroll") for (int e = 0; e < 8; ++e) { DLDG(0, e, sl); DLDG(1, e, sl); }
#define DLOADM_ALL(sl) _Pragma("unroll") for (int e = 0; e < 8; ++e) DLDM(e, sl)

  // the slab count is rounded up to an even number (a pad slab reads zeros through its empty descriptor); phases after the head
  // come in pairs plus one.
  const int nslab2 = (max(nslab, 1) + 1) & ~1;
  f16x8 aH[4], aL[4], bP[4], bQ[4], bL[4];
  DLOAD_ALL(0);
  DLOADM_ALL(0);
  DSTAGE_ALL(s0);
  DLOAD_ALL(1);
  DLOADM_ALL(1);
  __syncthreads();
  // head (left to the compiler): the first half of slab 0, slab 1 -> s1, slab 2 requested
#pragma unroll
  for (int i = 0; i < 4; ++i) { aL[i] = DLA(s0, 1, i); aH[i] = DLA(s0, 0, i); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { bP[j] = DLB(s0, 0, j); bL[j] = DLB(s0, 1, j); }
  MM(aL, bP); MM(aH, bL);                                     // lo hi, hi lo of slab 0; aH / bP = its hi pieces stay for the next phase
  DSTAGE_ALL(s1);
  DLOAD_ALL(2);
  DLOADM_ALL(2);
  SYNC();
  for (int s = 1; s + 1 < nslab2; s += 2) {
    DHPHASE(s1, s0, s, bP, bQ);
    SYNC();
    DHPHASE(s0, s1, s + 1, bQ, bP);
    SYNC();
  }
  DHPHASE(s1, s0, nslab2 - 1, bP, bQ);
  SB();
  MM(aH, bQ);                                                 // hi hi of the last slab
  __syncthreads();
#undef DSLAB
#undef DLDG
#undef DLDM
#undef DLOADM_ALL
#undef PK_SUB
#undef DU1
#undef DU2
#undef DSTA
#undef DLA
#undef DLB
#undef SB
#undef M1
#undef MM
#undef SYNC
#undef DHPHASE
#undef DSTAGE_ALL
#undef DLOAD_ALL
#undef NOP_
  const float unscale = h2_inv_scale(am_dy) * h2_inv_scale(am_x);
  const float post = MASK ? g.ascale : 1.f;
  u32x4* lds = lds_all;
  // bias gradient partial: the two chunk threads of a column group add up through LDS
  if (HASDB && tk == 0) {
    float* red = reinterpret_cast<float*>(lds);
    if (isA) *reinterpret_cast<f32x4*>(&red[c * DT + 4 * cg]) = csum * post;
    __syncthreads();
    if (tid < DT) {
      const int n = n0 + tid;
      if (n < g.N) g.colsum[(size_t)split * g.colsum_split_stride + n] = red[tid] + red[DT + tid];
    }
    __syncthreads();
  }
  // epilogue: one row of 32 x 32 blocks (32 x 128) at a time through the wave's private LDS slice
  float* Cb = g.C + (size_t)split * g.c_split_stride;
  const bool full = (n0 + DT <= g.N) && (k0 + DT <= g.K) && (g.K % 4 == 0);
  constexpr int WK = DT / 2, ES = WK + 4, LPR = WK / 4, RPI = 64 / LPR;
  float* w = reinterpret_cast<float*>(lds) + wave * (32 * ES);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2) + 4 * kh) * ES + j * 32 + l31] = acc[i][j][r] * (MASK ? post * unscale : unscale);
#pragma unroll
    for (int p = 0; p < 32 / RPI; ++p) {
      const int rr = p * RPI + lane / LPR, cc = (lane % LPR) * 4;
      const int row = n0 + wm * 128 + i * 32 + rr, col = k0 + wn * WK + cc;
      const float4 v = *reinterpret_cast<const float4*>(w + rr * ES + cc);
      if (full) {
        *reinterpret_cast<float4*>(Cb + (size_t)row * g.K + col) = v;
      } else if (row < g.N) {
        float* cp = Cb + (size_t)row * g.K + col;
        if (col + 0 < g.K) cp[0] = v.x;
        if (col + 1 < g.K) cp[1] = v.y;
        if (col + 2 < g.K) cp[2] = v.z;
        if (col + 3 < g.K) cp[3] = v.w;
      }
    }
  }
}

// out[i] = sum_s part[s * stride + i], deterministic: a block owns 256 consecutive floats (64 lanes x float4), its 16 waves sum
// the slices s = w, w + 16, ... in order (16 independent 1 KB streams per block keep the loads in flight) and the 16 partial sums
// are combined in wave order through LDS.  n must be a multiple of 4 (N * K and N are).
// Two reductions in one launch (dW and db of a grad-weight call): blocks [0, blocks0) serve (part, stride, out, n), the rest
// (part1, stride1, out1, n1).
__global__ __launch_bounds__(1024) void emu_reduce_partials_kernel(const float* __restrict__ part, long stride, int splits,
                                                                   float* __restrict__ out, long n, int blocks0,
                                                                   const float* __restrict__ part1, long stride1,
                                                                   float* __restrict__ out1, long n1) {
  __shared__ float4 red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int blk = blockIdx.x;
  if (blk >= blocks0) { blk -= blocks0; part = part1; stride = stride1; out = out1; n = n1; }
  const long i = ((long)blk * 64 + lane) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
    const float* p = part + i;
    int k = w;
    for (; k + 16 < splits; k += 32) {
      const float4 u = *reinterpret_cast<const float4*>(p + (size_t)k * stride);
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)(k + 16) * stride);
      s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (k < splits) {
      const float4 u = *reinterpret_cast<const float4*>(p + (size_t)k * stride);
      s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
    }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && i < n) {
    float4 t = red[0][lane];
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      const float4 v = red[j][lane];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = t;
  }
}

namespace {
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// HOISDF_EMU_FORM: "h2" (default) = the f16x2 form, "b3" = bf16x3 (six products).  Process-wide: it fixes the weight-image format.
bool form_h2() {
  static int f = -1;
  if (f < 0) { const char* e = getenv("HOISDF_EMU_FORM"); f = (e && (e[0] == 'b' || e[0] == 'B')) ? 0 : 1; }
  return f == 1;
}

}  // namespace

// row / head magnitudes for an operand nobody described: stream-ordered scratch from one arena per (device, stream) - the stream
// handle alone is not a key (torch's default stream is handle 0 on every device).  Everything that reads a slot runs on the slot's
// stream behind the pass that filled it, so wrapping around is safe whatever the arena's size; an arena that is too small for a
// request is replaced by a larger one (the old one stays allocated: launches in flight may still read it).  The first use on a
// stream allocates (hipMalloc synchronises and is illegal under graph capture: hosts that capture pass their own words).
uint32_t* mag_scratch(hipStream_t st, long words) {
  struct Arena { char* base = nullptr; size_t cap = 0, off = 0; };
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, Arena> arenas;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  const size_t bytes = ((size_t)(words > 0 ? words : 1) * 4 + 255) & ~(size_t)255;
  std::lock_guard<std::mutex> lk(mu);
  Arena& a = arenas[std::make_pair(dev, st)];
  if (!a.base || bytes * 4 > a.cap) {
    const size_t cap = bytes * 8 > ((size_t)32 << 20) ? bytes * 8 : ((size_t)32 << 20);
    char* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), cap) != hipSuccess) return nullptr;
    a.base = p; a.cap = cap; a.off = 0;
  }
  if (a.off + bytes > a.cap) a.off = 0;
  uint32_t* r = reinterpret_cast<uint32_t*>(a.base + a.off);
  a.off += bytes;
  return r;
}

// HOISDF_MAG_TRACE=1: every operand the library had to measure itself, on stderr (who asked, rows x columns) - which producers to teach
static void mag_trace(const char* who, long M, int K) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("HOISDF_MAG_TRACE"); on = (e && atoi(e) != 0) ? 1 : 0; }
  if (on) fprintf(stderr, "[hoisdf mag] measured by the library: %s operand %ld x %d\n", who, M, K);
}
// row magnitudes of a row-major matrix into `words` (M words, every one written)
int emu_rowmag_launch(const float* x, long ld, long M, int K, uint32_t* words, hipStream_t st) {
  if (M <= 0) return HOISDF_OK;
  hipLaunchKernelGGL(emu_rowmag_kernel, dim3((unsigned)cdiv(M, 32)), dim3(256), 0, st, x, ld, M, K, words);
  return check_launch("emu_rowmag");
}
// the same pass for an operand several contractions will read
int emu_mag_measure(const float* x, long ld, long M, int K, uint32_t* words, hipStream_t st) {
  mag_trace("a chain, once for all its readers:", M, K);
  return emu_rowmag_launch(x, ld, M, K, words, st);
}
// head magnitudes (common.h) of x[M][groups * 64 ...]: words[group * nb + row / L], nb = ceil(M / L); cleared here
int emu_headmag_launch(const float* x, long ld, long M, int groups, int L, uint32_t* words, hipStream_t st) {
  const int nb = cdiv(M, L);
  if (hipMemsetAsync(words, 0, (size_t)groups * nb * 4, st) != hipSuccess) { set_error("head magnitudes: memset failed"); return HOISDF_ERR_LAUNCH; }
  if (M <= 0) return HOISDF_OK;
  hipLaunchKernelGGL(emu_headmag_kernel, dim3((unsigned)cdiv(M, 64)), dim3(256), 0, st, x, ld, M, groups, L, nb, words);
  return check_launch("emu_headmag");
}
bool emu_form_h2() { return form_h2(); }

namespace {
int launch_emu(EmuArgs g, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(emu_kc_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(emu_kc_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_BYTES) != hipSuccess) {
      set_error("linear_emu: cannot raise the dynamic LDS limit to %u bytes", LDS_BYTES);
      return HOISDF_ERR_LAUNCH;
    }
    attr_set = true;
  }
  g.vecC = al16(g.C) && (g.ldc % 4 == 0);
  if (g.beta) g.amax_out = nullptr;           // (the tile is added to what is there: its own magnitude says nothing)
  if (form_h2()) {
    // tile width: 256 x 128, two workgroups per CU (the prologue / epilogue of one under the main loop of the other; finer tiles for
    // the 16 384 / 49 152-row shapes) except for the masked grad-input over a long contraction, where the 256 x 256 tile's halved
    // staging work per MFMA wins (tools/mb_kc2.py, profiles/r05_h2_tile_widths.txt: 65536 x 1024 x 256 forward + ReLU + dropout
    // 180 vs 149 TF, 49152 x 512 x 512 247 vs 212, 65536 x 256 x 256 176 vs 159; masked grad-input over 1024: 172 vs 181).
    // HOISDF_H2_TILE=128 / 256 forces one (A/B runs).
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("HOISDF_H2_TILE"); const int v = e ? atoi(e) : 0; forced = (v == 128 || v == 256) ? v : 0; }
    const bool wide_ok = cdiv(g.M, HTM) * cdiv(g.N, HTN) >= 208 && g.N % HTN == 0;
    const bool narrow = forced ? forced == 128 : !(g.abits && g.K >= 768 && wide_ok);
    const int tw = narrow ? 128 : HTN;
    g.tiles_m = cdiv(g.M, HTM);
    g.tiles_n = cdiv(g.N, tw);
    g.b_scale = reinterpret_cast<const float*>(h2_trailer(const_cast<u32x4*>(g.Bimg), g.N, g.K)) + 16;
    if (!g.a_amax) {
      uint32_t* part = mag_scratch(st, g.M);
      if (!part) { set_error("linear_emu: cannot allocate the row magnitudes"); return HOISDF_ERR_LAUNCH; }
      mag_trace(g.beta ? "grad-input (+=)" : g.abits ? "grad-input (masked)" : g.qkv.on ? "in-projection" : "forward / grad-input", g.M, g.K);
      if (int rc = emu_rowmag_launch(g.A, g.lda, g.M, g.K, part, st)) return rc;
      g.a_amax = part;
    }
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(NT);
    const bool kt = g.K % KS != 0 || (cdiv(g.K, KS) & 1);
    if (narrow) {
      if (g.abits && kt) hipLaunchKernelGGL((emu_h2_kernel<true, true, 2>), grid, block, 0, st, g);
      else if (g.abits) hipLaunchKernelGGL((emu_h2_kernel<true, false, 2>), grid, block, 0, st, g);
      else if (kt) hipLaunchKernelGGL((emu_h2_kernel<false, true, 2>), grid, block, 0, st, g);
      else hipLaunchKernelGGL((emu_h2_kernel<false, false, 2>), grid, block, 0, st, g);
    } else if (g.abits && kt) hipLaunchKernelGGL((emu_h2_kernel<true, true, 4>), grid, block, 0, st, g);
    else if (g.abits) hipLaunchKernelGGL((emu_h2_kernel<true, false, 4>), grid, block, 0, st, g);
    else if (kt) hipLaunchKernelGGL((emu_h2_kernel<false, true, 4>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((emu_h2_kernel<false, false, 4>), grid, block, 0, st, g);
    return check_launch("linear_emu (f16x2)");
  }
  g.tiles_m = cdiv(g.M, TM);
  g.tiles_n = cdiv(g.N, TN);
  const dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(NT);
  static int form = -1;                       // HOISDF_EMU_KC=1: the first main-loop form (A/B runs); default: the rotated form
  if (form < 0) { const char* e = getenv("HOISDF_EMU_KC"); form = (e && atoi(e) == 1) ? 1 : 2; }
  if (form == 2) {
    const bool kt = g.K % KS != 0 || (cdiv(g.K, KS) & 1);      // a pad slab (odd slab count) stages zeros through the k-tail test
    if (g.abits && kt) hipLaunchKernelGGL((emu_kc2_kernel<true, true>), grid, block, 0, st, g);
    else if (g.abits) hipLaunchKernelGGL((emu_kc2_kernel<true, false>), grid, block, 0, st, g);
    else if (kt) hipLaunchKernelGGL((emu_kc2_kernel<false, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((emu_kc2_kernel<false, false>), grid, block, 0, st, g);
  } else if (g.abits) hipLaunchKernelGGL((emu_kc_kernel<true>), grid, block, LDS_BYTES, st, g);
  else hipLaunchKernelGGL((emu_kc_kernel<false>), grid, block, LDS_BYTES, st, g);
  return check_launch("linear_emu");
}
}  // namespace

}  // namespace hoisdf

using namespace hoisdf;

extern "C" long hoisdf_linear_emu_image_bytes(int rows, int K) {
  if (rows <= 0 || K <= 0) return 0;
  if (form_h2()) return (long)cdiv(rows, HTN) * cdiv(K, KS) * HB_U4 * 16 + H_TRAILER;
  return (long)cdiv(rows, TN) * cdiv(K, KS) * B_U4 * 16;
}

extern "C" int hoisdf_linear_emu_prepare(const float* W, int ldw, int N, int K, int transpose, void* image, void* stream) {
  HOISDF_REQUIRE(W && image && N > 0 && K > 0 && ldw >= K, HOISDF_ERR_INVALID, "linear_emu_prepare: bad arguments");
  HOISDF_REQUIRE(al16(image), HOISDF_ERR_INVALID, "linear_emu_prepare: the image must be 16-byte aligned");
  const int R = transpose ? K : N, Kc = transpose ? N : K;
  const int nslab = cdiv(Kc, KS);
  if (form_h2()) {
    const long total = (long)cdiv(R, HTN) * nslab * 2 * HTN;
    hipLaunchKernelGGL(h2_weight_amax_kernel, dim3(16), dim3(256), 0, as_stream(stream), W, ldw, N, K, R, Kc, image);
    hipLaunchKernelGGL(h2_prep_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), W, ldw, R, Kc,
                       transpose, nslab, total, static_cast<u32x4*>(image));
    return check_launch("linear_emu_prepare (f16x2)");
  }
  const long total = (long)cdiv(R, TN) * nslab * 2 * TN;
  hipLaunchKernelGGL(emu_prep_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), W, ldw, R, Kc,
                     transpose, nslab, total, static_cast<u32x4*>(image));
  return check_launch("linear_emu_prepare");
}

extern "C" long hoisdf_linear_emu_prepare_blocks(int N, int K, int transpose) {
  if (N <= 0 || K <= 0) return 0;
  const int R = transpose ? K : N, Kc = transpose ? N : K;
  if (form_h2()) return ((long)cdiv(R, HTN) * cdiv(Kc, KS) * 2 * HTN + 255) / 256;
  return ((long)cdiv(R, TN) * cdiv(Kc, KS) * 2 * TN + 255) / 256;
}

extern "C" int hoisdf_linear_emu_prepare_batch(const hoisdf_emu_prep_item* d_items, int n, long total_blocks, void* stream) {
  HOISDF_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < (1L << 31), HOISDF_ERR_INVALID, "linear_emu_prepare_batch: bad sizes");
  if (n == 0 || total_blocks == 0) return HOISDF_OK;
  HOISDF_REQUIRE(d_items, HOISDF_ERR_INVALID, "linear_emu_prepare_batch: null table");
  if (form_h2()) {
    hipLaunchKernelGGL(h2_weight_amax_batch_kernel, dim3((unsigned)n * 16), dim3(256), 0, as_stream(stream), d_items);
    hipLaunchKernelGGL(h2_prep_weight_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), d_items, n);
    return check_launch("linear_emu_prepare_batch (f16x2)");
  }
  hipLaunchKernelGGL(emu_prep_weight_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), d_items, n);
  return check_launch("linear_emu_prepare_batch");
}

extern "C" int hoisdf_linear_emu_supported(const float* a, long lda, int Kc) {
  return a && al16(a) && (lda % 4 == 0) && (Kc % 4 == 0) && Kc >= 4;
}

extern "C" int hoisdf_linear_fwd_emu(const float* x, int ldx, const void* w_image, const float* bias, float* y, int ldy, long M,
                                     int N, int K, int act, float drop_p, uint64_t seed, uint32_t* relu_bits, void* stream) {
  return linear_fwd_emu_mag(x, ldx, w_image, bias, y, ldy, M, N, K, act, drop_p, seed, relu_bits, nullptr, nullptr, stream);
}
extern "C" int hoisdf_linear_fwd_emu_mag(const float* x, int ldx, const void* w_image, const float* bias, float* y, int ldy, long M,
                                         int N, int K, int act, float drop_p, uint64_t seed, uint32_t* relu_bits, const uint32_t* x_mag,
                                         uint32_t* y_mag, void* stream) {
  return linear_fwd_emu_mag(x, ldx, w_image, bias, y, ldy, M, N, K, act, drop_p, seed, relu_bits, x_mag, y_mag, stream);
}
// row magnitudes (include/hoisdf.h): u32 words a matrix of `rows` rows takes
extern "C" long hoisdf_mag_words(long rows) { return rows > 0 ? rows : 0; }
// the row magnitudes of a matrix nobody left words for: one read of x (hosts that chain the *_mag entries themselves call this once per
// operand instead of letting every contraction measure it again); every word is written, no clearing needed
extern "C" int hoisdf_mag_measure(const float* x, long ldx, long M, int K, uint32_t* words, void* stream) {
  HOISDF_REQUIRE(words && (M == 0 || x) && M >= 0 && K > 0 && ldx >= K, HOISDF_ERR_INVALID, "mag_measure: bad arguments");
  HOISDF_REQUIRE(M == 0 || hoisdf_linear_emu_supported(x, ldx, K), HOISDF_ERR_INVALID, "mag_measure: x must be 16-byte aligned with ldx and K multiples of 4");
  return emu_rowmag_launch(x, ldx, M, K, words, as_stream(stream));
}
// head magnitudes (include/hoisdf.h) of an attention operand matrix x[M][>= groups * 64], samples of L rows: words a host must provide,
// and the pass that fills them (clears first)
extern "C" long hoisdf_head_mag_words(long M, int groups, int L) { return (M > 0 && groups > 0 && L > 0) ? (long)groups * cdiv(M, L) : 0; }
extern "C" int hoisdf_head_mag_measure(const float* x, long ldx, long M, int groups, int L, uint32_t* words, void* stream) {
  HOISDF_REQUIRE(words && (M == 0 || x) && M >= 0 && groups > 0 && L > 0 && ldx >= (long)groups * 64 && ldx % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0, HOISDF_ERR_INVALID, "head_mag_measure: bad arguments");
  return emu_headmag_launch(x, ldx, M, groups, L, words, as_stream(stream));
}
extern "C" int hoisdf_linear_emu_pieces(void) { return form_h2() ? 2 : 3; }

int hoisdf::linear_fwd_emu_mag(const float* x, int ldx, const void* w_image, const float* bias, float* y, int ldy, long M, int N, int K,
                               int act, float drop_p, uint64_t seed, uint32_t* relu_bits, const uint32_t* x_mag, uint32_t* y_mag,
                               void* stream, uint32_t* y_heads, int head_L) {
  HOISDF_REQUIRE(M == 0 || (x && w_image && y), HOISDF_ERR_INVALID, "linear_fwd_emu: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldy >= N && M < (1L << 31), HOISDF_ERR_INVALID,
                 "linear_fwd_emu: bad sizes M=%ld N=%d K=%d ldx=%d ldy=%d", M, N, K, ldx, ldy);
  HOISDF_REQUIRE(drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID, "linear_fwd_emu: drop_p=%f", drop_p);
  if (M == 0) return HOISDF_OK;
  HOISDF_REQUIRE(hoisdf_linear_emu_supported(x, ldx, K), HOISDF_ERR_INVALID,
                 "linear_fwd_emu: x must be 16-byte aligned with ldx and K multiples of 4 (use hoisdf_linear_fwd otherwise)");
  EmuArgs g{};
  g.A = x; g.lda = ldx; g.Bimg = static_cast<const u32x4*>(w_image);
  g.C = y; g.ldc = ldy; g.bias = bias; g.M = (int)M; g.N = N; g.K = K;
  g.act = act; g.drop_p = drop_p; g.inv_keep = 1.f / (1.f - drop_p); g.thresh = drop_threshold(drop_p); g.seed = seed;
  g.bits_out = relu_bits; g.ldbits_out = (N + 31) / 32;
  g.a_amax = x_mag; g.amax_out = y_mag;
  if (y_heads) {
    HOISDF_REQUIRE(head_L > 0 && N % 64 == 0, HOISDF_ERR_INVALID, "linear_fwd_emu: head magnitudes need N %% 64 == 0 and the rows per sample");
    g.head_out = y_heads; g.head_L = head_L; g.head_nb = cdiv(M, head_L);
  }
  return launch_emu(g, as_stream(stream));
}

// hoisdf_linear_fwd_emu whose output goes into attention planes (common.h QkvPlanes; internal: the coarse layer entries use it)
int hoisdf::linear_fwd_emu_qkv(const float* x, int ldx, const void* w_image, const float* bias, long M, int N, int K,
                               const QkvPlanes& pl, void* stream, const uint32_t* x_mag) {
  HOISDF_REQUIRE(x && w_image && M > 0 && N > 0 && K > 0 && ldx >= K && M < (1L << 31), HOISDF_ERR_INVALID, "linear_fwd_emu_qkv: bad arguments");
  HOISDF_REQUIRE(hoisdf_linear_emu_supported(x, ldx, K), HOISDF_ERR_INVALID, "linear_fwd_emu_qkv: x alignment / K");
  HOISDF_REQUIRE(pl.on && pl.L > 0 && pl.L % 128 == 0 && M % pl.L == 0 && N % 64 == 0 && pl.E % 64 == 0 && pl.Lp >= pl.L &&
                     pl.col0 % 64 == 0 && pl.col0 + N <= 3 * pl.E,
                 HOISDF_ERR_INVALID, "linear_fwd_emu_qkv: plane geometry (L=%d Lp=%d N=%d E=%d col0=%d)", pl.L, pl.Lp, N, pl.E, pl.col0);
  EmuArgs g{};
  g.A = x; g.lda = ldx; g.Bimg = static_cast<const u32x4*>(w_image);
  g.C = nullptr; g.ldc = N; g.bias = bias; g.M = (int)M; g.N = N; g.K = K;
  g.inv_keep = 1.f; g.qkv = pl;
  g.a_amax = x_mag;
  return launch_emu(g, as_stream(stream));
}

extern "C" int hoisdf_linear_fwd_emu_heads(const float* x, int ldx, const void* w_image, const float* bias, float* y, int ldy, long M, int N,
                                           int K, const uint32_t* x_mag, uint32_t* y_mag, uint32_t* y_heads, int L, void* stream) {
  HOISDF_REQUIRE(y_heads, HOISDF_ERR_INVALID, "linear_fwd_emu_heads: y_heads is required");
  return linear_fwd_emu_mag(x, ldx, w_image, bias, y, ldy, M, N, K, 0, 0.f, 0, nullptr, x_mag, y_mag, stream, y_heads, L);
}
extern "C" int hoisdf_linear_bwd_input_emu_heads(const float* dy, int lddy, const void* wt_image, float* dx, int lddx, long M, int N, int K,
                                                 const uint32_t* dy_mag, uint32_t* dx_mag, uint32_t* dx_heads, int L, void* stream) {
  HOISDF_REQUIRE(dx_heads, HOISDF_ERR_INVALID, "linear_bwd_input_emu_heads: dx_heads is required");
  return linear_bwd_input_emu_mag(dy, lddy, nullptr, 0.f, wt_image, dx, lddx, M, N, K, 0, dy_mag, dx_mag, stream, dx_heads, L);
}
extern "C" int hoisdf_linear_bwd_input_emu(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p,
                                           const void* wt_image, float* dx, int lddx, long M, int N, int K, int accumulate,
                                           void* stream) {
  return linear_bwd_input_emu_mag(dy, lddy, relu_bits, drop_p, wt_image, dx, lddx, M, N, K, accumulate, nullptr, nullptr, stream);
}
extern "C" int hoisdf_linear_bwd_input_emu_mag(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p,
                                               const void* wt_image, float* dx, int lddx, long M, int N, int K, int accumulate,
                                               const uint32_t* dy_mag, uint32_t* dx_mag, void* stream) {
  return linear_bwd_input_emu_mag(dy, lddy, relu_bits, drop_p, wt_image, dx, lddx, M, N, K, accumulate, dy_mag, dx_mag, stream);
}

int hoisdf::linear_bwd_input_emu_mag(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const void* wt_image, float* dx,
                                     int lddx, long M, int N, int K, int accumulate, const uint32_t* dy_mag, uint32_t* dx_mag,
                                     void* stream, uint32_t* dx_heads, int head_L) {
  HOISDF_REQUIRE(M == 0 || (dy && wt_image && dx), HOISDF_ERR_INVALID, "linear_bwd_input_emu: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && lddx >= K && M < (1L << 31) && drop_p >= 0.f && drop_p < 1.f,
                 HOISDF_ERR_INVALID, "linear_bwd_input_emu: bad sizes");
  if (M == 0) return HOISDF_OK;
  HOISDF_REQUIRE(hoisdf_linear_emu_supported(dy, lddy, N), HOISDF_ERR_INVALID,
                 "linear_bwd_input_emu: dy must be 16-byte aligned with lddy and N multiples of 4");
  EmuArgs g{};
  // dx[m][k] = sum_n dy[m][n] W[n][k]: A = dy rows (contraction n contiguous), B = the transposed image ([k][n])
  g.A = dy; g.lda = lddy; g.Bimg = static_cast<const u32x4*>(wt_image);
  g.abits = relu_bits; g.ldbits = (N + 31) / 32; g.ascale = 1.f / (1.f - drop_p);
  g.C = dx; g.ldc = lddx; g.M = (int)M; g.N = K; g.K = N;
  g.inv_keep = 1.f;
  g.beta = accumulate ? 1 : 0;
  g.a_amax = dy_mag; g.amax_out = dx_mag;
  if (dx_heads) {
    HOISDF_REQUIRE(head_L > 0 && K % 64 == 0 && !accumulate, HOISDF_ERR_INVALID, "linear_bwd_input_emu: head magnitudes need K %% 64 == 0, the rows per sample, no accumulation");
    g.head_out = dx_heads; g.head_L = head_L; g.head_nb = cdiv(M, head_L);
  }
  return launch_emu(g, as_stream(stream));
}

namespace {
// row slices for grad-weight: one workgroup per CU (256 slots), >= 8 slabs per slice
// k-tile width: 256 (one workgroup per CU, the rotated emu_dw2_kernel) unless K <= 128 (half of a 256-wide tile would be padding).
// With the round-3 main loop (HOISDF_EMU_DW=1) the 128-wide form (two workgroups per CU) was the faster one for one or two output
// tiles; the rotated loop reversed that (round 4, tools/mb_kc2.py: 65536 x 256 x 256 109 -> 118 TF, masked 92 -> 109;
// 294912 x 256 x 256 123 -> 141 / 127 -> 150; 49152 x 512 x 256 124 -> 139).  HOISDF_EMU_DW_TILE=128 / 256 forces one form.
bool dw_old_form() {
  static int form = -1;
  if (form < 0) { const char* e = getenv("HOISDF_EMU_DW"); form = (e && atoi(e) == 1) ? 1 : 2; }
  return form == 1;
}
int dw_tile(int N, int K) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("HOISDF_EMU_DW_TILE"); const int v = e ? atoi(e) : 0; forced = (v == 128 || v == 256) ? v : 0; }
  if (forced) return forced;
  if (dw_old_form()) return cdiv(N, DT) * cdiv(K, 256) <= 2 ? 128 : 256;
  return K <= 128 ? 128 : 256;
}
void plan_dw(long M, int N, int K, int& splitk, int& mper) {
  const int dtk = dw_tile(N, K);
  const int ntile = cdiv(N, DT) * cdiv(K, dtk);
  const int slabs = cdiv(M, KS);
  const int slots = dtk == 256 ? 256 : 512;
  // slices of a tile go to the XCDs round-robin (split & 7): a whole number of slices per XCD that fits its share of the
  // slots in ONE round (768 x 256: 6 tiles x 85 slices put 66 workgroups on XCDs 0-3 with 64 slots - a second round for 2)
  const int per_xcd = slots / 8 / ntile;
  int want = per_xcd >= 1 ? per_xcd * 8 : (ntile >= slots ? 1 : slots / ntile);
  if (want > slabs / 8) want = slabs / 8 > 0 ? slabs / 8 : 1;
  mper = cdiv(slabs, want) * KS;
  splitk = cdiv(M, mper);
}
}  // namespace

extern "C" long hoisdf_linear_bwd_weight_emu_workspace(long M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int splitk, mper;
  plan_dw(M, N, K, splitk, mper);
  if (splitk <= 1) return 0;
  return (long)splitk * ((long)N * K + N);
}

namespace {
int bwd_weight_emu(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x, int ldx, float* dW, int lddw,
                   float* db, long M, int N, int K, float* workspace, long workspace_floats, bool h2, const uint32_t* dy_mag,
                   const uint32_t* x_mag, void* stream);
}
extern "C" int hoisdf_linear_bwd_weight_emu(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x,
                                            int ldx, float* dW, int lddw, float* db, long M, int N, int K, float* workspace,
                                            long workspace_floats, void* stream) {
  return bwd_weight_emu(dy, lddy, relu_bits, drop_p, x, ldx, dW, lddw, db, M, N, K, workspace, workspace_floats, false, nullptr, nullptr, stream);
}
// the f16x2 form (when the process runs it, hoisdf_linear_emu_pieces() == 2, and the tile is the 256-wide one; otherwise as above):
// dy_mag / x_mag = magnitude words of the two operands, NULL = measured here
extern "C" int hoisdf_linear_bwd_weight_emu_mag(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x,
                                                int ldx, float* dW, int lddw, float* db, long M, int N, int K, float* workspace,
                                                long workspace_floats, const uint32_t* dy_mag, const uint32_t* x_mag, void* stream) {
  return bwd_weight_emu(dy, lddy, relu_bits, drop_p, x, ldx, dW, lddw, db, M, N, K, workspace, workspace_floats, form_h2(), dy_mag, x_mag, stream);
}
int hoisdf::linear_bwd_weight_emu_mag(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x, int ldx,
                                      float* dW, int lddw, float* db, long M, int N, int K, float* workspace, long workspace_floats,
                                      const uint32_t* dy_mag, const uint32_t* x_mag, void* stream) {
  return bwd_weight_emu(dy, lddy, relu_bits, drop_p, x, ldx, dW, lddw, db, M, N, K, workspace, workspace_floats, form_h2(), dy_mag, x_mag, stream);
}

namespace {
int bwd_weight_emu(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x, int ldx, float* dW, int lddw,
                   float* db, long M, int N, int K, float* workspace, long workspace_floats, bool h2, const uint32_t* dy_mag,
                   const uint32_t* x_mag, void* stream) {
  HOISDF_REQUIRE(dW && (M == 0 || (dy && x)), HOISDF_ERR_INVALID, "linear_bwd_weight_emu: null pointer");
  HOISDF_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw == K && M < (1L << 31) && drop_p >= 0.f && drop_p < 1.f,
                 HOISDF_ERR_INVALID, "linear_bwd_weight_emu: bad sizes (a dense dW, lddw == K, is required)");
  HOISDF_REQUIRE(al16(dy) && al16(x) && al16(dW) && (lddy % 4 == 0) && (ldx % 4 == 0) && (N % 4 == 0) && (K % 4 == 0),
                 HOISDF_ERR_INVALID, "linear_bwd_weight_emu: operands must be 16-byte aligned with N, K and leading dims multiples of 4");
  hipStream_t st = as_stream(stream);
  static bool attr_set = false;
  if (!attr_set) {
    bool ok = true;
#define DW_ATTR(M_, T_) ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(emu_dw_kernel<M_, T_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2u * (3 * 2 * DT + 3 * 2 * T_) * 16u)) == hipSuccess
    DW_ATTR(false, 256); DW_ATTR(true, 256); DW_ATTR(false, 128); DW_ATTR(true, 128);
#undef DW_ATTR
    if (!ok) {
      set_error("linear_bwd_weight_emu: cannot raise the dynamic LDS limit");
      return HOISDF_ERR_LAUNCH;
    }
    attr_set = true;
  }
  DwArgs g{};
  g.dy = dy; g.lddy = lddy; g.x = x; g.ldx = ldx;
  g.bits = relu_bits; g.ldbits = (N + 31) / 32; g.ascale = 1.f / (1.f - drop_p);
  g.M = (int)M; g.N = N; g.K = K;
  const int dtk = dw_tile(N, K);
  g.tiles_n = cdiv(N, DT); g.tiles_k = cdiv(K, dtk);
  plan_dw(M, N, K, g.splitk, g.m_per_split);
  const long need = g.splitk > 1 ? (long)g.splitk * ((long)N * K + N) : 0;
  HOISDF_REQUIRE(need == 0 || (workspace && workspace_floats >= need && al16(workspace)), HOISDF_ERR_WORKSPACE,
                 "linear_bwd_weight_emu: workspace of %ld floats needed", need);
  if (g.splitk > 1) {
    g.C = workspace; g.c_split_stride = (long)N * K;
    g.colsum = db ? workspace + (size_t)g.splitk * N * K : nullptr; g.colsum_split_stride = N;
  } else {
    g.C = dW; g.c_split_stride = 0; g.colsum = db; g.colsum_split_stride = 0;
  }
  const int ntile = g.tiles_n * g.tiles_k;
  const dim3 grid((unsigned)(ntile * 8 * cdiv(g.splitk, 8))), block(NT);
  const unsigned lb = 2u * (3 * 2 * DT + 3 * 2 * dtk) * 16u;
  const int form = dw_old_form() ? 1 : 2;     // HOISDF_EMU_DW=1: the first main-loop form for the 256-wide tiles (A/B runs)
  if (dtk == 256 && form == 2 && h2) {
    g.dy_amax = dy_mag; g.x_amax = x_mag;
    if (!dy_mag) {
      uint32_t* part = mag_scratch(st, M);
      if (!part) { set_error("linear_bwd_weight_emu: cannot allocate the row magnitudes"); return HOISDF_ERR_LAUNCH; }
      mag_trace("grad-weight dy", M, N);
      if (int rc = emu_rowmag_launch(dy, lddy, M, N, part, st)) return rc;
      g.dy_amax = part;
    }
    if (!x_mag) {
      uint32_t* part = mag_scratch(st, M);
      if (!part) { set_error("linear_bwd_weight_emu: cannot allocate the row magnitudes"); return HOISDF_ERR_LAUNCH; }
      mag_trace("grad-weight x", M, K);
      if (int rc = emu_rowmag_launch(x, ldx, M, K, part, st)) return rc;
      g.x_amax = part;
    }
    const bool hasdb = g.colsum != nullptr;
    if (relu_bits && hasdb) hipLaunchKernelGGL((emu_dw2h_kernel<true, true>), grid, block, 0, st, g);
    else if (relu_bits) hipLaunchKernelGGL((emu_dw2h_kernel<true, false>), grid, block, 0, st, g);
    else if (hasdb) hipLaunchKernelGGL((emu_dw2h_kernel<false, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((emu_dw2h_kernel<false, false>), grid, block, 0, st, g);
  } else if (dtk == 256 && form == 2) {
    const bool hasdb = g.colsum != nullptr;
    if (relu_bits && hasdb) hipLaunchKernelGGL((emu_dw2_kernel<true, true>), grid, block, 0, st, g);
    else if (relu_bits) hipLaunchKernelGGL((emu_dw2_kernel<true, false>), grid, block, 0, st, g);
    else if (hasdb) hipLaunchKernelGGL((emu_dw2_kernel<false, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((emu_dw2_kernel<false, false>), grid, block, 0, st, g);
  } else if (dtk == 256) {
    if (relu_bits) hipLaunchKernelGGL((emu_dw_kernel<true, 256>), grid, block, lb, st, g);
    else hipLaunchKernelGGL((emu_dw_kernel<false, 256>), grid, block, lb, st, g);
  } else {
    if (relu_bits) hipLaunchKernelGGL((emu_dw_kernel<true, 128>), grid, block, lb, st, g);
    else hipLaunchKernelGGL((emu_dw_kernel<false, 128>), grid, block, lb, st, g);
  }
  if (int rc = check_launch("linear_bwd_weight_emu")) return rc;
  if (g.splitk > 1) {
    const long n = (long)N * K;
    const int b0 = (int)((n + 255) / 256), b1 = db ? (N + 255) / 256 : 0;
    hipLaunchKernelGGL(emu_reduce_partials_kernel, dim3((unsigned)(b0 + b1)), dim3(1024), 0, st, workspace, n, g.splitk, dW, n, b0,
                       workspace + (size_t)g.splitk * N * K, (long)N, db, (long)N);
    return check_launch("linear_bwd_weight_emu reduce");
  }
  return HOISDF_OK;
}
}  // namespace
