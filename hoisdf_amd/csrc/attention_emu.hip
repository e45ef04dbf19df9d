// fp32 attention EMULATED on the bf16 MFMA pipe ("bf16x3"), forward + backward for training and evaluation.
// reference: common/nets/transformer.py:269,286-302 (nn.MultiheadAttention inside the encoder layers), forward + autograd
// backward.  Every f32 operand of the five contractions (Q, K, V, dO and the in-kernel P, dS) is split EXACTLY into three bf16
// pieces (8 + 8 + 8 significand bits, bf16 has the f32 exponent range: no scaling) and every product is accumulated in f32
// from six v_mfma_f32_32x32x16_bf16 products (the three dropped cross terms are <= 2^-24 of the product) - the arithmetic of
// gemm_emu.hip; softmax state, dropout and the dS algebra stay f32.  Results carry the error of the exact-f32 kernels of
// attention.hip (tests/test_gpu_emu.py), the contractions run on a pipe that is 16 x faster.
//
//   pre-pass  : f32 head slices -> bf16 planes, row-major [bh][Lp][64] and (V only) transposed [bh][64][Lp] (Q pre-scaled by
//               log2(e)/8: softmax in the log2 domain, the LSE convention of attention.hip)
//   forward   : block = 128 queries (lane = query), streams 32-key tiles of K rows / V^T:  S^T = K.Q^T, O^T += V^T.P^T
//   backward  : ONE pass, 5 GEMM-equivalents: block = 128 keys in 8 waves of 16 (lane = key: K, V fragments and dK, dV
//               accumulators in registers), streams 32-query tiles of Q / dO rows (the transposed fragments come out of the
//               same tiles through ds_read_b64_tr_b16):  S = Q.K^T, dP = dO.V^T,
//               dV^T += dO^T.Pd, dK^T += Q^T.dS; every wave drops its dS block (bf16 triples) into a shared LDS tile
//               T[32 q][128 keys], and each wave then contracts T with the block's K rows over all 128 keys for one 16 x 16
//               tile of the block's 32 x 64 dQ contribution.  It goes to a per-key-block partial buffer [kb][bh][q][64]; a
//               reduce pass sums the key blocks in order: no atomics anywhere, run-to-run identical.
// The dropout mask is the same function of (seed, query, key) as in the f32 kernels.
#include <stdlib.h>

#include "attention_emu.h"

namespace hoisdf {
using emu_attn::EmuAttn;
using emu_attn::emu_block;

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int D = 64;
constexpr int RP = 72;              // bf16 per row of a row-major [32][64] tile in LDS (144 B)
constexpr int TPH = 36;             // bf16 per row of a transposed [64][32] tile in LDS (72 B: conflict-free 8-byte reads)
constexpr int ROWS_T = 32 * RP;     // bf16 per row-major plane tile
constexpr int TRN_T = 64 * TPH;     // bf16 per transposed plane tile
constexpr float QS2 = 0.125f * 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
#define CR(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))
#define MB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// x y accumulated from six products (small terms first): x2y0 + x0y2 + x1y1 + x1y0 + x0y1 + x0y0
#define MB6(acc, x0, x1, x2, y0, y1, y2) \
  do {                                   \
    acc = MB(x2, y0, acc);               \
    acc = MB(x0, y2, acc);               \
    acc = MB(x1, y1, acc);               \
    acc = MB(x1, y0, acc);               \
    acc = MB(x0, y1, acc);               \
    acc = MB(x0, y0, acc);               \
  } while (0)

#define SPLIT1(x, i)                             \
  do {                                           \
    const __bf16 a_ = (__bf16)(x);               \
    const float r1_ = (x) - (float)a_;           \
    const __bf16 b_ = (__bf16)r1_;               \
    const float r2_ = r1_ - (float)b_;           \
    p0[i] = a_; p1[i] = b_; p2[i] = (__bf16)r2_; \
  } while (0)

// tile copies global -> registers -> LDS.  Row-major plane tile: 32 rows x 64 bf16, thread -> (row t >> 3, 8 values at
// (t & 7) * 8).  Transposed plane tile: 64 rows (d) x 32 bf16, thread -> (row t >> 2, 8 values at (t & 3) * 8), stored as two
// 8-byte halves (rows are 72 B apart).
__device__ __forceinline__ u32x4 ld_rows(const __bf16* base, size_t row0, int tid) {
  return *reinterpret_cast<const u32x4*>(base + (row0 + (tid >> 3)) * D + (tid & 7) * 8);
}
__device__ __forceinline__ void st_rows(__bf16* lds, const u32x4 v, int tid) {
  *reinterpret_cast<u32x4*>(lds + (tid >> 3) * RP + (tid & 7) * 8) = v;
}
__device__ __forceinline__ u32x4 ld_trn(const __bf16* base, size_t Lp, size_t col0, int tid) {
  return *reinterpret_cast<const u32x4*>(base + (size_t)(tid >> 2) * Lp + col0 + (tid & 3) * 8);
}
__device__ __forceinline__ void st_trn(__bf16* lds, const u32x4 v, int tid) {
  u32x2* p = reinterpret_cast<u32x2*>(lds + (tid >> 2) * TPH + (tid & 3) * 8);
  p[0] = u32x2{v.x, v.y};
  p[1] = u32x2{v.z, v.w};
}
// fragment of a transposed tile for k-step jj: the 8 "k" entries held by accumulator registers 8 jj .. 8 jj + 7 of a lane
// (rows CR(r, h)): two 8-byte reads
__device__ __forceinline__ bf16x8 frag_trn(const __bf16* tile, int row, int jj, int h) {
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(tile + row * TPH + 16 * jj + 4 * h);
  const bf16x4 b = *reinterpret_cast<const bf16x4*>(tile + row * TPH + 16 * jj + 8 + 4 * h);
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ void split8(const f32x16& s, int jj, float mul, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = s[8 * jj + i] * mul;
    SPLIT1(x, i);
  }
}
}  // namespace

// ---- conversion pre-pass: f32 [B][L][ld] (head slice) -> three bf16 planes, row-major [bh][Lp][64] and / or transposed
// [bh][64][Lp]; rows >= L are zero.  One block per (bh, 64-row tile): a thread owns 16 consecutive d of one row (4 threads per
// row: 256-byte coalesced reads, 128-byte coalesced row-major writes); the transposed copies go through an LDS tile.
// F16 (the f16x2 form of the attention, round 5): TWO planes of f16 bits - hi = f16(x scale s), lo = f16(x scale s - hi) with s the
// power of two that the head magnitude of the block's (sample, head) gives (round 6; common.h: mag[head * B + b] - a sample's
// planes do not depend on the other samples of the batch); the third plane pointers are not touched.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <bool F16>
__global__ __launch_bounds__(256) void emu_attn_convert_kernel(const float* __restrict__ src, int ld, int L, int Lp, int B, int H,
                                                               float scale, __bf16* __restrict__ r0, __bf16* __restrict__ r1,
                                                               __bf16* __restrict__ r2, __bf16* __restrict__ t0,
                                                               __bf16* __restrict__ t1, __bf16* __restrict__ t2,
                                                               const uint32_t* __restrict__ mag) {
  constexpr int TP = 72;
  __shared__ __attribute__((aligned(16))) __bf16 tile[3][64 * TP];
  const int tid = threadIdx.x;
  const int nb = Lp / 64;
  const int kb = blockIdx.x % nb, bh = blockIdx.x / nb, b = bh / H, head = bh - b * H;
  if (F16) { f16_saturate_on(); scale *= mag_scale(mag[head * B + b]); }
  const int r = tid >> 2, dc = (tid & 3) * 16;
  const int row = kb * 64 + r;
  const bool valid = row < L;
  const float* sr = src + ((size_t)b * L + (valid ? row : 0)) * ld + head * D + dc;
  bf16x8 pa[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float4 u = valid ? *reinterpret_cast<const float4*>(sr + 8 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w = valid ? *reinterpret_cast<const float4*>(sr + 8 * i + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    u.x *= scale; u.y *= scale; u.z *= scale; u.w *= scale;
    w.x *= scale; w.y *= scale; w.z *= scale; w.w *= scale;
    if (F16) {
      const float e_[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
      f16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)e_[j]; lo[j] = (_Float16)(e_[j] - (float)hi[j]); }
      pa[i][0] = __builtin_bit_cast(bf16x8, hi); pa[i][1] = __builtin_bit_cast(bf16x8, lo); pa[i][2] = pa[i][1];
    } else {
      bf16x8 p0, p1, p2;
      SPLIT1(u.x, 0); SPLIT1(u.y, 1); SPLIT1(u.z, 2); SPLIT1(u.w, 3);
      SPLIT1(w.x, 4); SPLIT1(w.y, 5); SPLIT1(w.z, 6); SPLIT1(w.w, 7);
      pa[i][0] = p0; pa[i][1] = p1; pa[i][2] = p2;
    }
  }
  if (r0) {
    const size_t o = ((size_t)bh * Lp + row) * D + dc;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<bf16x8*>(r0 + o + 8 * i) = pa[i][0];
      *reinterpret_cast<bf16x8*>(r1 + o + 8 * i) = pa[i][1];
      if (!F16 && r2) *reinterpret_cast<bf16x8*>(r2 + o + 8 * i) = pa[i][2];  // (two-plane callers pass null third planes)
    }
  }
  if (t0) {                                                // block-uniform
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        tile[0][(dc + 8 * i + e) * TP + r] = pa[i][0][e];
        tile[1][(dc + 8 * i + e) * TP + r] = pa[i][1][e];
        if (!F16) tile[2][(dc + 8 * i + e) * TP + r] = pa[i][2][e];
      }
    __syncthreads();
    const int d = tid >> 2, rc = (tid & 3) * 16;           // 16 consecutive rows of d-row d
    const size_t o = ((size_t)bh * D + d) * Lp + kb * 64 + rc;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(t0 + o + 8 * i) = *reinterpret_cast<const u32x4*>(&tile[0][d * TP + rc + 8 * i]);
      *reinterpret_cast<u32x4*>(t1 + o + 8 * i) = *reinterpret_cast<const u32x4*>(&tile[1][d * TP + rc + 8 * i]);
      if (!F16 && t2) *reinterpret_cast<u32x4*>(t2 + o + 8 * i) = *reinterpret_cast<const u32x4*>(&tile[2][d * TP + rc + 8 * i]);
    }
  }
}

// ============================================================================================================================
// forward: block = 128 queries (lane = query), streams 32-key tiles of K rows (3 planes) and V^T (3 planes)
// ============================================================================================================================
__global__ __launch_bounds__(256, 2) void emu_attn_fwd_kernel(EmuAttn a) {
  constexpr int BUF = 3 * ROWS_T + 3 * TRN_T;
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  int qtile, bh;
  if (!emu_block((a.Lq + 127) / 128, a.B * a.H, qtile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int qrow = qtile * 128 + wave * 32 + c;
  const __bf16* kp[3] = {a.k[0] + (size_t)bh * a.Lkp * D, a.k[1] + (size_t)bh * a.Lkp * D, a.k[2] + (size_t)bh * a.Lkp * D};
  const __bf16* vp[3] = {a.vt[0] + (size_t)bh * D * a.Lkp, a.vt[1] + (size_t)bh * D * a.Lkp, a.vt[2] + (size_t)bh * D * a.Lkp};

  bf16x8 qf[4][3];                        // Q^T fragments: k-step j <-> d = 16 j + 8 h .. + 7 of the lane's query
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const __bf16* s = a.q[p] + ((size_t)bh * a.Lqp + qrow) * D;       // qrow < Lqp always (padded with zero rows)
#pragma unroll
    for (int j = 0; j < 4; ++j) qf[j][p] = *reinterpret_cast<const bf16x8*>(s + 16 * j + 8 * h);
  }
  const uint32_t rowkey = drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qrow));
  float m = -INFINITY, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

  const int ntiles = (a.kv_len + 31) / 32;
  u32x4 r0, r1, r2, r3, r4, r5;
#define FWD_LOAD(kt)                                   \
  do {                                                 \
    r0 = ld_rows(kp[0], (size_t)(kt) * 32, tid);       \
    r1 = ld_rows(kp[1], (size_t)(kt) * 32, tid);       \
    r2 = ld_rows(kp[2], (size_t)(kt) * 32, tid);       \
    r3 = ld_trn(vp[0], a.Lkp, (size_t)(kt) * 32, tid); \
    r4 = ld_trn(vp[1], a.Lkp, (size_t)(kt) * 32, tid); \
    r5 = ld_trn(vp[2], a.Lkp, (size_t)(kt) * 32, tid); \
  } while (0)
#define FWD_STORE(buf)                                 \
  do {                                                 \
    st_rows((buf), r0, tid);                           \
    st_rows((buf) + ROWS_T, r1, tid);                  \
    st_rows((buf) + 2 * ROWS_T, r2, tid);              \
    st_trn((buf) + 3 * ROWS_T, r3, tid);               \
    st_trn((buf) + 3 * ROWS_T + TRN_T, r4, tid);       \
    st_trn((buf) + 3 * ROWS_T + 2 * TRN_T, r5, tid);   \
  } while (0)
  FWD_LOAD(0);
  FWD_STORE(lds);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    FWD_LOAD(min(kt + 1, ntiles - 1));              // unconditional (past the end the last tile is re-read and dropped)
    const __bf16* K0 = lds + cur * BUF;
    const __bf16* V0 = K0 + 3 * ROWS_T;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(&K0[c * RP + 16 * j + 8 * h]);
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(&K0[ROWS_T + c * RP + 16 * j + 8 * h]);
      const bf16x8 k2 = *reinterpret_cast<const bf16x8*>(&K0[2 * ROWS_T + c * RP + 16 * j + 8 * h]);
      MB6(s, k0, k1, k2, qf[j][0], qf[j][1], qf[j][2]);
    }
    if (kt == ntiles - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * 32 + CR(r, h) >= a.kv_len) s[r] = -INFINITY;
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    // lazy rescale: the running reference m only moves when some query of the wave sees a score more than 2^8 above it (always
    // on the first tile, almost never afterwards) - exp2(s - m) <= 256 stays exact through the three-way split, O / l and the LSE
    // m + log2(l) are unchanged by the choice of reference, and the 32 multiplies of the accumulator are skipped
    if (__any(mt > m + 8.f)) {
      const float mn = fmaxf(m, mt);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      lsum *= alpha;
      m = mn;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    }
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] - m);
      ps += p;
      s[r] = p;
    }
    lsum += ps;
    if (a.drop_p > 0.f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= drop_scale(rowkey, (uint32_t)(kt * 32 + CR(r, h)), a.thresh, a.inv_keep);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      bf16x8 p0, p1, p2;
      split8(s, jj, 1.f, p0, p1, p2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8 v0 = frag_trn(V0, dt * 32 + c, jj, h);
        const bf16x8 v1 = frag_trn(V0 + TRN_T, dt * 32 + c, jj, h);
        const bf16x8 v2 = frag_trn(V0 + 2 * TRN_T, dt * 32 + c, jj, h);
        MB6(o[dt], v0, v1, v2, p0, p1, p2);
      }
    }
    if (kt + 1 < ntiles) FWD_STORE(lds + (cur ^ 1) * BUF);
    __syncthreads();
  }
#undef FWD_LOAD
#undef FWD_STORE
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  uint32_t omax = 0u;
  if (qrow < a.Lq) {
    const float inv = 1.f / ltot;
    float* op = a.out + ((size_t)b * a.Lq + qrow) * a.ldo + head * D;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ov = make_float4(o[t][4 * g + 0] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + 32 * t + 8 * g + 4 * h) = ov;
        omax = max(omax, mag_bits4(ov));
      }
    if (h == 0 && a.lse) a.lse[(size_t)bh * a.Lq + qrow] = m + log2f(ltot);       // log2 domain
  }
  if (a.mag) {                               // row magnitudes of o (common.h): lanes c, c + 32 hold the two halves of the row's head slice
    omax = max(omax, (uint32_t)__shfl_xor((int)omax, 32, 64));
    if (h == 0 && qrow < a.Lq) atomicMax(a.mag + (size_t)b * a.Lq + qrow, omax);
  }
}


// ============================================================================================================================
// forward, second form (round 4): the same tiles, fragments, product order and online softmax as emu_attn_fwd_kernel (with
// dropout off the outputs are bit identical), software-pipelined over the key tiles and hand-interleaved:
//     S phase : the 24 MFMAs of S(t + 1) = K(t + 1) . Q^T   with the softmax / dropout / three-way split of tile t behind them
//     PV phase: the 24 MFMAs of O^T += V(t)^T . P(t)^T      with the running maximum of tile t + 1, its dropout decisions (one hash
//               per two keys), the staging writes of K(t + 2) / V(t + 1) and the loads of K(t + 3) / V(t + 2) behind them
// In the first form a wave runs [24 MFMAs | ~250 VALU | 24 MFMAs] per key tile and the MFMA pipe is 40 % busy (PMC, round 4): the
// VALU work of a tile is about as long as its MFMAs, and on this part it only hides under the SAME wave's MFMAs.  Here every unit
// of it is pinned behind one MFMA of the other tile (tools/gen/attn_fwd2_phase.py).  K rows and V^T tiles are double-buffered
// separately (K(t + 1) and V(t) are read while K(t + 2) and V(t + 1) are written): four __shared__ objects of 13.5 KB, one barrier
// per tile.
// ============================================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// NPL = planes per operand: 3 = fp32-equivalent (six products per product); 2 = bf16 hi + lo operands, three products - the 16-bit-operand
// evaluation kernel of BASELINE configs[4] (hoisdf_attention_fwd_bf16x2; ~2^-17 relative operand error, f32 softmax and accumulation)
// F16 (with NPL = 2): the f16x2 form - the planes hold f16 hi + lo pieces of Q sQ, K sK, V sV (the powers of two from the head
// magnitudes of the block's (sample, head): a.q_hm / k_hm / v_hm; emu_attn_convert_kernel<true>), the products run on
// v_mfma_f32_32x32x16_f16.  The scores come out multiplied by sQ sK: the softmax works on them as they are (running maximum, rescale threshold and exponent argument carry
// the factor 1 / s^2 in one fused multiply-add); P is formed as 2^6 P (<= 2^14 at the lazy rescale's 2^8 head room) so that its f16
// pieces keep 22 bits down to 2^-22 of the row's largest probability; the row sum carries the same 2^6 and O = acc / (l s).
template <bool DROP, int NPL, bool F16>
__global__ __launch_bounds__(256, 2) void emu_attn_fwd2_kernel(EmuAttn a) {
  static_assert(!F16 || NPL == 2, "the f16 form has two planes");
  __shared__ __attribute__((aligned(16))) __bf16 Kb0[3 * ROWS_T];
  __shared__ __attribute__((aligned(16))) __bf16 Kb1[3 * ROWS_T];
  __shared__ __attribute__((aligned(16))) __bf16 Vb0[3 * TRN_T];
  __shared__ __attribute__((aligned(16))) __bf16 Vb1[3 * TRN_T];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  int qtile, bh;
  if (!emu_block((a.Lq + 127) / 128, a.B * a.H, qtile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int qrow = qtile * 128 + wave * 32 + c;
  const __bf16* kp[3] = {a.k[0] + (size_t)bh * a.Lkp * D, a.k[1] + (size_t)bh * a.Lkp * D, a.k[NPL == 3 ? 2 : 1] + (size_t)bh * a.Lkp * D};
  const __bf16* vp[3] = {a.vt[0] + (size_t)bh * D * a.Lkp, a.vt[1] + (size_t)bh * D * a.Lkp, a.vt[NPL == 3 ? 2 : 1] + (size_t)bh * D * a.Lkp};
  // f16 form: cs = 1 / s^2 takes the accumulated scores to the log2-domain scores, vinv = 1 / s takes the accumulated output back
  float cs = 1.f, vinv = 1.f;
  constexpr float PB = F16 ? 6.f : 0.f;                     // P is formed as 2^PB P
  if (F16) {                                                // the (sample, head)'s own scales (common.h head magnitudes)
    f16_saturate_on();
    const float iq_ = mag_inv_scale(a.q_hm[head * a.B + b]), ik_ = mag_inv_scale(a.k_hm[head * a.B + b]);
    cs = fmaxf(iq_ * ik_, 0x1p-100f); vinv = mag_inv_scale(a.v_hm[head * a.B + b]);      // (floored: operands below ~2^-37 have scores of exactly zero either way)
  }
  const float thr8 = F16 ? 8.f / cs : 8.f;                  // the lazy rescale's threshold in accumulator units
  float mneg = INFINITY;                                    // f16 form: PB - m cs (the exponent argument is fma(score, cs, mneg))

  bf16x8 qf[4][3];                        // Q^T fragments: k-step j <-> d = 16 j + 8 h .. + 7 of the lane's query
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    const __bf16* sq = a.q[p] + ((size_t)bh * a.Lqp + qrow) * D;
#pragma unroll
    for (int j = 0; j < 4; ++j) qf[j][p] = *reinterpret_cast<const bf16x8*>(sq + 16 * j + 8 * h);
  }
  const uint32_t rowkey = drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qrow));
  const uint32_t dthr = a.thresh & 0xffff0000u;
  float m = -INFINITY, lsum = 0.f, ps = 0.f;
  f32x16 o[2], s_cur, s_nxt;
  float dsc[16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) dsc[r] = 1.f;
  const int ntiles = (a.kv_len + 31) / 32;
  const int lastt = ntiles - 1;
  u32x4 rk[3], rv[3];
  bf16x8 kf[2][3], vf[2][3], pw[2][3];
  uint32_t w0[8], w1[8], w2[8], hx[8];     // the three planes of P(t) (two scores per word), the dropout hashes of tile t + 1
  f32x2 e_[8], f_[8];
  float mt = -INFINITY;
#define KFRAG(KB, p, j) (*reinterpret_cast<const bf16x8*>(&(KB)[(p) * ROWS_T + c * RP + 16 * (j) + 8 * h]))
#define VFRAG(VB, p, dt, jj) frag_trn((VB) + (p) * TRN_T, (dt) * 32 + c, jj, h)
#define SB() __builtin_amdgcn_sched_barrier(0)
#define PK_SUB(d, x, y) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y))
// ---- units of the S phase: scores 2 pr, 2 pr + 1 of tile t -> probabilities -> (dropout) -> three bf16 planes
#define PE1(pr)                                                                                                        \
  do {                                                                                                                 \
    if constexpr (F16) e_[pr] = f32x2{__builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[2 * (pr)], cs, mneg)),               \
                                     __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[2 * (pr) + 1], cs, mneg))};           \
    else e_[pr] = f32x2{__builtin_amdgcn_exp2f(s_cur[2 * (pr)] - m), __builtin_amdgcn_exp2f(s_cur[2 * (pr) + 1] - m)}; \
  } while (0)
#define PE2(pr)                                                                                                        \
  do {                                                                                                                 \
    ps += e_[pr].x + e_[pr].y;                                                                                         \
    if (DROP) e_[pr] = f32x2{e_[pr].x * dsc[2 * (pr)], e_[pr].y * dsc[2 * (pr) + 1]};                                   \
    if constexpr (F16) {                                                                                               \
      const f16x2 hh_ = __builtin_convertvector(e_[pr], f16x2);                                                        \
      w0[pr] = __builtin_bit_cast(uint32_t, hh_);                                                                      \
      f_[pr] = __builtin_convertvector(hh_, f32x2);                                                                    \
    } else {                                                                                                           \
      const uint32_t h_ = __builtin_bit_cast(uint32_t, __builtin_convertvector(e_[pr], bf16x2));                       \
      w0[pr] = h_;                                                                                                     \
      f_[pr] = f32x2{__builtin_bit_cast(float, h_ << 16), __builtin_bit_cast(float, h_ & 0xffff0000u)};                \
    }                                                                                                                  \
  } while (0)
#define PE3(pr)                                                                                                        \
  do {                                                                                                                 \
    PK_SUB(e_[pr], e_[pr], f_[pr]);                                                                                    \
    const uint32_t h_ = __builtin_bit_cast(uint32_t, __builtin_convertvector(e_[pr], bf16x2));                         \
    w1[pr] = h_;                                                                                                       \
    f_[pr] = f32x2{__builtin_bit_cast(float, h_ << 16), __builtin_bit_cast(float, h_ & 0xffff0000u)};                  \
  } while (0)
#define PE4(pr)                                                                                                        \
  do {                                                                                                                 \
    f32x2 r_; PK_SUB(r_, e_[pr], f_[pr]);                                                                              \
    w2[pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r_, bf16x2));                                        \
    if (((pr) & 3) == 3) {                      /* a key half-tile is complete: its three B fragments */                \
      pw[(pr) >> 2][0] = __builtin_bit_cast(bf16x8, u32x4{w0[(pr) - 3], w0[(pr) - 2], w0[(pr) - 1], w0[pr]});          \
      pw[(pr) >> 2][1] = __builtin_bit_cast(bf16x8, u32x4{w1[(pr) - 3], w1[(pr) - 2], w1[(pr) - 1], w1[pr]});          \
      pw[(pr) >> 2][2] = __builtin_bit_cast(bf16x8, u32x4{w2[(pr) - 3], w2[(pr) - 2], w2[(pr) - 1], w2[pr]});          \
    }                                                                                                                  \
  } while (0)
// (two planes: the second piece closes the split)
#define PE3B(pr)                                                                                                       \
  do {                                                                                                                 \
    f32x2 r_; PK_SUB(r_, e_[pr], f_[pr]);                                                                              \
    if constexpr (F16) w1[pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r_, f16x2));                      \
    else w1[pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r_, bf16x2));                                   \
    if (((pr) & 3) == 3) {                                                                                             \
      pw[(pr) >> 2][0] = __builtin_bit_cast(bf16x8, u32x4{w0[(pr) - 3], w0[(pr) - 2], w0[(pr) - 1], w0[pr]});          \
      pw[(pr) >> 2][1] = __builtin_bit_cast(bf16x8, u32x4{w1[(pr) - 3], w1[(pr) - 2], w1[(pr) - 1], w1[pr]});          \
    }                                                                                                                  \
  } while (0)
// ---- units of the PV phase: running maximum of tile t + 1 (four scores each), dropout decisions of tile t + 1 (pair pr = keys
// CR(2 pr, h), CR(2 pr, h) + 1: one hash, the low half decides the even key), staging
// (the empty asm statements pin a unit's result HERE: without them the optimiser sinks the maximum into the rarely taken statistics
// branch and the hashes into the next iteration, i.e. out from under the MFMAs)
// (v_max3_f32 through asm: fmaxf() costs three instructions per score here - hipcc canonicalises both operands of every v_max_f32 -
// 62 per tile against 8; the scores are never NaN.  hipcc does not pad hazards for asm operands: the units sit >= 4 MFMAs = 128
// cycles behind the last MFMA that wrote s_nxt, far beyond the 11 wait states an 8-pass result needs)
#define PM(i)                                                                                                          \
  do {                                                                                                                 \
    float t_;                                                                                                          \
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(t_) : "v"(s_nxt[4 * (i)]), "v"(s_nxt[4 * (i) + 1]), "v"(s_nxt[4 * (i) + 2])); \
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mt) : "v"(t_), "v"(s_nxt[4 * (i) + 3]));                           \
  } while (0)
#define PH1(pr) do { if (DROP) { uint32_t x_ = hbase + (uint32_t)(CR(2 * (pr), 0) >> 1) * 0x9E3779B9U; x_ ^= x_ >> 15; hx[pr] = x_ * 0x2C1B3C6DU; asm volatile("" : "+v"(hx[pr])); } } while (0)
#define PH2(pr) do { if (DROP) { hx[pr] ^= hx[pr] >> 12; asm volatile("" : "+v"(hx[pr])); } } while (0)
#define PH3(pr) do { if (DROP) { dsc[2 * (pr)] = (hx[pr] << 16) >= dthr ? a.inv_keep : 0.f; dsc[2 * (pr) + 1] = hx[pr] >= dthr ? a.inv_keep : 0.f; asm volatile("" : "+v"(dsc[2 * (pr)]), "+v"(dsc[2 * (pr) + 1])); } } while (0)
#define STK(p) st_rows(kw + (p) * ROWS_T, rk[p], tid)
#define STV(p) st_trn(vw + (p) * TRN_T, rv[p], tid)
#define LDK(p) rk[p] = ld_rows(kp[p], (size_t)ktn * 32, tid)
#define LDV(p) rv[p] = ld_trn(vp[p], a.Lkp, (size_t)vtn * 32, tid)
#ifndef FWD2_ABL
#define FWD2_ABL 0
#endif
#if FWD2_ABL & 1            /* ablation (timing only, wrong results): no softmax / split / hash arithmetic */
#undef PE1
#undef PE2
#undef PE3
#undef PE4
#undef PE3B
#undef PM
#undef PH1
#undef PH2
#undef PH3
#define PE1(pr) ((void)0)
#define PE2(pr) ((void)0)
#define PE3(pr) ((void)0)
#define PE4(pr) ((void)0)
#define PE3B(pr) ((void)0)
#define PM(i) ((void)0)
#define PH1(pr) ((void)0)
#define PH2(pr) ((void)0)
#define PH3(pr) ((void)0)
#endif
#if FWD2_ABL & 8            /* no exp / subtract */
#undef PE1
#define PE1(pr) do { e_[pr] = f32x2{s_cur[2 * (pr)], s_cur[2 * (pr) + 1]}; } while (0)
#endif
#if FWD2_ABL & 16           /* no split arithmetic (planes = raw bits) */
#undef PE2
#undef PE3
#undef PE4
#define PE2(pr) do { ps += e_[pr].x + e_[pr].y; w0[pr] = __builtin_bit_cast(uint32_t, e_[pr].x); } while (0)
#define PE3(pr) do { w1[pr] = __builtin_bit_cast(uint32_t, e_[pr].y); } while (0)
#define PE4(pr)                                                                                                        \
  do {                                                                                                                 \
    w2[pr] = w0[pr] ^ w1[pr];                                                                                          \
    if (((pr) & 3) == 3) {                                                                                             \
      pw[(pr) >> 2][0] = __builtin_bit_cast(bf16x8, u32x4{w0[(pr) - 3], w0[(pr) - 2], w0[(pr) - 1], w0[pr]});          \
      pw[(pr) >> 2][1] = __builtin_bit_cast(bf16x8, u32x4{w1[(pr) - 3], w1[(pr) - 2], w1[(pr) - 1], w1[pr]});          \
      pw[(pr) >> 2][2] = __builtin_bit_cast(bf16x8, u32x4{w2[(pr) - 3], w2[(pr) - 2], w2[(pr) - 1], w2[pr]});          \
    }                                                                                                                  \
  } while (0)
#endif
#if FWD2_ABL & 32           /* no running maximum */
#undef PM
#define PM(i) ((void)0)
#endif
#if FWD2_ABL & 2            /* no staging: the tiles of the prologue are re-read */
#undef STK
#undef STV
#undef LDK
#undef LDV
#define STK(p) ((void)0)
#define STV(p) ((void)0)
#define LDK(p) ((void)0)
#define LDV(p) ((void)0)
#endif
#if FWD2_ABL & 4            /* no barrier */
#define FWD2_SYNC() ((void)0)
#else
#define FWD2_SYNC() __syncthreads()
#endif
#pragma push_macro("MB")
#undef MB
#define MB(a_, b_, c_)                                                                                                             \
  (F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a_)), __builtin_bit_cast(f16x8, (b_)), (c_), 0, 0, 0) \
       : __builtin_amdgcn_mfma_f32_32x32x16_bf16((a_), (b_), (c_), 0, 0, 0))
#include "attn_fwd2_phase.inc"

  // ---- prologue: K(0), V(0), K(1) staged; K(2), V(1) in registers; S(0) and its statistics
#pragma unroll
  for (int p = 0; p < NPL; ++p) { rk[p] = ld_rows(kp[p], 0, tid); rv[p] = ld_trn(vp[p], a.Lkp, 0, tid); }
#pragma unroll
  for (int p = 0; p < NPL; ++p) { st_rows(Kb0 + p * ROWS_T, rk[p], tid); st_trn(Vb0 + p * TRN_T, rv[p], tid); }
#pragma unroll
  for (int p = 0; p < NPL; ++p) rk[p] = ld_rows(kp[p], (size_t)min(1, lastt) * 32, tid);
#pragma unroll
  for (int p = 0; p < NPL; ++p) st_rows(Kb1 + p * ROWS_T, rk[p], tid);
#pragma unroll
  for (int p = 0; p < NPL; ++p) { rk[p] = ld_rows(kp[p], (size_t)min(2, lastt) * 32, tid); rv[p] = ld_trn(vp[p], a.Lkp, (size_t)min(1, lastt) * 32, tid); }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) s_nxt[r] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16x8 k0 = KFRAG(Kb0, 0, j), k1 = KFRAG(Kb0, 1, j);
    if constexpr (NPL == 3) {
      const bf16x8 k2 = KFRAG(Kb0, 2, j);
      MB6(s_nxt, k0, k1, k2, qf[j][0], qf[j][1], qf[j][2]);
    } else {
      s_nxt = MB(k1, qf[j][0], s_nxt); s_nxt = MB(k0, qf[j][1], s_nxt); s_nxt = MB(k0, qf[j][0], s_nxt);
    }
  }
  uint32_t hbase = rowkey + (uint32_t)(2 * h) * 0x9E3779B9U;      // + (kt * 16) * G per tile: the hash input of the lane's key pair 0 (keys 4 h, 4 h + 1)
  // the statistics step every tile goes through once its scores exist (s_nxt = scores of tile `KT`): mask the keys past kv_len,
  // running maximum with the lazy rescale of the first form, dropout decisions (prologue only: the loop does them in units)
#define FWD2_TILE_STATS(KT, DO_HASH)                                                                                   \
  do {                                                                                                                 \
    if ((KT) == lastt) {                                                                                               \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) if ((KT) * 32 + CR(r, h) >= a.kv_len) s_nxt[r] = -INFINITY;       \
      mt = -INFINITY;                                                                                                  \
      PM(0); PM(1); PM(2); PM(3);                                                                                      \
    }                                                                                                                  \
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));                                                                            \
    if (__any(mt > m + thr8)) {                                                                                        \
      const float mn = fmaxf(m, mt);                                                                                   \
      const float alpha = __builtin_amdgcn_exp2f(F16 ? (m - mn) * cs : m - mn);                                        \
      lsum *= alpha;                                                                                                   \
      m = mn;                                                                                                          \
      mneg = PB - m * cs;                                                                                              \
      _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) _Pragma("unroll") for (int r = 0; r < 16; ++r) o[t_][r] *= alpha; \
    }                                                                                                                  \
    if (DO_HASH) { _Pragma("unroll") for (int pr = 0; pr < 8; ++pr) { PH1(pr); PH2(pr); PH3(pr); } }                    \
  } while (0)
  // (plain C++ here, not PM: the maximum follows the MFMAs of S(0) directly, and only compiler-visible instructions get the wait
  // states an MFMA result needs before a VALU read - an asm v_max3 here read s_nxt before the last product had landed: a slightly
  // wrong running maximum, i.e. run-to-run last-bit differences and, rarely, an overflowing exponent)
#pragma unroll
  for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s_nxt[r]);
  FWD2_TILE_STATS(0, true);
  s_cur = s_nxt;
  __syncthreads();                          // every wave is through with Kb0 (K(0)): the first PV phase overwrites it
  // one key tile: S(t + 1) from KR with the softmax of tile t, PV(t) from VR with the statistics of tile t + 1; K(t + 2) -> KW,
  // V(t + 1) -> VW, K(t + 3) / V(t + 2) requested.  (Past the end the last tile is re-read and its scores are dropped.)
#define FWD2_ITER(t, KR, KW, VR, VW)                                                                                   \
  do {                                                                                                                 \
    __bf16* kw = (KW); __bf16* vw = (VW);                                                                              \
    const int ktn = min((t) + 3, lastt), vtn = min((t) + 2, lastt);                                                    \
    kf[0][0] = KFRAG(KR, 0, 0); kf[0][1] = KFRAG(KR, 1, 0); if constexpr (NPL == 3) kf[0][2] = KFRAG(KR, 2, 0);        \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) s_nxt[r] = 0.f;                                                     \
    ps = 0.f;                                                                                                          \
    SB();                                                                                                              \
    if constexpr (NPL == 3) FWD2_S(KR); else FWD2_S2(KR);                                                              \
    lsum += ps;                                                                                                        \
    vf[0][0] = VFRAG(VR, 0, 0, 0); vf[0][1] = VFRAG(VR, 1, 0, 0); if constexpr (NPL == 3) vf[0][2] = VFRAG(VR, 2, 0, 0); \
    mt = -INFINITY;                                                                                                    \
    hbase += 16u * 0x9E3779B9U;                                                                                        \
    SB();                                                                                                              \
    if constexpr (NPL == 3) FWD2_PV(VR); else FWD2_PV2(VR);                                                            \
    if ((t) + 1 <= lastt) FWD2_TILE_STATS((t) + 1, false);      /* (the re-read tile past the end is dropped) */          \
    s_cur = s_nxt;                                                                                                     \
    FWD2_SYNC();                                                                                                       \
  } while (0)
  int t = 0;
  for (; t + 1 < ntiles; t += 2) {
    FWD2_ITER(t, Kb1, Kb0, Vb0, Vb1);
    FWD2_ITER(t + 1, Kb0, Kb1, Vb1, Vb0);
  }
  if (t < ntiles) FWD2_ITER(t, Kb1, Kb0, Vb0, Vb1);
#undef KFRAG
#undef VFRAG
#undef SB
#undef PK_SUB
#undef PE1
#undef PE2
#undef PE3
#undef PE4
#undef PM
#undef PH1
#undef PH2
#undef PH3
#undef STK
#undef STV
#undef LDK
#undef LDV
#undef FWD2_S
#undef FWD2_PV
#undef FWD2_S2
#undef FWD2_PV2
#undef PE3B
#undef FWD2_TILE_STATS
#undef FWD2_ITER
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  uint32_t omax = 0u;
  if (qrow < a.Lq) {
    const float inv = vinv / ltot;
    float* op = a.out + ((size_t)b * a.Lq + qrow) * a.ldo + head * D;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ov = make_float4(o[tt][4 * g + 0] * inv, o[tt][4 * g + 1] * inv, o[tt][4 * g + 2] * inv, o[tt][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + 32 * tt + 8 * g + 4 * h) = ov;
        omax = max(omax, mag_bits4(ov));
      }
    if (h == 0 && a.lse) a.lse[(size_t)bh * a.Lq + qrow] = (F16 ? m * cs - PB : m) + log2f(ltot);       // log2 domain
  }
  if (a.mag) {                               // row magnitudes of o (common.h): lanes c, c + 32 hold the two halves of the row's head slice
    omax = max(omax, (uint32_t)__shfl_xor((int)omax, 32, 64));
    if (h == 0 && qrow < a.Lq) atomicMax(a.mag + (size_t)b * a.Lq + qrow, omax);
  }
}
#pragma pop_macro("MB")


namespace {
// element offset (bf16) of the 16-byte chunk `ch` of row `r` of a [rows][128] tile (256-byte rows, 16 chunks)
__device__ __forceinline__ int b2_wide_off(int r, int ch) { return r * 128 + ((ch ^ (r & 15)) << 3); }
#define MF16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a_), (b_), (c_), 0, 0, 0)
#define MF6(acc, x0, x1, x2, y0, y1, y2) \
  do {                                   \
    acc = MF16(x2, y0, acc);             \
    acc = MF16(x0, y2, acc);             \
    acc = MF16(x1, y1, acc);             \
    acc = MF16(x1, y0, acc);             \
    acc = MF16(x0, y1, acc);             \
    acc = MF16(x0, y0, acc);             \
  } while (0)
}  // namespace

// ============================================================================================================================
// backward, fused dK / dV / dQ in one pass (5 GEMM-equivalents).  Block = 128 keys, 8 waves, wave = 16 keys (lane l: key l % 16,
// k-group g = l / 16), every contraction on v_mfma_f32_16x16x32_bf16 - 16-wide tiles keep the per-lane state (K, V fragments 48
// registers, dK, dV accumulators 32) under 256 registers, so two waves share a SIMD (a first form with 32 keys per wave on
// 32x32x16 tiles needed > 400 registers, ran one wave per SIMD and took 3.4 ms where this one takes 2.2 ms at B = 32, S = 2048):
//   S, dP      [32 q x 16 keys] = two 16 x 16 tiles, A = Q / dO rows from LDS, B = the lane's K / V fragments (registers)
//   dV^T, dK^T [64 d x 16 keys] = four tiles each, A = dO^T / Q^T fragments, B = Pd / dS straight from the S / dP accumulator
//              registers (a lane holds q = 16 qh + 4 g + i: k-slot 8 g + i <-> q = 4 g + i, 8 g + 4 + i <-> q = 16 + 4 g + i)
//   dQ         [32 q x 64 d]    = eight 16 x 16 tiles, one per wave, A = the dS tile T (LDS, written by all waves), B = K^T fragments
// The two waves of a SIMD do not run the same phase at the same time: waves 0-3 ("early") and 4-7 ("late", one half-step behind;
// wave w and w + 4 share a SIMD) alternate two half-steps, separated by workgroup barriers (a lock-step variant of the same
// kernel - all eight waves in phase, transposed operand tiles staged separately - measured 2.36 ms against 2.29):
//     X(t): S / dP of query tile t  +  dQ of tile t - 2 (early) or t - 1 (late)          [MFMA only]
//     Y(t): softmax / splits of tile t (dS -> T[t & 1])  ->  dV / dK of tile t           [VALU, then MFMA]
// so that in every half-step one wave of each SIMD issues MFMAs while the other runs its VALU phase.  That needs tile t + 1 staged
// while tile t is still being read and dS tiles of two query tiles alive: both double-buffered - which fits 160 KB only because
// the TRANSPOSED operand fragments (dO^T / Q^T for dV / dK, K^T for dQ) are no longer staged as separate tiles but read from the
// row-major tiles with ds_read_b64_tr_b16 (within 16 lanes: lane 4 r + c supplies the address of columns 4 c .. 4 c + 3 of row r,
// lane j receives [row0[j], row1[j], row2[j], row3[j]] - measured, tools/ubench/ds_tr_probe.hip).  LDS: Q / dO row tiles 2 x 24 KB,
// T 2 x 24 KB, the block's K rows 48 KB.  The early waves' 256 threads stage everything.
// ============================================================================================================================
namespace {
constexpr int B3_ROWS = 32 * 64;                 // bf16 per plane tile [32 q][64 d]
constexpr int B3_ST = 6 * B3_ROWS;               // one staging buffer: Q planes 0-2, dO planes 0-2
constexpr int B3_T = 32 * 128;                   // bf16 per dS plane [32 q][128 keys]
constexpr int B3_TB = 3 * B3_T;
constexpr int B3_KS = 128 * 64;                  // bf16 per K plane [128 keys][64 d]
constexpr int B3_TS0 = 2 * B3_ST, B3_KS0 = B3_TS0 + 2 * B3_TB, B3_BF16 = B3_KS0 + 3 * B3_KS;
constexpr unsigned B3_LDS_BYTES = B3_BF16 * 2u + 2u * 64u * 4u;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// 16-byte chunk `ch` of row `r`.  Row tiles: chunk ^ (r & 6) serves both the 16-byte reads of S / dP (lane groups mix rows 0-3 /
// 12-15 at one chunk with rows 4-11 at the next) and the transpose reads (32 lanes = 8 consecutive rows x 32 bytes: rows of equal
// parity share their banks and must land in different chunk pairs).  K rows: the transpose reads of dQ take rows {0-3, 8-11} /
// {4-7, 12-15} of every 16 together, so bits 1 and 3 of the row select the chunk pair.
__device__ __forceinline__ int b3_rows_off(int r, int ch) { return r * 64 + ((ch ^ (r & 6)) << 3); }
__device__ __forceinline__ int b3_ks_off(int r, int ch) { return r * 64 + ((ch ^ ((((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2))) << 3); }
__device__ __forceinline__ bf16x8 b3_tr8(const __bf16* lo, const __bf16* hi) {
  const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lo));
  const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(hi));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}
}  // namespace

template <bool DROP>
__global__ __launch_bounds__(512, 1) void emu_attn_bwd_stag_kernel(EmuAttn a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  float* stats = reinterpret_cast<float*>(lds + B3_BF16);       // [2][lse 32 (log2 domain) | delta 32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  int ktile, bh;
  const int nkb = (a.Lk + 127) / 128;
  if (!emu_block(nkb, a.B * a.H, ktile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int key = ktile * 128 + wave * 16 + l16;
  const bool kvalid = key < a.kv_len;
  const int nq = ktile * 128 < a.kv_len ? (a.Lq + 31) / 32 : 0;
  const bool early = wave < 4;
  const int lag = early ? 0 : 1;

  // resident B operands of S / dP: this lane's key, d = 32 ks + 8 g .. + 7
  bf16x8 kf[2][3], vf[2][3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const size_t ro = ((size_t)bh * a.Lkp + key) * D;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks][p] = *reinterpret_cast<const bf16x8*>(a.k[p] + ro + 32 * ks + 8 * g);
      vf[ks][p] = *reinterpret_cast<const bf16x8*>(a.v[p] + ro + 32 * ks + 8 * g);
    }
  }
  // the block's K rows: 3 planes x [128 keys][64 d] = 3 x 1024 chunks, two per thread and plane
  if (nq > 0) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int id = tid + 512 * i, r = id >> 3, ch = id & 7;
        *reinterpret_cast<u32x4*>(lds + B3_KS0 + p * B3_KS + b3_ks_off(r, ch)) =
            *reinterpret_cast<const u32x4*>(a.k[p] + ((size_t)bh * a.Lkp + ktile * 128 + r) * D + ch * 8);
      }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { dk[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // staging (early waves only: thread -> row tid >> 3, chunk tid & 7 of each of the six plane tiles), running pointers
  const int tt = tid & 255;
  u32x4 sg[6];
  float rl = INFINITY, re = 0.f;
  // (uniform plane bases + one 32-bit element offset per thread: the loads take the scalar-base form, no 64-bit pointers in VGPRs)
  const size_t rowbase = (size_t)bh * a.Lqp * D;
  const __bf16* qb0 = a.q[0] + rowbase; const __bf16* qb1 = a.q[1] + rowbase; const __bf16* qb2 = a.q[2] + rowbase;
  const __bf16* db0 = a.d[0] + rowbase; const __bf16* db1 = a.d[1] + rowbase; const __bf16* db2 = a.d[2] + rowbase;
  unsigned goff = (unsigned)((tt >> 3) * D + (tt & 7) * 8);      // < 2^31 elements per (b, head): Lqp * 64
  const int st_o = b3_rows_off(tt >> 3, tt & 7);
#define B3_LOAD(QTI_)                                                                                                  \
  do {                                                                                                                 \
    sg[0] = *reinterpret_cast<const u32x4*>(qb0 + goff); sg[1] = *reinterpret_cast<const u32x4*>(qb1 + goff);          \
    sg[2] = *reinterpret_cast<const u32x4*>(qb2 + goff); sg[3] = *reinterpret_cast<const u32x4*>(db0 + goff);          \
    sg[4] = *reinterpret_cast<const u32x4*>(db1 + goff); sg[5] = *reinterpret_cast<const u32x4*>(db2 + goff);          \
    if (tid < 32) {                                                                                                    \
      const int q_ = (QTI_) * 32 + tid;                                                                                \
      rl = q_ < a.Lq ? a.lse_in[(size_t)bh * a.Lq + q_] : INFINITY;                                                    \
      re = q_ < a.Lq ? a.delta[(size_t)bh * a.Lq + q_] : 0.f;                                                          \
    }                                                                                                                  \
  } while (0)
#define B3_ADVANCE() do { goff += 32 * D; } while (0)
#define B3_STAGE(BUF_)                                                                                                 \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                                      \
      *reinterpret_cast<u32x4*>(lds + (BUF_) * B3_ST + i * B3_ROWS + st_o) = sg[i];                                    \
    if (tid < 32) { stats[(BUF_) * 64 + tid] = rl; stats[(BUF_) * 64 + 32 + tid] = re; }                              \
  } while (0)
  float* part = a.dq_part + ((size_t)ktile * a.B * a.H + bh) * a.Lq * D;
  const int qh_o = wave >> 2, dt_o = wave & 3;          // this wave's dQ output tile
  // where this lane's dS values go in a T buffer: row q = 16 qh + 4 g + i, key column 16 wave + l16 (swizzled chunk)
  int tw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) tw[i] = (4 * g + i) * 128 + ((((2 * wave + (l16 >> 3)) ^ (4 * g)) ^ i) << 3) + (l16 & 7);
  // transpose-read addresses: row 4 g + (l16 >> 2) (+ 16), columns 4 (l16 & 3) .. + 3 of a 16-column block
  const int trq = 4 * g + (l16 >> 2), trc = l16 & 3;
  if (nq > 0 && early) {
    B3_LOAD(0);
    B3_STAGE(0);
    if (nq > 1) B3_ADVANCE();
    B3_LOAD(min(1, nq - 1));
  }
  __syncthreads();                         // tile 0 and the K rows are in LDS
  f32x4 s[2], dp[2];
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) { s[qh] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qh] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  bf16x8 p0, p1, p2, g0, g1, g2;           // Pd and dS of the lane's 8 (query, key) pairs as bf16 triples: element 4 qh + i
#pragma unroll
  for (int e = 0; e < 8; ++e) { p0[e] = p1[e] = p2[e] = g0[e] = g1[e] = g2[e] = (__bf16)0.f; }
  // softmax / dropout / dS algebra + the two exact three-way splits for the four queries 16 QH + 4 g + i of tile T_ (statistics in ST_)
#define B3_SOFTMAX(QH, T_, ST_)                                                                                        \
  do {                                                                                                                 \
    const f32x4 lsq = *reinterpret_cast<const f32x4*>((ST_) + 16 * (QH) + 4 * g);                                      \
    const f32x4 esq = *reinterpret_cast<const f32x4*>((ST_) + 32 + 16 * (QH) + 4 * g);                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
      const int qi = 16 * (QH) + 4 * g + i, e = 4 * (QH) + i;                                                          \
      const float pe = __builtin_amdgcn_exp2f(s[QH][i] - lsq[i]);                                                      \
      const float pr = kvalid ? pe : 0.f;                                                                              \
      float dsc = 1.f;                                                                                                 \
      if (DROP)                                                                                                        \
        dsc = drop_scale(drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + (T_) * 32) + (uint32_t)qi), (uint32_t)key, a.thresh, a.inv_keep); \
      const float pd = pr * dsc;                                                                                       \
      const float ds = pr * (dp[QH][i] * dsc - esq[i]);                                                                \
      {                                                                                                                \
        const __bf16 a_ = (__bf16)pd; const float r1_ = pd - (float)a_; const __bf16 b_ = (__bf16)r1_; const float r2_ = r1_ - (float)b_; \
        p0[e] = a_; p1[e] = b_; p2[e] = (__bf16)r2_;                                                                   \
      }                                                                                                                \
      {                                                                                                                \
        const __bf16 a_ = (__bf16)ds; const float r1_ = ds - (float)a_; const __bf16 b_ = (__bf16)r1_; const float r2_ = r1_ - (float)b_; \
        g0[e] = a_; g1[e] = b_; g2[e] = (__bf16)r2_;                                                                   \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)
  const int nhalf = nq > 0 ? 2 * nq + 3 : 0;
  for (int hs = 0; hs < nhalf; ++hs) {
    const int kk = hs - lag;
    if (kk >= 0) {
      const int t = kk >> 1;
      if (!(kk & 1)) {
        // ---------------- X(t): S / dP of tile t, dQ of an older tile -------------------------------------------------------
        if (t < nq) {
          const __bf16* ST = lds + (t & 1) * B3_ST;
          // (the six products of BOTH k-steps in order of magnitude - x2 y0, x0 y2, x1 y1 | x1 y0, x0 y1 | x0 y0)
#define B3_SDP(acc, TILE, BF)                                                                                          \
  do {                                                                                                                 \
    const int o0_ = b3_rows_off(16 * qh + l16, g), o1_ = b3_rows_off(16 * qh + l16, 4 + g);                             \
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ST + (TILE) + o0_);                                             \
    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ST + (TILE) + B3_ROWS + o0_);                                   \
    const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(ST + (TILE) + 2 * B3_ROWS + o0_);                               \
    const bf16x8 c0 = *reinterpret_cast<const bf16x8*>(ST + (TILE) + o1_);                                             \
    const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(ST + (TILE) + B3_ROWS + o1_);                                   \
    const bf16x8 c2 = *reinterpret_cast<const bf16x8*>(ST + (TILE) + 2 * B3_ROWS + o1_);                               \
    f32x4 t_ = {0.f, 0.f, 0.f, 0.f};                                                                                   \
    t_ = MF16(a2, BF[0][0], t_); t_ = MF16(c2, BF[1][0], t_);                                                           \
    t_ = MF16(a0, BF[0][2], t_); t_ = MF16(c0, BF[1][2], t_);                                                           \
    t_ = MF16(a1, BF[0][1], t_); t_ = MF16(c1, BF[1][1], t_);                                                           \
    t_ = MF16(a1, BF[0][0], t_); t_ = MF16(c1, BF[1][0], t_);                                                           \
    t_ = MF16(a0, BF[0][1], t_); t_ = MF16(c0, BF[1][1], t_);                                                           \
    t_ = MF16(a0, BF[0][0], t_); t_ = MF16(c0, BF[1][0], t_);                                                           \
    acc = t_;                                                                                                          \
  } while (0)
#pragma unroll
          for (int qh = 0; qh < 2; ++qh) {
            B3_SDP(s[qh], 0, kf);
            B3_SDP(dp[qh], 3 * B3_ROWS, vf);
          }
#undef B3_SDP
          // the first half of the tile's VALU work already here, under this wave's own remaining MFMAs (the other half and the
          // T stores follow in Y): balances the two half-steps
          B3_SOFTMAX(0, t, stats + (t & 1) * 64);
        }
        const int td = t - 2 + lag;
        if (td >= 0 && td < nq) {
          // dQ tile of this wave for query tile td: rows 16 qh_o .., columns 16 dt_o ..: sum over the block's 128 keys
          const __bf16* TS = lds + B3_TS0 + (td & 1) * B3_TB;
          const __bf16* KS = lds + B3_KS0;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, mid = {0.f, 0.f, 0.f, 0.f}, big = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const int ot = b2_wide_off(16 * qh_o + l16, 4 * ks + g);
            const int r0 = 32 * ks + 8 * g + (l16 >> 2);
            const int k0 = b3_ks_off(r0, 2 * dt_o + (trc >> 1)) + (trc & 1) * 4, k1 = b3_ks_off(r0 + 4, 2 * dt_o + (trc >> 1)) + (trc & 1) * 4;
            const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(TS + ot);
            const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(TS + B3_T + ot);
            const bf16x8 x2 = *reinterpret_cast<const bf16x8*>(TS + 2 * B3_T + ot);
            const bf16x8 y0 = b3_tr8(KS + k0, KS + k1);
            const bf16x8 y1 = b3_tr8(KS + B3_KS + k0, KS + B3_KS + k1);
            const bf16x8 y2 = b3_tr8(KS + 2 * B3_KS + k0, KS + 2 * B3_KS + k1);
            acc = MF16(x2, y0, acc); acc = MF16(x0, y2, acc); acc = MF16(x1, y1, acc);
            mid = MF16(x1, y0, mid); mid = MF16(x0, y1, mid);
            big = MF16(x0, y0, big);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int q = td * 32 + 16 * qh_o + 4 * g + i;
            if (q < a.Lq) part[(size_t)q * D + 16 * dt_o + l16] = (acc[i] + mid[i]) + big[i];
          }
        }
      } else {
        // ---------------- Y(t): stage tile t + 1 (early), softmax of tile t, dV / dK of tile t ---------------------------------
        if (early && t + 1 < nq) {
          B3_STAGE((t + 1) & 1);
          if (t + 2 < nq) B3_ADVANCE();
          B3_LOAD(min(t + 2, nq - 1));
        }
        if (t < nq) {
          B3_SOFTMAX(1, t, stats + (t & 1) * 64);
          __bf16* TW = lds + B3_TS0 + (t & 1) * B3_TB;
#pragma unroll
          for (int qh = 0; qh < 2; ++qh)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              __bf16* tp = TW + tw[i] + qh * (16 * 128);
              tp[0] = g0[4 * qh + i]; tp[B3_T] = g1[4 * qh + i]; tp[2 * B3_T] = g2[4 * qh + i];
            }
          // dV^T[d][key] += dO^T[d][q] . Pd[q][key] ;  dK^T[d][key] += Qs^T[d][q] . dS[q][key]: the transposed fragments come out
          // of the row tiles through the transpose read (k-slots 8 g + i <-> q = 4 g + i, 8 g + 4 + i <-> q = 16 + 4 g + i)
          const __bf16* ST = lds + (t & 1) * B3_ST;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const int o0 = b3_rows_off(trq, 2 * dt + (trc >> 1)) + (trc & 1) * 4, o1 = b3_rows_off(trq + 16, 2 * dt + (trc >> 1)) + (trc & 1) * 4;
            const bf16x8 t0 = b3_tr8(ST + o0, ST + o1);
            const bf16x8 t1 = b3_tr8(ST + B3_ROWS + o0, ST + B3_ROWS + o1);
            const bf16x8 t2 = b3_tr8(ST + 2 * B3_ROWS + o0, ST + 2 * B3_ROWS + o1);
            const bf16x8 o0_ = b3_tr8(ST + 3 * B3_ROWS + o0, ST + 3 * B3_ROWS + o1);
            const bf16x8 o1_ = b3_tr8(ST + 4 * B3_ROWS + o0, ST + 4 * B3_ROWS + o1);
            const bf16x8 o2_ = b3_tr8(ST + 5 * B3_ROWS + o0, ST + 5 * B3_ROWS + o1);
            MF6(dv[dt], o0_, o1_, o2_, p0, p1, p2);
            MF6(dk[dt], t0, t1, t2, g0, g1, g2);
          }
        }
      }
    }
    __syncthreads();
  }
#undef B3_LOAD
#undef B3_ADVANCE
#undef B3_STAGE
#undef B3_SOFTMAX
  uint32_t gmax = 0u;
  if (key < a.Lk) {
    float* pk = a.dk + ((size_t)b * a.Lk + key) * a.ldk + head * D;
    float* pv = a.dv + ((size_t)b * a.Lk + key) * a.ldv + head * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      // Q was pre-scaled by log2(e)/8: dK = dS^T.Q / 8 = (dS^T.Qs) * ln 2
      const float4 gk = make_float4(dk[dt][0] * LN2, dk[dt][1] * LN2, dk[dt][2] * LN2, dk[dt][3] * LN2);
      const float4 gv = make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
      *reinterpret_cast<float4*>(pk + 16 * dt + 4 * g) = gk;
      *reinterpret_cast<float4*>(pv + 16 * dt + 4 * g) = gv;
      gmax = max(gmax, max(mag_bits4(gk), mag_bits4(gv)));
    }
  }
  if (a.mag) {                               // row magnitudes of [dq | dk | dv] (common.h): the four k-groups of a key hold its row
    gmax = max(gmax, (uint32_t)__shfl_xor((int)gmax, 16, 64));
    gmax = max(gmax, (uint32_t)__shfl_xor((int)gmax, 32, 64));
    if (g == 0 && key < a.Lk) atomicMax(a.mag + (size_t)b * a.Lk + key, gmax);
  }
}

// delta[bh][q] = sum_d dO[q][d] * O[q][d]  (f32; 16 lanes per (q, head))
__global__ __launch_bounds__(256) void emu_attn_delta_kernel(const float* __restrict__ o, int ldo, const float* __restrict__ dout,
                                                             int lddo, float* __restrict__ delta, int B, int H, int Lq) {
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int l16 = threadIdx.x & 15;
  if (g >= (long)B * Lq * H) return;
  const int head = (int)(g % H);
  const long bq = g / H;
  const int q = (int)(bq % Lq), b = (int)(bq / Lq);
  const float4 x = *reinterpret_cast<const float4*>(o + ((size_t)b * Lq + q) * ldo + head * D + l16 * 4);
  const float4 y = *reinterpret_cast<const float4*>(dout + ((size_t)b * Lq + q) * lddo + head * D + l16 * 4);
  float s = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
  if (l16 == 0) delta[((size_t)b * H + head) * Lq + q] = s;
}

// dq[b][q][head * 64 + d] = 0.125 * sum_kb part[kb][bh][q][d] in key-block order (one float4 per thread)
// scale: null = 0.125 (the bf16x3 backward: Q pre-scaled by log2(e) / 8 ...); the f16x2 backward leaves 0.125 / (sK sS) per (b, head) there
__global__ __launch_bounds__(256) void emu_attn_dq_reduce_kernel(const float* __restrict__ part, int nkb, float* __restrict__ dq,
                                                                 int ldq, int B, int H, int Lq, uint32_t* __restrict__ mag,
                                                                 const float* __restrict__ scale) {
  // blocks walk the (bh, q) rows from the LAST one down: the backward kernel wrote the high heads' partials last, so they are the ones
  // still in the 256 MB memory-side cache
  const long i = (long)(gridDim.x - 1 - blockIdx.x) * 256 + threadIdx.x;          // (bh, q, d / 4)
  const long n = (long)B * H * Lq * 16;
  uint32_t qmax = 0u;
  if (i < n) {
    const int d4 = (int)(i & 15);
    const long bq = i >> 4;
    const int q = (int)(bq % Lq);
    const int bh = (int)(bq / Lq), b = bh / H, head = bh - b * H;
    const size_t stride = (size_t)B * H * Lq * D;
    const float* p = part + ((size_t)bh * Lq + q) * D + d4 * 4;
    float4 s = *reinterpret_cast<const float4*>(p);
    // eight partials in flight per thread, added in key-block order (the sum is the serial loop's, bit for bit)
    for (int k0 = 1; k0 < nkb; k0 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = k0 + j < nkb ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)(k0 + j) * stride)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < nkb) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    }
    const float sc = scale ? scale[bh] : 0.125f;
    const float4 r = make_float4(s.x * sc, s.y * sc, s.z * sc, s.w * sc);
    *reinterpret_cast<float4*>(dq + ((size_t)b * Lq + q) * ldq + head * D + d4 * 4) = r;
    qmax = mag_bits4(r);
    if (mag) {                             // dq's share of the row magnitudes of [dq | dk | dv] (common.h): 16 lanes per row
      qmax = group_max_u32<16>(qmax);
      if (d4 == 0) atomicMax(mag + (size_t)b * Lq + q, qmax);
    }
  }
}

}  // namespace hoisdf

using namespace hoisdf;

namespace {
inline long pad128(long L) { return (L + 127) / 128 * 128; }
inline size_t plane_elems(int B, int H, long Lp) { return (size_t)B * H * Lp * 64; }

// one tensor's planes inside a workspace: rows r[3] then transposed t[3] (either group may be absent)
struct Planes { __bf16 *r[3], *t[3]; };
inline Planes carve(__bf16*& w, size_t n, bool rows, bool trn) {
  Planes p{};
  for (int i = 0; i < 3; ++i) { p.r[i] = rows ? w : nullptr; if (rows) w += n; }
  for (int i = 0; i < 3; ++i) { p.t[i] = trn ? w : nullptr; if (trn) w += n; }
  return p;
}
// mag != null: the f16x2 planes (two planes, scaled per (sample, head) by the head magnitudes mag[head * B + b]); null: the bf16 planes
int convert(const float* src, int ld, int L, int Lp, int B, int H, float scale, const Planes& p, hipStream_t st, const uint32_t* mag = nullptr) {
  const long nblk = (long)B * H * (Lp / 64);
  if (mag) hipLaunchKernelGGL(emu_attn_convert_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, st, src, ld, L, Lp, B, H, scale, p.r[0], p.r[1],
                              p.r[2], p.t[0], p.t[1], p.t[2], mag);
  else hipLaunchKernelGGL(emu_attn_convert_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, st, src, ld, L, Lp, B, H, scale, p.r[0], p.r[1],
                          p.r[2], p.t[0], p.t[1], p.t[2], mag);
  return check_launch("attention_emu_convert");
}
int check_emu(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, int B, int H, int Lq, int Lk, int kv_len,
              float drop_p, const char* who) {
  HOISDF_REQUIRE(q && k && v, HOISDF_ERR_INVALID, "%s: null pointer", who);
  HOISDF_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && kv_len > 0 && kv_len <= Lk, HOISDF_ERR_INVALID,
                 "%s: bad sizes B=%d H=%d Lq=%d Lk=%d kv_len=%d", who, B, H, Lq, Lk, kv_len);
  HOISDF_REQUIRE(ldq >= H * 64 && ldk >= H * 64 && ldv >= H * 64 && ((ldq | ldk | ldv) & 3) == 0 &&
                     (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0,
                 HOISDF_ERR_INVALID, "%s: leading dims must be multiples of 4 and >= H*64, pointers 16-byte aligned", who);
  HOISDF_REQUIRE(drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID, "%s: drop_p=%f", who, drop_p);
  return HOISDF_OK;
}
}  // namespace

// mode 0: forward only (Q rows, K rows, V^T).  mode 2: forward that also leaves the V rows the backward needs (Q rows, K rows, V rows, V^T).
extern "C" long hoisdf_attention_emu_workspace(int B, int H, int Lq, int Lk, int mode) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  const size_t q = plane_elems(B, H, pad128(Lq)), k = plane_elems(B, H, pad128(Lk));
  const size_t planes = mode == 0 ? 3 * q + 6 * k : 3 * q + 9 * k;
  return (long)(planes * sizeof(__bf16));
}

namespace {
// the forward over planes that are in the workspace already (layout (kept form): [Q rows | K rows | V rows, V^T]; forward-only
// form: [Q rows | K rows | V^T])
// q_hm != null: the planes are the f16x2 ones (two planes, made with those head magnitudes): the f16 kernel
int fwd_over_planes(float* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed,
                    void* workspace, int keep, hipStream_t st, uint32_t* o_mag, const uint32_t* q_hm = nullptr, const uint32_t* k_hm = nullptr,
                    const uint32_t* v_hm = nullptr) {
  const int Lqp = (int)pad128(Lq), Lkp = (int)pad128(Lk);
  __bf16* w = reinterpret_cast<__bf16*>(workspace);
  const size_t nq = plane_elems(B, H, Lqp), nk = plane_elems(B, H, Lkp);
  const Planes pq = carve(w, nq, true, false);
  const Planes pk = carve(w, nk, true, false);
  const Planes pv = carve(w, nk, keep != 0, true);
  EmuAttn a{};
  for (int i = 0; i < 3; ++i) { a.q[i] = pq.r[i]; a.k[i] = pk.r[i]; a.vt[i] = pv.t[i]; }
  a.out = o; a.lse = lse; a.ldo = ldo; a.mag = o_mag; a.q_hm = q_hm; a.k_hm = k_hm; a.v_hm = v_hm;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.Lqp = Lqp; a.Lkp = Lkp; a.kv_len = kv_len;
  a.drop_p = drop_p; a.inv_keep = 1.f / (1.f - drop_p); a.thresh = drop_threshold(drop_p); a.seed = seed;
  static int form = -1;                       // HOISDF_EMU_ATTN_FWD=1: the first (unpipelined) form (A/B runs)
  if (form < 0) { const char* e = getenv("HOISDF_EMU_ATTN_FWD"); form = (e && atoi(e) == 1) ? 1 : 2; }
  const dim3 fgrid(cdiv(Lq, 128) * 8 * cdiv(B * H, 8));
  if (q_hm) {
    // (P is carried as 2^6 P up to 2^14 and the keep factor 1 / (1 - p) goes in before the f16 split: p < 0.75 keeps it below 65504)
    HOISDF_REQUIRE(drop_p < 0.75f, HOISDF_ERR_INVALID, "attention_fwd_emu (f16x2 form): drop_p = %f, must be below 0.75", drop_p);
    if (drop_p > 0.f) hipLaunchKernelGGL((emu_attn_fwd2_kernel<true, 2, true>), fgrid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((emu_attn_fwd2_kernel<false, 2, true>), fgrid, dim3(256), 0, st, a);
  } else if (form == 1) hipLaunchKernelGGL(emu_attn_fwd_kernel, fgrid, dim3(256), 0, st, a);
  else if (drop_p > 0.f) hipLaunchKernelGGL((emu_attn_fwd2_kernel<true, 3, false>), fgrid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((emu_attn_fwd2_kernel<false, 3, false>), fgrid, dim3(256), 0, st, a);
  return check_launch("attention_fwd_emu");
}
}  // namespace

extern "C" int hoisdf_attention_fwd_emu(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
                                        int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len, float drop_p,
                                        uint64_t seed, void* workspace, long workspace_bytes, int keep, void* stream) {
  return attention_fwd_emu_mag(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, H, Lq, Lk, kv_len, drop_p, seed, workspace, workspace_bytes, keep,
                               nullptr, stream, nullptr, nullptr, nullptr);
}
// the f16x2 form of the forward: q_mag / k_mag / v_mag = the head magnitudes (include/hoisdf.h) of q, k and v - one power-of-two scale
// per (sample, head) and operand, two f16 planes per operand, three products per product; o_mag receives o's row magnitudes when given
extern "C" int hoisdf_attention_fwd_emu_mag(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
                                            int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len, float drop_p,
                                            uint64_t seed, void* workspace, long workspace_bytes, int keep, const uint32_t* q_mag,
                                            const uint32_t* k_mag, const uint32_t* v_mag, uint32_t* o_mag, void* stream) {
  HOISDF_REQUIRE(q_mag && k_mag && v_mag, HOISDF_ERR_INVALID, "attention_fwd_emu_mag: the head magnitudes of q, k and v are required (hoisdf_head_mag_measure)");
  return attention_fwd_emu_mag(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, H, Lq, Lk, kv_len, drop_p, seed, workspace, workspace_bytes, keep,
                               o_mag, stream, q_mag, k_mag, v_mag);
}
// (internal, common.h) + o_mag: the row magnitudes of o (zero on entry; null = not wanted)
int hoisdf::attention_fwd_emu_mag(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, float* lse,
                                  int B, int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed, void* workspace,
                                  long workspace_bytes, int keep, uint32_t* o_mag, void* stream, const uint32_t* q_hm, const uint32_t* k_hm,
                                  const uint32_t* v_hm) {
  if (int rc = check_emu(q, k, v, ldq, ldk, ldv, B, H, Lq, Lk, kv_len, drop_p, "attention_fwd_emu")) return rc;
  HOISDF_REQUIRE(o && workspace && ldo >= H * 64 && (ldo & 3) == 0 && (((uintptr_t)o | (uintptr_t)workspace) & 15) == 0,
                 HOISDF_ERR_INVALID, "attention_fwd_emu: bad output / workspace");
  HOISDF_REQUIRE((!q_hm && !k_hm && !v_hm) || (q_hm && k_hm && v_hm), HOISDF_ERR_INVALID, "attention_fwd_emu: head magnitudes of all of q, k, v or of none");
  const long need = hoisdf_attention_emu_workspace(B, H, Lq, Lk, keep ? 2 : 0);
  HOISDF_REQUIRE(workspace_bytes >= need, HOISDF_ERR_WORKSPACE, "attention_fwd_emu: workspace %ld < %ld bytes", workspace_bytes, need);
  const int Lqp = (int)pad128(Lq), Lkp = (int)pad128(Lk);
  hipStream_t st = as_stream(stream);
  __bf16* w = reinterpret_cast<__bf16*>(workspace);
  const size_t nq = plane_elems(B, H, Lqp), nk = plane_elems(B, H, Lkp);
  const Planes pq = carve(w, nq, true, false);
  const Planes pk = carve(w, nk, true, false);
  const Planes pv = carve(w, nk, keep != 0, true);
  if (int rc = convert(q, ldq, Lq, Lqp, B, H, QS2, pq, st, q_hm)) return rc;
  if (int rc = convert(k, ldk, Lk, Lkp, B, H, 1.f, pk, st, k_hm)) return rc;
  if (int rc = convert(v, ldv, Lk, Lkp, B, H, 1.f, pv, st, v_hm)) return rc;
  return fwd_over_planes(o, ldo, lse, B, H, Lq, Lk, kv_len, drop_p, seed, workspace, keep, st, o_mag, q_hm, k_hm, v_hm);
}

// ---- 16-bit-operand evaluation attention (BASELINE configs[4] "fp16 MFMA attention"): the pipelined forward over TWO bf16 planes per
// operand (hi + lo: 16 significant bits, the f32 exponent range - no scaling, no overflow at trained sigma gates), three MFMA
// products per product, f32 softmax and accumulation.  No dropout, no LSE.
extern "C" long hoisdf_attention_bf16x2_workspace(int B, int H, int Lq, int Lk) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  return (long)((2 * plane_elems(B, H, pad128(Lq)) + 4 * plane_elems(B, H, pad128(Lk))) * sizeof(__bf16));
}

extern "C" int hoisdf_attention_fwd_bf16x2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                           int B, int H, int Lq, int Lk, int kv_len, void* workspace, long workspace_bytes, void* stream) {
  if (int rc = check_emu(q, k, v, ldq, ldk, ldv, B, H, Lq, Lk, kv_len, 0.f, "attention_fwd_bf16x2")) return rc;
  HOISDF_REQUIRE(o && workspace && ldo >= H * 64 && (ldo & 3) == 0 && (((uintptr_t)o | (uintptr_t)workspace) & 15) == 0,
                 HOISDF_ERR_INVALID, "attention_fwd_bf16x2: bad output / workspace");
  const long need = hoisdf_attention_bf16x2_workspace(B, H, Lq, Lk);
  HOISDF_REQUIRE(workspace_bytes >= need, HOISDF_ERR_WORKSPACE, "attention_fwd_bf16x2: workspace %ld < %ld bytes", workspace_bytes, need);
  const int Lqp = (int)pad128(Lq), Lkp = (int)pad128(Lk);
  hipStream_t st = as_stream(stream);
  __bf16* w = reinterpret_cast<__bf16*>(workspace);
  const size_t nq = plane_elems(B, H, Lqp), nk = plane_elems(B, H, Lkp);
  Planes pq{}, pk{}, pv{};
  pq.r[0] = w; pq.r[1] = w + nq; w += 2 * nq;
  pk.r[0] = w; pk.r[1] = w + nk; w += 2 * nk;
  pv.t[0] = w; pv.t[1] = w + nk;
  if (int rc = convert(q, ldq, Lq, Lqp, B, H, QS2, pq, st)) return rc;
  if (int rc = convert(k, ldk, Lk, Lkp, B, H, 1.f, pk, st)) return rc;
  if (int rc = convert(v, ldv, Lk, Lkp, B, H, 1.f, pv, st)) return rc;
  EmuAttn a{};
  for (int i = 0; i < 2; ++i) { a.q[i] = pq.r[i]; a.k[i] = pk.r[i]; a.vt[i] = pv.t[i]; }
  a.out = o; a.lse = nullptr; a.ldo = ldo;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.Lqp = Lqp; a.Lkp = Lkp; a.kv_len = kv_len;
  a.drop_p = 0.f; a.inv_keep = 1.f; a.thresh = 0; a.seed = 0;
  const dim3 fgrid(cdiv(Lq, 128) * 8 * cdiv(B * H, 8));
  hipLaunchKernelGGL((emu_attn_fwd2_kernel<false, 2, false>), fgrid, dim3(256), 0, st, a);
  return check_launch("attention_fwd_bf16x2");
}

// (internal, common.h) the plane addresses of a forward workspace as targets of linear_fwd_emu_qkv: q_part for a GEMM over the
// B Lq query rows, kv_part for one over the B Lk memory rows (the same struct twice when Lq == Lk and one GEMM makes all three parts)
void hoisdf::attention_emu_plane_targets(void* workspace, int B, int H, int Lq, int Lk, int keep, QkvPlanes& qp, QkvPlanes& kvp) {
  const int Lqp = (int)pad128(Lq), Lkp = (int)pad128(Lk);
  __bf16* w = reinterpret_cast<__bf16*>(workspace);
  const size_t nq = plane_elems(B, H, Lqp), nk = plane_elems(B, H, Lkp);
  const Planes pq = carve(w, nq, true, false);
  const Planes pk = carve(w, nk, true, false);
  const Planes pv = carve(w, nk, keep != 0, true);
  qp = QkvPlanes{}; kvp = QkvPlanes{};
  qp.on = kvp.on = 1; qp.H = kvp.H = H; qp.E = kvp.E = H * 64; qp.qscale = kvp.qscale = QS2;
  qp.L = Lq; qp.Lp = Lqp; kvp.L = Lk; kvp.Lp = Lkp;
  for (int i = 0; i < 3; ++i) {
    qp.r[0][i] = pq.r[i];
    kvp.r[1][i] = pk.r[i]; kvp.r[2][i] = pv.r[i]; kvp.vt[i] = pv.t[i];
    if (Lq == Lk) { kvp.r[0][i] = pq.r[i]; qp.r[1][i] = pk.r[i]; qp.r[2][i] = pv.r[i]; qp.vt[i] = pv.t[i]; }
  }
}

int hoisdf::attention_fwd_emu_planes(float* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len, float drop_p,
                                     uint64_t seed, void* workspace, int keep, void* stream, uint32_t* o_mag) {
  HOISDF_REQUIRE(o && workspace && ldo >= H * 64 && (ldo & 3) == 0 && (((uintptr_t)o | (uintptr_t)workspace) & 15) == 0 && B > 0 &&
                     H > 0 && Lq > 0 && Lk > 0 && kv_len > 0 && kv_len <= Lk && drop_p >= 0.f && drop_p < 1.f,
                 HOISDF_ERR_INVALID, "attention_fwd_emu_planes: bad arguments");
  return fwd_over_planes(o, ldo, lse, B, H, Lq, Lk, kv_len, drop_p, seed, workspace, keep, as_stream(stream), o_mag);
}

// backward workspace: dO rows (3 planes) + the dQ partials [ceil(Lk / 128)][B H][Lq][64] f32
//   (+ Q, K, V row planes when the forward did not keep them: kept == 0)
extern "C" long hoisdf_attention_bwd_emu_workspace(int B, int H, int Lq, int Lk, int kept) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  const size_t q = plane_elems(B, H, pad128(Lq)), k = plane_elems(B, H, pad128(Lk));
  size_t bytes = 3 * q * sizeof(__bf16) + (size_t)cdiv(Lk, 128) * B * H * Lq * 64 * sizeof(float);
  if (!kept) bytes += (3 * q + 6 * k) * sizeof(__bf16);
  return (long)bytes;
}

extern "C" int hoisdf_attention_bwd_emu(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                        int ldo, const float* dout, int lddo, const float* lse, float* delta, float* dq, float* dk,
                                        float* dv, int B, int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed,
                                        const void* fwd_workspace, void* workspace, long workspace_bytes, void* stream) {
  return attention_bwd_emu_mag(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, delta, dq, dk, dv, B, H, Lq, Lk, kv_len, drop_p, seed,
                               fwd_workspace, workspace, workspace_bytes, nullptr, stream, nullptr, nullptr, nullptr, nullptr);
}
// the backward in the f16x2 form (emu_attn_bwd4h_kernel): q_mag / k_mag / v_mag as in hoisdf_attention_fwd_emu_mag, do_mag = the head
// magnitudes of dout; fwd_workspace = the planes a forward OF THE SAME FORM kept (keep = 1), or NULL (q, k, v are converted here)
extern "C" int hoisdf_attention_bwd_emu_mag(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                            int ldo, const float* dout, int lddo, const float* lse, float* delta, float* dq, float* dk,
                                            float* dv, int B, int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed,
                                            const void* fwd_workspace, void* workspace, long workspace_bytes, const uint32_t* q_mag,
                                            const uint32_t* k_mag, const uint32_t* v_mag, const uint32_t* do_mag, uint32_t* g_mag, void* stream) {
  HOISDF_REQUIRE(q_mag && k_mag && v_mag && do_mag, HOISDF_ERR_INVALID, "attention_bwd_emu_mag: the head magnitudes of q, k, v and of dout are required");
  return attention_bwd_emu_mag(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, delta, dq, dk, dv, B, H, Lq, Lk, kv_len, drop_p, seed,
                               fwd_workspace, workspace, workspace_bytes, g_mag, stream, q_mag, k_mag, v_mag, do_mag);
}
// (internal, common.h) + g_mag: ONE array of row magnitudes for dq, dk and dv together (rows b L + s of a [dq | dk | dv] matrix: Lq == Lk;
// zero on entry; null = not wanted)
int hoisdf::attention_bwd_emu_mag(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                                  const float* dout, int lddo, const float* lse, float* delta, float* dq, float* dk, float* dv, int B,
                                  int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed, const void* fwd_workspace,
                                  void* workspace, long workspace_bytes, uint32_t* g_mag, void* stream, const uint32_t* q_hm,
                                  const uint32_t* k_hm, const uint32_t* v_hm, const uint32_t* do_hm) {
  const bool h2 = q_hm && k_hm && v_hm && do_hm;     // the f16x2 form: two scaled f16 planes per operand (emu_attn_bwd4h_kernel)
  HOISDF_REQUIRE(h2 || (!q_hm && !k_hm && !v_hm && !do_hm), HOISDF_ERR_INVALID, "attention_bwd_emu: head magnitudes of all of q, k, v, dout or of none");
  HOISDF_REQUIRE(!g_mag || Lq == Lk, HOISDF_ERR_INVALID, "attention_bwd_emu: one row-magnitude array for dq, dk, dv needs Lq == Lk");
  HOISDF_REQUIRE(!h2 || drop_p < 0.75f, HOISDF_ERR_INVALID, "attention_bwd_emu (f16x2 form): drop_p = %f, must be below 0.75 (as in the forward)", drop_p);
  // with the forward's planes (fwd_workspace) q, k, v themselves are not read: they may be null; ldq / ldk / ldv still give the
  // layouts of dq / dk / dv
  if (int rc = check_emu(fwd_workspace && !q ? o : q, fwd_workspace && !k ? o : k, fwd_workspace && !v ? o : v, ldq, ldk, ldv, B, H, Lq,
                         Lk, kv_len, drop_p, "attention_bwd_emu")) return rc;
  HOISDF_REQUIRE(o && dout && lse && delta && dq && dk && dv && workspace, HOISDF_ERR_INVALID, "attention_bwd_emu: null pointer");
  HOISDF_REQUIRE(ldo >= H * 64 && lddo >= H * 64 && ((ldo | lddo) & 3) == 0 &&
                     (((uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)workspace |
                       (uintptr_t)fwd_workspace) & 15) == 0,
                 HOISDF_ERR_INVALID, "attention_bwd_emu: bad leading dims / alignment");
  const long need = hoisdf_attention_bwd_emu_workspace(B, H, Lq, Lk, fwd_workspace ? 1 : 0);
  HOISDF_REQUIRE(workspace_bytes >= need, HOISDF_ERR_WORKSPACE, "attention_bwd_emu: workspace %ld < %ld bytes", workspace_bytes, need);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(emu_attn_bwd_stag_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)B3_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(emu_attn_bwd_stag_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)B3_LDS_BYTES) != hipSuccess) {
      set_error("attention_bwd_emu: cannot raise the dynamic LDS limit to %u bytes", B3_LDS_BYTES);
      return HOISDF_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int Lqp = (int)pad128(Lq), Lkp = (int)pad128(Lk);
  hipStream_t st = as_stream(stream);
  const size_t nq = plane_elems(B, H, Lqp), nk = plane_elems(B, H, Lkp);
  __bf16* w = reinterpret_cast<__bf16*>(workspace);
  const Planes pd = carve(w, nq, true, false);
  float* part = reinterpret_cast<float*>(w);
  w += (size_t)cdiv(Lk, 128) * B * H * Lq * 64 * 2;                 // (float = 2 bf16 slots)
  Planes pq, pk, pv;
  if (fwd_workspace) {          // the planes hoisdf_attention_fwd_emu(keep = 1) left for the same q, k, v
    __bf16* f = reinterpret_cast<__bf16*>(const_cast<void*>(fwd_workspace));
    pq = carve(f, nq, true, false); pk = carve(f, nk, true, false); pv = carve(f, nk, true, true);
  } else {
    pq = carve(w, nq, true, false); pk = carve(w, nk, true, false); pv = carve(w, nk, true, false);
    if (int rc = convert(q, ldq, Lq, Lqp, B, H, QS2, pq, st, q_hm)) return rc;
    if (int rc = convert(k, ldk, Lk, Lkp, B, H, 1.f, pk, st, k_hm)) return rc;
    if (int rc = convert(v, ldv, Lk, Lkp, B, H, 1.f, pv, st, v_hm)) return rc;
  }
  if (int rc = convert(dout, lddo, Lq, Lqp, B, H, 1.f, pd, st, do_hm)) return rc;
  const long ng = (long)B * Lq * H;
  hipLaunchKernelGGL(emu_attn_delta_kernel, dim3((unsigned)((ng * 16 + 255) / 256)), dim3(256), 0, st, o, ldo, dout, lddo, delta, B, H, Lq);
  if (int rc = check_launch("attention_emu_delta")) return rc;
  EmuAttn a{};
  for (int i = 0; i < 3; ++i) {
    a.q[i] = pq.r[i]; a.k[i] = pk.r[i]; a.v[i] = pv.r[i]; a.d[i] = pd.r[i];
  }
  a.lse_in = lse; a.delta = delta; a.dq_part = part; a.dk = dk; a.dv = dv; a.ldk = ldk; a.ldv = ldv; a.mag = g_mag;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.Lqp = Lqp; a.Lkp = Lkp; a.kv_len = kv_len;
  a.drop_p = drop_p; a.inv_keep = 1.f / (1.f - drop_p); a.thresh = drop_threshold(drop_p); a.seed = seed;
  if (h2) {
    a.q_hm = q_hm; a.k_hm = k_hm; a.v_hm = v_hm; a.d_hm = do_hm;
    a.dq_scale = reinterpret_cast<float*>(pd.r[2]);          // (the third dO plane is free in this form: the reduce pass's B H factors live there)
    if (int rc = attention_bwd4h_emu_launch(a, st)) return rc;
    const long n4h = (long)B * H * Lq * 16;
    hipLaunchKernelGGL(emu_attn_dq_reduce_kernel, dim3((unsigned)((n4h + 255) / 256)), dim3(256), 0, st, part, cdiv(kv_len, 128), dq, ldq,
                       B, H, Lq, g_mag, (const float*)a.dq_scale);
    return check_launch("attention_bwd_emu dq reduce");
  }
  // HOISDF_EMU_ATTN_BWD: (default) the round-5 kernel with dQ summed across the key blocks through an ordered running sum in L2;
  // "4p": the same kernel with the per-key-block partial buffer + reduce pass; "3": the round-3 kernel (8 waves x 16 keys) - A/B runs
  static int form = -1;
  if (form < 0) { const char* e = getenv("HOISDF_EMU_ATTN_BWD"); form = !e ? 5 : (atoi(e) == 3 ? 3 : (e[0] == '4' && e[1] == 'p' ? 4 : 5)); }
  if (form == 5) {
    // HOISDF_EMU_ATTN_BWD_CHAIN=G (default 1 = one partial per key block + the reduce pass): chains of G key blocks add their dQ
    // contributions to ONE running sum in order, through the XCD's L2.  An EXPERIMENT, off by default: G = 16 takes the launch from
    // 1.61 to ~0.6 GB of HBM traffic with bit-identical results, at the SAME speed (2.150 vs 2.155 ms per call at B = 32, S = 2048;
    // G = 4: 2.18, G = 2: 2.23, and slower at 512 queries: 0.83 vs 0.74) - the partial traffic was never what bounds this kernel (the
    // board sits at its 1400 W cap, HBM at 0.8 TB/s) and the waits of the chain eat what the reduce pass cost.  It is also only
    // correct while all key blocks of a (b, head) run on one XCD (observed dispatch behaviour, not a HIP guarantee).
    static int G = -1;
    if (G < 0) { const char* e = getenv("HOISDF_EMU_ATTN_BWD_CHAIN"); G = e ? atoi(e) : 1; if (G < 1) G = 1; }
    const int nkb = cdiv(Lk, 128), nlive = cdiv(kv_len, 128), ngrp = cdiv(nlive, G);
    // the counters sit behind the ngrp running sums (the region holds nkb partial slots: always room unless G = 1)
    int* flags = reinterpret_cast<int*>(part + (size_t)ngrp * B * H * Lq * 64);
    const size_t flag_bytes = (size_t)B * H * nkb * 4 * sizeof(int);
    const bool chain = G > 1 && nlive > 1 && (size_t)(nkb - ngrp) * B * H * Lq * 64 * sizeof(float) >= flag_bytes &&
                       !g_mag;       // (a chain of all key blocks writes dq itself: nobody would fold its magnitude)
    if (chain) {
      a.dq = dq; a.ldq = ldq; a.dq_flags = flags; a.chain_group = G;
      if (hipMemsetAsync(flags, 0, flag_bytes, st) != hipSuccess) { set_error("attention_bwd_emu: clearing the chain counters failed"); return HOISDF_ERR_LAUNCH; }
      if (int rc = attention_bwd4_emu_launch(a, true, st)) return rc;
      if (ngrp == 1) return HOISDF_OK;          // (the chain's last block wrote dq itself)
      const long n4c = (long)B * H * Lq * 16;
      hipLaunchKernelGGL(emu_attn_dq_reduce_kernel, dim3((unsigned)((n4c + 255) / 256)), dim3(256), 0, st, part, ngrp, dq, ldq, B, H, Lq, g_mag, (const float*)nullptr);
      return check_launch("attention_bwd_emu dq reduce");
    }
    if (int rc = attention_bwd4_emu_launch(a, false, st)) return rc;
  } else
  if (form == 4) {
    if (int rc = attention_bwd4_emu_launch(a, false, st)) return rc;
  } else {
    const dim3 grid(cdiv(Lk, 128) * 8 * cdiv(B * H, 8));
    if (drop_p > 0.f) hipLaunchKernelGGL(emu_attn_bwd_stag_kernel<true>, grid, dim3(512), B3_LDS_BYTES, st, a);
    else hipLaunchKernelGGL(emu_attn_bwd_stag_kernel<false>, grid, dim3(512), B3_LDS_BYTES, st, a);
    if (int rc = check_launch("attention_bwd_emu")) return rc;
  }
  const long n4 = (long)B * H * Lq * 16;
  hipLaunchKernelGGL(emu_attn_dq_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, cdiv(kv_len, 128), dq, ldq,
                     B, H, Lq, g_mag, (const float*)nullptr);
  return check_launch("attention_bwd_emu dq reduce");
}
