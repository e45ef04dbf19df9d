// Split-precision attention for TRAINING (opt-in, cfg.attention_split): every contraction of the forward and the backward
// runs on the 16-bit MFMA pipe (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate) with both operands split into f16
// hi + lo parts and three products per contraction (hi.hi + hi.lo + lo.hi, f32 accumulation): ~21-22 significant bits,
// the precision class of the eval kernel in attention_f16.hip, whose forward this file generalises (dropout + LSE).
// Softmax state, dropout and the dS algebra stay f32.  reference: common/nets/transformer.py:269,286-302
// (nn.MultiheadAttention inside the encoder layers), forward + autograd backward.
//
// One structure, three kernels - a lane owns one row of the "resident" side in registers (hi/lo fragments) and the
// block streams 32-row tiles of the other side through LDS (double buffered), exactly like the eval forward:
//   forward : lane = query;  streams K rows / V^T          S^T = K.Q^T, O^T += V^T.P^T
//   dQ      : lane = query;  streams K rows / V rows / K^T  S^T, dP^T = V.dO^T, dQ^T += K^T.dS^T      (no atomics)
//   dK/dV   : lane = key;    streams Q rows / dO rows / Q^T / dO^T   S = Q.K^T, dP = dO.V^T, dV^T += dO^T.Pd, dK^T += Q^T.dS
// (7 GEMM-equivalents; the two extra ones are cheap here - these kernels are bound by the softmax / conversion VALU
// work, not by the MFMA pipe.)  A conversion pre-pass writes the f16 hi/lo copies both row-major [bh][Lp][64] and
// transposed [bh][64][Lp] so that every MFMA A operand is a contiguous 8- or 16-byte LDS read; Q is pre-scaled by
// log2(e)/8 (softmax in the log2 domain, same LSE convention as attention.hip).  The dropout mask is the same function
// of (seed, query, key) as in the f32 kernels.  Deterministic by construction (no atomics).
#include "common.h"
#include <hip/hip_fp16.h>

namespace hoisdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int D = 64;
constexpr int RP = 72;      // halves per row of a row-major [32][64] tile in LDS (144 B)
constexpr int TPH = 36;     // halves per row of a transposed [64][32] tile in LDS (72 B: 18 dwords, conflict-free b64 reads)
constexpr int ROWS_T = 32 * RP;     // halves per row-major tile
constexpr int TRN_T = 64 * TPH;     // halves per transposed tile
constexpr float QS2 = 0.125f * 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
// f16 keeps 11 + 11 bits in a hi + lo pair only while the lo part stays a normal number, i.e. for |x| >= 2^-3; below that
// the pair degrades to an ABSOLUTE resolution of 2^-25.  Probabilities (~1/L) and gradients are far below 2^-3, so they are
// moved up by exact powers of two before the split and the factor is taken out of the f32 result:
//   P -> P * 2^12 (max 4096);  dO -> dO * sd with sd = the power of two that brings max|dO| into [2, 4) (device scalar);
//   dS -> dS * sd * 2^5.
constexpr float PSC = 4096.f, DSC = 32.f;
#define CR(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct SplitArgs {
  // converted operands (any may be null when a kernel does not use it)
  const _Float16 *qh, *ql, *qth, *qtl;      // Q rows (pre-scaled), Q^T
  const _Float16 *kh, *kl, *kth, *ktl;      // K rows, K^T
  const _Float16 *vh, *vl, *vth, *vtl;      // V rows, V^T
  const _Float16 *dh, *dl, *dth, *dtl;      // dO rows, dO^T
  const float* lse_in; const float* delta; const float* dscale;     // dscale: device scalar sd (nullptr = 1)
  float* out; float* lse; float* dq; float* dk; float* dv;
  int ldo, ldq, ldk, ldv;
  int B, H, Lq, Lk, Lqp, Lkp, kv_len;
  float drop_p, inv_keep;
  uint32_t thresh;
  uint64_t seed;
};

__device__ __forceinline__ bool split_block(int nx, int nbh, int& tile, int& bh) {
  const int L = blockIdx.x, slot = L >> 3;        // every tile of a (b, head) on one XCD, as in attention.hip
  bh = (slot / nx) * 8 + (L & 7);
  tile = slot % nx;
  return bh < nbh;
}

__device__ __forceinline__ void split8(const float (&e)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hi[i] = (_Float16)e[i];
    lo[i] = (_Float16)(e[i] - (float)hi[i]);
  }
}

// tile copies global -> registers -> LDS.  Row-major tile: 32 rows x 64 halves, thread -> (row t>>3, 8 halves at (t&7)*8).
// Transposed tile: 64 rows (d) x 32 halves, thread -> (row t>>2, 8 halves at (t&3)*8), stored as two 8-byte halves.
__device__ __forceinline__ uint4 ld_rows(const _Float16* base, size_t row0, int tid) {
  return *reinterpret_cast<const uint4*>(base + (row0 + (tid >> 3)) * D + (tid & 7) * 8);
}
__device__ __forceinline__ void st_rows(_Float16* lds, const uint4& v, int tid) {
  *reinterpret_cast<uint4*>(lds + (tid >> 3) * RP + (tid & 7) * 8) = v;
}
__device__ __forceinline__ uint4 ld_trn(const _Float16* base, size_t Lp, size_t col0, int tid) {
  return *reinterpret_cast<const uint4*>(base + (size_t)(tid >> 2) * Lp + col0 + (tid & 3) * 8);
}
__device__ __forceinline__ void st_trn(_Float16* lds, const uint4& v, int tid) {
  uint2* p = reinterpret_cast<uint2*>(lds + (tid >> 2) * TPH + (tid & 3) * 8);
  p[0] = make_uint2(v.x, v.y);
  p[1] = make_uint2(v.z, v.w);
}
// fragment of a transposed tile for k-step jj: the 8 "k" entries held by accumulator registers 8 jj .. 8 jj + 7
__device__ __forceinline__ f16x8 frag_trn(const _Float16* tile, int row, int jj, int h) {
  const f16x4 a = *reinterpret_cast<const f16x4*>(tile + row * TPH + 16 * jj + 4 * h);
  const f16x4 b = *reinterpret_cast<const f16x4*>(tile + row * TPH + 16 * jj + 8 + 4 * h);
  return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
}  // namespace

// ---- conversion pre-pass: f32 [B][L][ld] (head slice) -> f16 hi / lo, row-major [bh][Lp][64] and transposed
// [bh][64][Lp]; rows >= L are zero.  One block per (bh, 64-row tile): a thread owns 16 consecutive d of one row (4 threads
// per row: 256-byte coalesced reads, 128-byte coalesced row-major writes); the transposed copies go through an LDS tile so
// that they leave as 32-byte pieces of 128-byte d-rows as well (a lane-per-row version ran at 1.4 TB/s).
__global__ __launch_bounds__(256) void split_convert_kernel(const float* __restrict__ src, int ld, int L, int Lp, int B,
                                                            int H, float scale, const float* __restrict__ scale_ptr,
                                                            _Float16* __restrict__ rh, _Float16* __restrict__ rl,
                                                            _Float16* __restrict__ th, _Float16* __restrict__ tl) {
  constexpr int TP = 72;
  if (scale_ptr) scale *= scale_ptr[0];                                   // halves per LDS row of the [64 d][64 rows] tile
  __shared__ __attribute__((aligned(16))) _Float16 tile[2][64 * TP];
  const int tid = threadIdx.x;
  const int nb = Lp / 64;
  const int kb = blockIdx.x % nb, bh = blockIdx.x / nb, b = bh / H, head = bh - b * H;
  const int r = tid >> 2, dc = (tid & 3) * 16;
  const int row = kb * 64 + r;
  const bool valid = row < L;
  const float* sr = src + ((size_t)b * L + (valid ? row : 0)) * ld + head * D + dc;
  _Float16 hi[16], lo[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 a = valid ? *reinterpret_cast<const float4*>(sr + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float e[4] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hi[4 * i + j] = (_Float16)e[j];
      lo[4 * i + j] = (_Float16)(e[j] - (float)hi[4 * i + j]);
    }
  }
  if (rh) {
    _Float16* oh = rh + ((size_t)bh * Lp + row) * D + dc;
    _Float16* ol = rl + ((size_t)bh * Lp + row) * D + dc;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f16x8*>(oh + 8 * i) = f16x8{hi[8 * i], hi[8 * i + 1], hi[8 * i + 2], hi[8 * i + 3], hi[8 * i + 4],
                                                    hi[8 * i + 5], hi[8 * i + 6], hi[8 * i + 7]};
      *reinterpret_cast<f16x8*>(ol + 8 * i) = f16x8{lo[8 * i], lo[8 * i + 1], lo[8 * i + 2], lo[8 * i + 3], lo[8 * i + 4],
                                                    lo[8 * i + 5], lo[8 * i + 6], lo[8 * i + 7]};
    }
  }
  if (th) {                                                // block-uniform
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      tile[0][(dc + i) * TP + r] = hi[i];
      tile[1][(dc + i) * TP + r] = lo[i];
    }
    __syncthreads();
    const int d = tid >> 2, rc = (tid & 3) * 16;           // 16 consecutive rows of d-row d
    _Float16* oh = th + ((size_t)bh * D + d) * Lp + kb * 64 + rc;
    _Float16* ol = tl + ((size_t)bh * D + d) * Lp + kb * 64 + rc;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<uint4*>(oh + 8 * i) = *reinterpret_cast<const uint4*>(&tile[0][d * TP + rc + 8 * i]);
      *reinterpret_cast<uint4*>(ol + 8 * i) = *reinterpret_cast<const uint4*>(&tile[1][d * TP + rc + 8 * i]);
    }
  }
}

// ============================================================================================
// forward: block = 128 queries (lane = query), streams 32-key tiles of K rows (hi, lo) and V^T (hi, lo)
// ============================================================================================
__global__ __launch_bounds__(256, 2) void split_fwd_kernel(SplitArgs a) {
  constexpr int BUF = 2 * ROWS_T + 2 * TRN_T;
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  int qtile, bh;
  if (!split_block((a.Lq + 127) / 128, a.B * a.H, qtile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int qrow = qtile * 128 + wave * 32 + c;
  const _Float16* kh = a.kh + (size_t)bh * a.Lkp * D;
  const _Float16* kl = a.kl + (size_t)bh * a.Lkp * D;
  const _Float16* vth = a.vth + (size_t)bh * D * a.Lkp;
  const _Float16* vtl = a.vtl + (size_t)bh * D * a.Lkp;

  f16x8 qf[4], qlo[4];                       // Q^T fragment: k-step j <-> d = 16 j + 8 h .. + 7 of the lane's query
  {
    const _Float16* p = a.qh + ((size_t)bh * a.Lqp + qrow) * D;      // qrow < Lqp always (padded with zero rows)
    const _Float16* pl = a.ql + ((size_t)bh * a.Lqp + qrow) * D;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qf[j] = *reinterpret_cast<const f16x8*>(p + 16 * j + 8 * h);
      qlo[j] = *reinterpret_cast<const f16x8*>(pl + 16 * j + 8 * h);
    }
  }
  const uint32_t rowkey = drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qrow));
  float m = -INFINITY, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

  const int ntiles = (a.kv_len + 31) / 32;
  uint4 r0, r1, r2, r3;
  auto load = [&](int kt) {
    r0 = ld_rows(kh, (size_t)kt * 32, tid);
    r1 = ld_rows(kl, (size_t)kt * 32, tid);
    r2 = ld_trn(vth, a.Lkp, (size_t)kt * 32, tid);
    r3 = ld_trn(vtl, a.Lkp, (size_t)kt * 32, tid);
  };
  auto store = [&](_Float16* buf) {
    st_rows(buf, r0, tid);
    st_rows(buf + ROWS_T, r1, tid);
    st_trn(buf + 2 * ROWS_T, r2, tid);
    st_trn(buf + 2 * ROWS_T + TRN_T, r3, tid);
  };
  load(0);
  store(lds);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ntiles) load(kt + 1);
    const _Float16* Kh = lds + cur * BUF;
    const _Float16* Kl = Kh + ROWS_T;
    const _Float16* Vh = Kh + 2 * ROWS_T;
    const _Float16* Vl = Vh + TRN_T;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16x8 kk = *reinterpret_cast<const f16x8*>(&Kh[c * RP + 16 * j + 8 * h]);
      const f16x8 k8 = *reinterpret_cast<const f16x8*>(&Kl[c * RP + 16 * j + 8 * h]);
      s = MF(k8, qf[j], s);                    // small terms first
      s = MF(kk, qlo[j], s);
      s = MF(kk, qf[j], s);
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
    const float mn = fmaxf(m, mt);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] - mn);
      ps += p;
      s[r] = p;
    }
    lsum = lsum * alpha + ps;
    m = mn;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    if (a.drop_p > 0.f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= drop_scale(rowkey, (uint32_t)(kt * 32 + CR(r, h)), a.thresh, a.inv_keep);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = s[8 * jj + i] * PSC;
      f16x8 pf, pl;
      split8(e, pf, pl);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f16x8 vv = frag_trn(Vh, dt * 32 + c, jj, h);
        const f16x8 ww = frag_trn(Vl, dt * 32 + c, jj, h);
        o[dt] = MF(ww, pf, o[dt]);
        o[dt] = MF(vv, pl, o[dt]);
        o[dt] = MF(vv, pf, o[dt]);
      }
    }
    if (kt + 1 < ntiles) store(lds + (cur ^ 1) * BUF);
    __syncthreads();
  }
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  if (qrow < a.Lq) {
    const float inv = 1.f / (ltot * PSC);
    float* op = a.out + ((size_t)b * a.Lq + qrow) * a.ldo + head * D;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(op + 32 * t + 8 * g + 4 * h) =
            make_float4(o[t][4 * g + 0] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
    if (h == 0 && a.lse) a.lse[(size_t)bh * a.Lq + qrow] = m + log2f(ltot);       // log2 domain
  }
}

// ============================================================================================
// backward, dQ: block = 128 queries (lane = query), streams K rows, V rows, K^T (hi, lo each)
// ============================================================================================
__global__ __launch_bounds__(256, 2) void split_bwd_dq_kernel(SplitArgs a) {
  constexpr int BUF = 4 * ROWS_T + 2 * TRN_T;
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  int qtile, bh;
  if (!split_block((a.Lq + 127) / 128, a.B * a.H, qtile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int qrow = qtile * 128 + wave * 32 + c;
  const bool qvalid = qrow < a.Lq;
  const _Float16* kh = a.kh + (size_t)bh * a.Lkp * D;
  const _Float16* kl = a.kl + (size_t)bh * a.Lkp * D;
  const _Float16* vh = a.vh + (size_t)bh * a.Lkp * D;
  const _Float16* vl = a.vl + (size_t)bh * a.Lkp * D;
  const _Float16* kth = a.kth + (size_t)bh * D * a.Lkp;
  const _Float16* ktl = a.ktl + (size_t)bh * D * a.Lkp;

  f16x8 qf[4], qlo[4], df[4], dlo[4];
  {
    const size_t ro = ((size_t)bh * a.Lqp + qrow) * D;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qf[j] = *reinterpret_cast<const f16x8*>(a.qh + ro + 16 * j + 8 * h);
      qlo[j] = *reinterpret_cast<const f16x8*>(a.ql + ro + 16 * j + 8 * h);
      df[j] = *reinterpret_cast<const f16x8*>(a.dh + ro + 16 * j + 8 * h);
      dlo[j] = *reinterpret_cast<const f16x8*>(a.dl + ro + 16 * j + 8 * h);
    }
  }
  const float sd = a.dscale ? a.dscale[0] : 1.f;
  const float lse = qvalid ? a.lse_in[(size_t)bh * a.Lq + qrow] : INFINITY;
  const float delta = qvalid ? a.delta[(size_t)bh * a.Lq + qrow] * sd : 0.f;
  const uint32_t rowkey = drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qrow));
  f32x16 dq[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  const int ntiles = (a.kv_len + 31) / 32;
  uint4 r0, r1, r2, r3, r4, r5;
  auto load = [&](int kt) {
    r0 = ld_rows(kh, (size_t)kt * 32, tid);
    r1 = ld_rows(kl, (size_t)kt * 32, tid);
    r2 = ld_rows(vh, (size_t)kt * 32, tid);
    r3 = ld_rows(vl, (size_t)kt * 32, tid);
    r4 = ld_trn(kth, a.Lkp, (size_t)kt * 32, tid);
    r5 = ld_trn(ktl, a.Lkp, (size_t)kt * 32, tid);
  };
  auto store = [&](_Float16* buf) {
    st_rows(buf, r0, tid);
    st_rows(buf + ROWS_T, r1, tid);
    st_rows(buf + 2 * ROWS_T, r2, tid);
    st_rows(buf + 3 * ROWS_T, r3, tid);
    st_trn(buf + 4 * ROWS_T, r4, tid);
    st_trn(buf + 4 * ROWS_T + TRN_T, r5, tid);
  };
  load(0);
  store(lds);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    const _Float16* Kh = lds + cur * BUF;
    const _Float16* Kl = Kh + ROWS_T;
    const _Float16* Vh = Kh + 2 * ROWS_T;
    const _Float16* Vl = Kh + 3 * ROWS_T;
    const _Float16* Th = Kh + 4 * ROWS_T;
    const _Float16* Tl = Th + TRN_T;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16x8 kk = *reinterpret_cast<const f16x8*>(&Kh[c * RP + 16 * j + 8 * h]);
      const f16x8 k8 = *reinterpret_cast<const f16x8*>(&Kl[c * RP + 16 * j + 8 * h]);
      const f16x8 vv = *reinterpret_cast<const f16x8*>(&Vh[c * RP + 16 * j + 8 * h]);
      const f16x8 v8 = *reinterpret_cast<const f16x8*>(&Vl[c * RP + 16 * j + 8 * h]);
      s = MF(k8, qf[j], s);    dp = MF(v8, df[j], dp);
      s = MF(kk, qlo[j], s);   dp = MF(vv, dlo[j], dp);
      s = MF(kk, qf[j], s);    dp = MF(vv, df[j], dp);
    }
    // dS^T[key][q] in place of dp
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + CR(r, h);
      const float p = key < a.kv_len ? __builtin_amdgcn_exp2f(s[r] - lse) : 0.f;
      float dsc = 1.f;
      if (a.drop_p > 0.f) dsc = drop_scale(rowkey, (uint32_t)key, a.thresh, a.inv_keep);
      dp[r] = DSC * p * (dp[r] * dsc - delta);
    }
    if (kt + 1 < ntiles) load(kt + 1);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = dp[8 * jj + i];
      f16x8 sf, sl;
      split8(e, sf, sl);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f16x8 tt = frag_trn(Th, dt * 32 + c, jj, h);
        const f16x8 t8 = frag_trn(Tl, dt * 32 + c, jj, h);
        dq[dt] = MF(t8, sf, dq[dt]);
        dq[dt] = MF(tt, sl, dq[dt]);
        dq[dt] = MF(tt, sf, dq[dt]);
      }
    }
    if (kt + 1 < ntiles) store(lds + (cur ^ 1) * BUF);
    __syncthreads();
  }
  if (qvalid) {
    const float us = 0.125f / (DSC * sd);
    float* op = a.dq + ((size_t)b * a.Lq + qrow) * a.ldq + head * D;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(op + 32 * t + 8 * g + 4 * h) =
            make_float4(dq[t][4 * g + 0] * us, dq[t][4 * g + 1] * us, dq[t][4 * g + 2] * us, dq[t][4 * g + 3] * us);
  }
}

// ============================================================================================
// backward, dK / dV: block = 128 keys (lane = key), streams Q rows, dO rows, Q^T, dO^T (hi, lo each) + lse / delta
// ============================================================================================
__global__ __launch_bounds__(256, 2) void split_bwd_dkv_kernel(SplitArgs a) {
  constexpr int BUF = 4 * ROWS_T + 4 * TRN_T;
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * BUF];
  __shared__ float stat[2][2][32];                   // [buf][lse | delta][q]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  int ktile, bh;
  if (!split_block((a.Lk + 127) / 128, a.B * a.H, ktile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int key = ktile * 128 + wave * 32 + c;
  const bool kvalid = key < a.kv_len;
  const _Float16* qh = a.qh + (size_t)bh * a.Lqp * D;
  const _Float16* ql = a.ql + (size_t)bh * a.Lqp * D;
  const _Float16* dh = a.dh + (size_t)bh * a.Lqp * D;
  const _Float16* dl = a.dl + (size_t)bh * a.Lqp * D;
  const _Float16* qth = a.qth + (size_t)bh * D * a.Lqp;
  const _Float16* qtl = a.qtl + (size_t)bh * D * a.Lqp;
  const _Float16* dth = a.dth + (size_t)bh * D * a.Lqp;
  const _Float16* dtl = a.dtl + (size_t)bh * D * a.Lqp;

  f16x8 kf[4], klo[4], vf[4], vlo[4];
  {
    const size_t ro = ((size_t)bh * a.Lkp + key) * D;          // key < Lkp always (padded with zero rows)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kf[j] = *reinterpret_cast<const f16x8*>(a.kh + ro + 16 * j + 8 * h);
      klo[j] = *reinterpret_cast<const f16x8*>(a.kl + ro + 16 * j + 8 * h);
      vf[j] = *reinterpret_cast<const f16x8*>(a.vh + ro + 16 * j + 8 * h);
      vlo[j] = *reinterpret_cast<const f16x8*>(a.vl + ro + 16 * j + 8 * h);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  const float sd = a.dscale ? a.dscale[0] : 1.f;
  const bool block_active = ktile * 128 < a.kv_len;
  const int nq = block_active ? (a.Lq + 31) / 32 : 0;
  // The eight tile pieces of the next query tile travel through FOUR staging registers in two phases (row-major pieces
  // during the S / dP products, transposed pieces during the dS algebra and the dV / dK products): a full eight-register
  // prefetch pushed the kernel over 256 VGPRs (24 spilled, 50 MB of scratch writes per launch in the PMC pass).
  uint4 r0, r1, r2, r3;
  float rl = INFINITY, re = 0.f;
  auto load_rows = [&](int qt) {
    r0 = ld_rows(qh, (size_t)qt * 32, tid);
    r1 = ld_rows(ql, (size_t)qt * 32, tid);
    r2 = ld_rows(dh, (size_t)qt * 32, tid);
    r3 = ld_rows(dl, (size_t)qt * 32, tid);
    if (tid < 32) {
      const int q = qt * 32 + tid;
      rl = q < a.Lq ? a.lse_in[(size_t)bh * a.Lq + q] : INFINITY;
      re = q < a.Lq ? a.delta[(size_t)bh * a.Lq + q] * sd : 0.f;
    }
  };
  auto store_rows = [&](int bufi) {
    _Float16* buf = lds + bufi * BUF;
    st_rows(buf, r0, tid);
    st_rows(buf + ROWS_T, r1, tid);
    st_rows(buf + 2 * ROWS_T, r2, tid);
    st_rows(buf + 3 * ROWS_T, r3, tid);
    if (tid < 32) { stat[bufi][0][tid] = rl; stat[bufi][1][tid] = re; }
  };
  auto load_trn = [&](int qt) {
    r0 = ld_trn(qth, a.Lqp, (size_t)qt * 32, tid);
    r1 = ld_trn(qtl, a.Lqp, (size_t)qt * 32, tid);
    r2 = ld_trn(dth, a.Lqp, (size_t)qt * 32, tid);
    r3 = ld_trn(dtl, a.Lqp, (size_t)qt * 32, tid);
  };
  auto store_trn = [&](int bufi) {
    _Float16* buf = lds + bufi * BUF + 4 * ROWS_T;
    st_trn(buf, r0, tid);
    st_trn(buf + TRN_T, r1, tid);
    st_trn(buf + 2 * TRN_T, r2, tid);
    st_trn(buf + 3 * TRN_T, r3, tid);
  };
  if (nq > 0) {
    load_rows(0);
    store_rows(0);
    load_trn(0);
    store_trn(0);
  }
  __syncthreads();
  for (int qt = 0; qt < nq; ++qt) {
    const int cur = qt & 1;
    const bool more = qt + 1 < nq;
    if (more) load_rows(qt + 1);
    const _Float16* Qh = lds + cur * BUF;
    const _Float16* Ql = Qh + ROWS_T;
    const _Float16* Dh = Qh + 2 * ROWS_T;
    const _Float16* Dl = Qh + 3 * ROWS_T;
    const _Float16* QTh = Qh + 4 * ROWS_T;
    const _Float16* QTl = QTh + TRN_T;
    const _Float16* DTh = QTh + 2 * TRN_T;
    const _Float16* DTl = QTh + 3 * TRN_T;
    const float* Ls = stat[cur][0];
    const float* Es = stat[cur][1];
    // S[q][key] = Qs.K^T and dP[q][key] = dO.V^T: rows = the tile's queries, column = this lane's key
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16x8 qq = *reinterpret_cast<const f16x8*>(&Qh[c * RP + 16 * j + 8 * h]);
      const f16x8 q8 = *reinterpret_cast<const f16x8*>(&Ql[c * RP + 16 * j + 8 * h]);
      const f16x8 dd = *reinterpret_cast<const f16x8*>(&Dh[c * RP + 16 * j + 8 * h]);
      const f16x8 d8 = *reinterpret_cast<const f16x8*>(&Dl[c * RP + 16 * j + 8 * h]);
      s = MF(q8, kf[j], s);    dp = MF(d8, vf[j], dp);
      s = MF(qq, klo[j], s);   dp = MF(dd, vlo[j], dp);
      s = MF(qq, kf[j], s);    dp = MF(dd, vf[j], dp);
    }
    if (more) {
      store_rows(cur ^ 1);
      load_trn(qt + 1);
    }
    // s <- Pd (dropped probabilities), dp <- dS
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = CR(r, h);
      const float p = kvalid ? __builtin_amdgcn_exp2f(s[r] - Ls[qi]) : 0.f;
      float dsc = 1.f;
      if (a.drop_p > 0.f)
        dsc = drop_scale(drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qt * 32) + (uint32_t)qi), (uint32_t)key, a.thresh,
                         a.inv_keep);
      s[r] = PSC * p * dsc;
      dp[r] = DSC * p * (dp[r] * dsc - Es[qi]);
    }
    // dV^T[d][key] += dO^T[d][q] . Pd[q][key];  dK^T[d][key] += Qs^T[d][q] . dS[q][key]
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      float e[8], g[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { e[i] = s[8 * jj + i]; g[i] = dp[8 * jj + i]; }
      f16x8 pf, pl, sf, sl;
      split8(e, pf, pl);
      split8(g, sf, sl);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f16x8 oo = frag_trn(DTh, dt * 32 + c, jj, h);
        const f16x8 o8 = frag_trn(DTl, dt * 32 + c, jj, h);
        const f16x8 tt = frag_trn(QTh, dt * 32 + c, jj, h);
        const f16x8 t8 = frag_trn(QTl, dt * 32 + c, jj, h);
        dv[dt] = MF(o8, pf, dv[dt]);   dk[dt] = MF(t8, sf, dk[dt]);
        dv[dt] = MF(oo, pl, dv[dt]);   dk[dt] = MF(tt, sl, dk[dt]);
        dv[dt] = MF(oo, pf, dv[dt]);   dk[dt] = MF(tt, sf, dk[dt]);
      }
    }
    if (more) store_trn(cur ^ 1);
    __syncthreads();
  }
  if (key < a.Lk) {
    const float uk = LN2 / (DSC * sd), uv = 1.f / (PSC * sd);
    float* pk = a.dk + ((size_t)b * a.Lk + key) * a.ldk + head * D;
    float* pv = a.dv + ((size_t)b * a.Lk + key) * a.ldv + head * D;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // Q was pre-scaled by log2(e)/8: dK = dS^T.Q / 8 = (dS^T.Qs) * ln 2; then the power-of-two factors
        *reinterpret_cast<float4*>(pk + 32 * t + 8 * g + 4 * h) =
            make_float4(dk[t][4 * g + 0] * uk, dk[t][4 * g + 1] * uk, dk[t][4 * g + 2] * uk, dk[t][4 * g + 3] * uk);
        *reinterpret_cast<float4*>(pv + 32 * t + 8 * g + 4 * h) =
            make_float4(dv[t][4 * g + 0] * uv, dv[t][4 * g + 1] * uv, dv[t][4 * g + 2] * uv, dv[t][4 * g + 3] * uv);
      }
  }
}

// delta[bh][q] = sum_d dO[q][d] * O[q][d]  (f32; one wave per 4 rows would be overkill: 16 lanes per (q, head))
__global__ __launch_bounds__(256) void split_delta_kernel(const float* __restrict__ o, int ldo,
                                                          const float* __restrict__ dout, int lddo,
                                                          float* __restrict__ delta, int B, int H, int Lq) {
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

}  // namespace hoisdf

using namespace hoisdf;

namespace {
inline long pad128(long L) { return (L + 127) / 128 * 128; }
inline size_t arr_halves(int B, int H, long Lp) { return (size_t)B * H * Lp * 64; }

int convert(const float* src, int ld, int L, int Lp, int B, int H, float scale, const float* scale_ptr, _Float16* rh,
            _Float16* rl, _Float16* th, _Float16* tl, hipStream_t st) {
  const long nblk = (long)B * H * (Lp / 64);
  hipLaunchKernelGGL(split_convert_kernel, dim3((unsigned)nblk), dim3(256), 0, st, src, ld, L, Lp, B, H, scale, scale_ptr, rh, rl,
                     th, tl);
  return check_launch("attention_split_convert");
}
}  // namespace

extern "C" long hoisdf_attention_split_workspace(int B, int H, int Lq, int Lk, int backward) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  const long Lqp = pad128(Lq), Lkp = pad128(Lk);
  // 0 forward: Q rows (2), K rows (2), V^T (2).  1 backward: Q rows + Q^T (4), K rows + K^T (4), V rows (2), dO rows + dO^T (4)
  // 2 forward that keeps every plane the backward needs: Q, K, V rows + transposed (12).  3 backward on top of a mode-2
  //   forward workspace: dO rows + dO^T (4)
  const size_t q = arr_halves(B, H, Lqp), k = arr_halves(B, H, Lkp);
  const size_t halves = backward == 0 ? 2 * q + 4 * k : backward == 1 ? 8 * q + 6 * k : backward == 2 ? 4 * q + 8 * k : 4 * q;
  return (long)(halves * sizeof(_Float16));
}

static int check_split(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, int B, int H, int Lq, int Lk,
                       int kv_len, float drop_p, const char* who) {
  HOISDF_REQUIRE(q && k && v, HOISDF_ERR_INVALID, "%s: null pointer", who);
  HOISDF_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && kv_len > 0 && kv_len <= Lk, HOISDF_ERR_INVALID,
                 "%s: bad sizes B=%d H=%d Lq=%d Lk=%d kv_len=%d", who, B, H, Lq, Lk, kv_len);
  HOISDF_REQUIRE(ldq >= H * 64 && ldk >= H * 64 && ldv >= H * 64 && ((ldq | ldk | ldv) & 3) == 0 &&
                     (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0,
                 HOISDF_ERR_INVALID, "%s: leading dims must be multiples of 4 and >= H*64, pointers 16-byte aligned", who);
  HOISDF_REQUIRE(drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID, "%s: drop_p=%f", who, drop_p);
  return HOISDF_OK;
}

// the planes of Q, K, V in the layout of a "kept" forward workspace (mode 2): [qh ql qth qtl | kh kl kth ktl | vh vl vth vtl]
struct KeptPlanes { _Float16 *qh, *ql, *qth, *qtl, *kh, *kl, *kth, *ktl, *vh, *vl, *vth, *vtl; };
static KeptPlanes kept_planes(void* workspace, int B, int H, int Lqp, int Lkp) {
  _Float16* w = reinterpret_cast<_Float16*>(workspace);
  const size_t nq = arr_halves(B, H, Lqp), nk = arr_halves(B, H, Lkp);
  KeptPlanes p;
  p.qh = w; p.ql = p.qh + nq; p.qth = p.ql + nq; p.qtl = p.qth + nq;
  p.kh = p.qtl + nq; p.kl = p.kh + nk; p.kth = p.kl + nk; p.ktl = p.kth + nk;
  p.vh = p.ktl + nk; p.vl = p.vh + nk; p.vth = p.vl + nk; p.vtl = p.vth + nk;
  return p;
}

static int attention_fwd_split_impl(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    float* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len,
                                    float drop_p, uint64_t seed, void* workspace, long workspace_bytes, int keep,
                                    void* stream) {
  if (int rc = check_split(q, k, v, ldq, ldk, ldv, B, H, Lq, Lk, kv_len, drop_p, "attention_fwd_split")) return rc;
  HOISDF_REQUIRE(o && workspace && ldo >= H * 64 && (ldo & 3) == 0 && (((uintptr_t)o | (uintptr_t)workspace) & 15) == 0,
                 HOISDF_ERR_INVALID, "attention_fwd_split: bad output / workspace");
  HOISDF_REQUIRE(workspace_bytes >= hoisdf_attention_split_workspace(B, H, Lq, Lk, keep ? 2 : 0), HOISDF_ERR_WORKSPACE,
                 "attention_fwd_split: workspace %ld < %ld bytes", workspace_bytes,
                 hoisdf_attention_split_workspace(B, H, Lq, Lk, keep ? 2 : 0));
  const int Lqp = (int)pad128(Lq), Lkp = (int)pad128(Lk);
  hipStream_t st = as_stream(stream);
  _Float16* w = reinterpret_cast<_Float16*>(workspace);
  const size_t nq = arr_halves(B, H, Lqp), nk = arr_halves(B, H, Lkp);
  _Float16 *qh = w, *ql = qh + nq, *kh = ql + nq, *kl = kh + nk, *vth = kl + nk, *vtl = vth + nk;
  if (keep) {
    // one pass per tensor writes the rows AND the transposed planes: the backward will not convert Q, K, V again
    const KeptPlanes p = kept_planes(workspace, B, H, Lqp, Lkp);
    qh = p.qh; ql = p.ql; kh = p.kh; kl = p.kl; vth = p.vth; vtl = p.vtl;
    if (int rc = convert(q, ldq, Lq, Lqp, B, H, QS2, nullptr, p.qh, p.ql, p.qth, p.qtl, st)) return rc;
    if (int rc = convert(k, ldk, Lk, Lkp, B, H, 1.f, nullptr, p.kh, p.kl, p.kth, p.ktl, st)) return rc;
    if (int rc = convert(v, ldv, Lk, Lkp, B, H, 1.f, nullptr, p.vh, p.vl, p.vth, p.vtl, st)) return rc;
  } else {
    if (int rc = convert(q, ldq, Lq, Lqp, B, H, QS2, nullptr, qh, ql, nullptr, nullptr, st)) return rc;
    if (int rc = convert(k, ldk, Lk, Lkp, B, H, 1.f, nullptr, kh, kl, nullptr, nullptr, st)) return rc;
    if (int rc = convert(v, ldv, Lk, Lkp, B, H, 1.f, nullptr, nullptr, nullptr, vth, vtl, st)) return rc;
  }
  SplitArgs a{};
  a.qh = qh; a.ql = ql; a.kh = kh; a.kl = kl; a.vth = vth; a.vtl = vtl;
  a.out = o; a.lse = lse; a.ldo = ldo;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.Lqp = Lqp; a.Lkp = Lkp; a.kv_len = kv_len;
  a.drop_p = drop_p; a.inv_keep = 1.f / (1.f - drop_p); a.thresh = drop_threshold(drop_p); a.seed = seed;
  hipLaunchKernelGGL(split_fwd_kernel, dim3(cdiv(Lq, 128) * 8 * cdiv(B * H, 8)), dim3(256), 0, st, a);
  return check_launch("attention_fwd_split");
}

extern "C" int hoisdf_attention_fwd_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                          float* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len,
                                          float drop_p, uint64_t seed, void* workspace, long workspace_bytes,
                                          void* stream) {
  return attention_fwd_split_impl(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, H, Lq, Lk, kv_len, drop_p, seed, workspace,
                                  workspace_bytes, 0, stream);
}
extern "C" int hoisdf_attention_fwd_split_keep(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                               float* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len,
                                               float drop_p, uint64_t seed, void* workspace, long workspace_bytes,
                                               void* stream) {
  return attention_fwd_split_impl(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, H, Lq, Lk, kv_len, drop_p, seed, workspace,
                                  workspace_bytes, 1, stream);
}

static int attention_bwd_split_impl(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    const float* o, int ldo, const float* dout, int lddo,
                                    const float* dout_scale, const float* lse, float* delta, float* dq,
                                    float* dk, float* dv, int B, int H, int Lq, int Lk,
                                    int kv_len, float drop_p, uint64_t seed, const void* kept, void* workspace,
                                    long workspace_bytes, void* stream) {
  if (int rc = check_split(q, k, v, ldq, ldk, ldv, B, H, Lq, Lk, kv_len, drop_p, "attention_bwd_split")) return rc;
  HOISDF_REQUIRE(o && dout && lse && delta && dq && dk && dv && workspace, HOISDF_ERR_INVALID,
                 "attention_bwd_split: null pointer");
  HOISDF_REQUIRE(ldo >= H * 64 && lddo >= H * 64 && ((ldo | lddo) & 3) == 0 &&
                     (((uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)workspace) & 15) == 0,
                 HOISDF_ERR_INVALID, "attention_bwd_split: bad leading dims / alignment");
  HOISDF_REQUIRE(workspace_bytes >= hoisdf_attention_split_workspace(B, H, Lq, Lk, kept ? 3 : 1), HOISDF_ERR_WORKSPACE,
                 "attention_bwd_split: workspace %ld < %ld bytes", workspace_bytes,
                 hoisdf_attention_split_workspace(B, H, Lq, Lk, kept ? 3 : 1));
  const int Lqp = (int)pad128(Lq), Lkp = (int)pad128(Lk);
  hipStream_t st = as_stream(stream);
  _Float16* w = reinterpret_cast<_Float16*>(workspace);
  const size_t nq = arr_halves(B, H, Lqp), nk = arr_halves(B, H, Lkp);
  _Float16 *qh, *ql, *qth, *qtl, *kh, *kl, *kth, *ktl, *vh, *vl, *dh;
  if (kept) {
    // Q, K, V planes left by hoisdf_attention_fwd_split_keep on the same q, k, v: only dO is converted here
    const KeptPlanes p = kept_planes(const_cast<void*>(kept), B, H, Lqp, Lkp);
    qh = p.qh; ql = p.ql; qth = p.qth; qtl = p.qtl; kh = p.kh; kl = p.kl; kth = p.kth; ktl = p.ktl; vh = p.vh; vl = p.vl;
    dh = w;
  } else {
    qh = w; ql = qh + nq; qth = ql + nq; qtl = qth + nq;
    kh = qtl + nq; kl = kh + nk; kth = kl + nk; ktl = kth + nk; vh = ktl + nk; vl = vh + nk;
    dh = vl + nk;
    if (int rc = convert(q, ldq, Lq, Lqp, B, H, QS2, nullptr, qh, ql, qth, qtl, st)) return rc;
    if (int rc = convert(k, ldk, Lk, Lkp, B, H, 1.f, nullptr, kh, kl, kth, ktl, st)) return rc;
    if (int rc = convert(v, ldv, Lk, Lkp, B, H, 1.f, nullptr, vh, vl, nullptr, nullptr, st)) return rc;
  }
  _Float16 *dl = dh + nq, *dth = dl + nq, *dtl = dth + nq;
  if (int rc = convert(dout, lddo, Lq, Lqp, B, H, 1.f, dout_scale, dh, dl, dth, dtl, st)) return rc;
  const long ng = (long)B * Lq * H;
  hipLaunchKernelGGL(split_delta_kernel, dim3((unsigned)((ng * 16 + 255) / 256)), dim3(256), 0, st, o, ldo, dout, lddo, delta,
                     B, H, Lq);
  if (int rc = check_launch("attention_split_delta")) return rc;
  SplitArgs a{};
  a.qh = qh; a.ql = ql; a.qth = qth; a.qtl = qtl; a.kh = kh; a.kl = kl; a.kth = kth; a.ktl = ktl;
  a.vh = vh; a.vl = vl; a.dh = dh; a.dl = dl; a.dth = dth; a.dtl = dtl;
  a.lse_in = lse; a.delta = delta; a.dscale = dout_scale; a.dq = dq; a.dk = dk; a.dv = dv; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.Lqp = Lqp; a.Lkp = Lkp; a.kv_len = kv_len;
  a.drop_p = drop_p; a.inv_keep = 1.f / (1.f - drop_p); a.thresh = drop_threshold(drop_p); a.seed = seed;
  hipLaunchKernelGGL(split_bwd_dkv_kernel, dim3(cdiv(Lk, 128) * 8 * cdiv(B * H, 8)), dim3(256), 0, st, a);
  if (int rc = check_launch("attention_bwd_split_dkv")) return rc;
  hipLaunchKernelGGL(split_bwd_dq_kernel, dim3(cdiv(Lq, 128) * 8 * cdiv(B * H, 8)), dim3(256), 0, st, a);
  return check_launch("attention_bwd_split_dq");
}

extern "C" int hoisdf_attention_bwd_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                          const float* o, int ldo, const float* dout, int lddo,
                                          const float* dout_scale, const float* lse, float* delta, float* dq,
                                          float* dk, float* dv, int B, int H, int Lq, int Lk,
                                          int kv_len, float drop_p, uint64_t seed, void* workspace, long workspace_bytes,
                                          void* stream) {
  return attention_bwd_split_impl(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, dout_scale, lse, delta, dq, dk, dv, B, H, Lq,
                                  Lk, kv_len, drop_p, seed, nullptr, workspace, workspace_bytes, stream);
}
extern "C" int hoisdf_attention_bwd_split_kept(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                               const float* o, int ldo, const float* dout, int lddo,
                                               const float* dout_scale, const float* lse, float* delta, float* dq,
                                               float* dk, float* dv, int B, int H, int Lq, int Lk,
                                               int kv_len, float drop_p, uint64_t seed, const void* fwd_workspace,
                                               void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(fwd_workspace && (((uintptr_t)fwd_workspace) & 15) == 0, HOISDF_ERR_INVALID,
                 "attention_bwd_split_kept: the forward's kept workspace is required");
  return attention_bwd_split_impl(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, dout_scale, lse, delta, dq, dk, dv, B, H, Lq,
                                  Lk, kv_len, drop_p, seed, fwd_workspace, workspace, workspace_bytes, stream);
}
