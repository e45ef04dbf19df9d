// K9/K10: streaming-softmax multi-head attention, head dim 64, exact fp32 on the gfx950 f32
// MFMA pipe (v_mfma_f32_32x32x2_f32).  Never materialises the Lq x Lk score matrix.
//
// Fragment conventions (32x32x2 f32 MFMA, wave64): lane l = 32*h + c supplies A[i=c][k=h] and
// B[k=h][j=c]; accumulator register r of lane l is C[row = (r&3) + 8*(r>>2) + 4*h][col = c].
// The order of the contraction index is free as long as A and B agree, so the kernels pair
// "whatever index the lane already holds" with a matching LDS read instead of moving
// probabilities between lanes:
//   forward  : S^T = K.Q^T  -> each lane holds one query column: row max / sum / rescale are
//              lane-local (one cross-half shuffle); O^T = V^T.P^T consumes P straight from the
//              accumulator registers (no LDS round trip, no permutes).
//   backward : dK/dV kernel keeps the wave's 32 keys in the lane dimension (S = Q.K^T), the
//              dQ kernel keeps the wave's 32 queries there (S^T = K.Q^T), so dS feeds the
//              second MFMA of each chain directly from registers.  Two kernels (7 GEMM-equivalents
//              instead of 5) instead of atomics on dQ: deterministic, no 2 GB/layer of atomics.
// LDS tiles are row-major with a 68-float pitch: ds_read_b128 of 4 consecutive d for 16 keys
// of a lane group lands on 16 distinct 4-bank slots; b32 column reads are lane-consecutive.
#include <cstdlib>
#include <stdlib.h>

#include "common.h"

namespace hoisdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DH = 64;        // head dim
constexpr int PITCH = 68;     // LDS row pitch in floats
#define CROW(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// softmax runs in the log2 domain: scores are scaled by log2(e)/sqrt(64) and exponentiated with
// the native v_exp_f32 (exp2); the saved LSE is log2-domain as well (internal to fwd/bwd).
constexpr float QSCALE2 = 0.125f * 1.4426950408889634f;
#define EXP2(x) __builtin_amdgcn_exp2f(x)

// Block -> (tile, b*H+h) with every tile of one (b, head) on the same XCD (block id % 8): K/V (fwd, dQ)
// or Q/dO (dK/dV) of that head are then fetched from HBM once into one L2 instead of into all eight
// (PMC before: 5.5x over-fetch, 50 % L2 hit rate).
__device__ __forceinline__ bool attn_block(int nx, int nbh, int& tile, int& bh) {
  const int L = blockIdx.x;
  const int slot = L >> 3;
  bh = (slot / nx) * 8 + (L & 7);
  tile = slot % nx;
  return bh < nbh;
}

struct AttnArgs {
  const float* q; const float* k; const float* v;
  const float* o; const float* dout; const float* lse_in; const float* delta_in;
  float* out; float* lse; float* dq; float* dk; float* dv;
  int ldq, ldk, ldv, ldo, lddo;
  int B, H, Lq, Lk, kv_len;
  float drop_p, inv_keep;
  uint32_t thresh;
  uint64_t seed;
};

// cooperative load of a [rows x 64] tile (row r0.., head column block) into registers:
// 256 threads x NV float4, row = (tid>>4) + 16*i, 4 floats at (tid&15)*4
template <int NV>
__device__ __forceinline__ void tile_load(float4 (&reg)[NV], const float* __restrict__ base, int ld, int r0,
                                          int rmax, int tid) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int r = r0 + (tid >> 4) + 16 * i;
    reg[i] = (r < rmax) ? *reinterpret_cast<const float4*>(base + (size_t)r * ld + (tid & 15) * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NV>
__device__ __forceinline__ void tile_store(const float4 (&reg)[NV], float* __restrict__ lds, int tid) {
#pragma unroll
  for (int i = 0; i < NV; ++i)
    *reinterpret_cast<float4*>(&lds[((tid >> 4) + 16 * i) * PITCH + (tid & 15) * 4]) = reg[i];
}

// ============================================================================================
// forward: block = 128 queries (4 waves x 32) of one (b, head); K/V tiles of 64 keys
// ============================================================================================
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 64 * PITCH];   // [buf][K|V][64][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  int qtile, bh;
  if (!attn_block((a.Lq + 127) / 128, a.B * a.H, qtile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int qrow = qtile * 128 + wave * 32 + c;
  const float* qb = a.q + (size_t)b * a.Lq * a.ldq + head * DH;
  const float* kb = a.k + (size_t)b * a.Lk * a.ldk + head * DH;
  const float* vb = a.v + (size_t)b * a.Lk * a.ldv + head * DH;

  float qf[32];
  if (qrow < a.Lq) {
    const float4* p = reinterpret_cast<const float4*>(qb + (size_t)qrow * a.ldq + 32 * h);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 t = p[i];
      qf[4 * i + 0] = t.x * QSCALE2; qf[4 * i + 1] = t.y * QSCALE2;
      qf[4 * i + 2] = t.z * QSCALE2; qf[4 * i + 3] = t.w * QSCALE2;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) qf[i] = 0.f;
  }

  const uint32_t rowkey = drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qrow));
  float m = -INFINITY, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

  const int ntiles = (a.kv_len + 63) / 64;
  float4 rk[4], rv[4];
  tile_load<4>(rk, kb, a.ldk, 0, a.kv_len, tid);
  tile_load<4>(rv, vb, a.ldv, 0, a.kv_len, tid);
  tile_store<4>(rk, lds, tid);
  tile_store<4>(rv, lds + 64 * PITCH, tid);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ntiles) {
      tile_load<4>(rk, kb, a.ldk, (kt + 1) * 64, a.kv_len, tid);
      tile_load<4>(rv, vb, a.ldv, (kt + 1) * 64, a.kv_len, tid);
    }
    const float* Ks = lds + cur * 2 * 64 * PITCH;
    const float* Vs = Ks + 64 * PITCH;

    // S^T tile (64 keys x 32 queries) as two 32-key sub-tiles
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 kk = *reinterpret_cast<const float4*>(&Ks[(t * 32 + c) * PITCH + 32 * h + 4 * c4]);
        s[t] = MFMA(kk.x, qf[4 * c4 + 0], s[t]);
        s[t] = MFMA(kk.y, qf[4 * c4 + 1], s[t]);
        s[t] = MFMA(kk.z, qf[4 * c4 + 2], s[t]);
        s[t] = MFMA(kk.w, qf[4 * c4 + 3], s[t]);
      }
    }
    // online softmax for this lane's query column
    if (kt == ntiles - 1) {               // only the last tile can hold keys >= kv_len
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 64 + t * 32 + CROW(r, h) >= a.kv_len) s[t][r] = -INFINITY;
    }
    float mt = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[t][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float alpha = EXP2(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = EXP2(s[t][r] - mn);
        ps += p;
        s[t][r] = p;
      }
    lsum = lsum * alpha + ps;
    m = mn;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    if (a.drop_p > 0.f) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s[t][r] *= drop_scale(rowkey, (uint32_t)(kt * 64 + t * 32 + CROW(r, h)), a.thresh, a.inv_keep);
    }
    // O^T += V^T . P^T : A = V[key][d] (d along lanes), B = P registers
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vr = &Vs[(t * 32 + CROW(r, h)) * PITCH + c];
        o[0] = MFMA(vr[0], s[t][r], o[0]);
        o[1] = MFMA(vr[32], s[t][r], o[1]);
      }
    if (kt + 1 < ntiles) {
      float* nb = lds + (cur ^ 1) * 2 * 64 * PITCH;
      tile_store<4>(rk, nb, tid);
      tile_store<4>(rv, nb + 64 * PITCH, tid);
    }
    __syncthreads();
  }

  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  if (qrow < a.Lq) {
    const float inv = 1.f / ltot;
    float* op = a.out + ((size_t)b * a.Lq + qrow) * a.ldo + head * DH;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 w = make_float4(o[t][4 * g + 0] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv,
                               o[t][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + 32 * t + 8 * g + 4 * h) = w;
      }
    if (h == 0 && a.lse) a.lse[(size_t)bh * a.Lq + qrow] = m + log2f(ltot);     // log2 domain
  }
}

// ============================================================================================
// forward, few queries (Lq <= 32: the 17 MANO queries attending 1536 hand keys, K10): the 128-query kernel would run
// one mostly empty tile per (b, head) and walk its 24 key tiles serially (130 us, latency bound).  Here one block of
// FEWQ_WAVES waves owns the (b, head); wave w takes the 32-key tiles w, w + FEWQ_WAVES, ... straight from global
// memory (no LDS staging: every tile is read exactly once), keeps its own streaming-softmax state, and the waves'
// (max, sum, O) states are merged through LDS at the end.  Same dropout hash / lse convention as attn_fwd_kernel,
// so the fused backward applies unchanged.
// ============================================================================================
// Round 6: with few (b, head) pairs and many keys (configs[3]: 64 blocks x 96 key tiles, 105 us) the key tiles are also cut over
// `nsplit` blocks per (b, head) (blockIdx.y); each leaves its merged (max, sum, O) state in `part` ([bh][split][64 + 2][32] floats) and
// attn_fewq_merge_kernel folds the splits in order.  nsplit = 1: the block writes the output itself, as before.
constexpr int FEWQ_WAVES = 8;
__global__ __launch_bounds__(64 * FEWQ_WAVES) void attn_fwd_fewq_kernel(AttnArgs a, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float Os[FEWQ_WAVES][DH][33];   // O^T partials [d][q]
  __shared__ float Ms[FEWQ_WAVES][32], Lsum[FEWQ_WAVES][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int bh = blockIdx.x, b = bh / a.H, head = bh - b * a.H;
  const int qrow = c;
  const float* kb = a.k + (size_t)b * a.Lk * a.ldk + head * DH;
  const float* vb = a.v + (size_t)b * a.Lk * a.ldv + head * DH;
  float qf[32];
  if (qrow < a.Lq) {
    const float4* p = reinterpret_cast<const float4*>(a.q + ((size_t)b * a.Lq + qrow) * a.ldq + head * DH + 32 * h);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 t = p[i];
      qf[4 * i + 0] = t.x * QSCALE2; qf[4 * i + 1] = t.y * QSCALE2;
      qf[4 * i + 2] = t.z * QSCALE2; qf[4 * i + 3] = t.w * QSCALE2;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) qf[i] = 0.f;
  }
  const uint32_t rowkey = drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qrow));
  float m = -INFINITY, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

  const int ntiles_all = (a.kv_len + 31) / 32;
  const int nsplit = gridDim.y, per_split = (ntiles_all + nsplit - 1) / nsplit;
  const int kt0 = blockIdx.y * per_split, ntiles = min(ntiles_all, kt0 + per_split);
  for (int kt = kt0 + wave; kt < ntiles; kt += FEWQ_WAVES) {
    const int key = kt * 32 + c;                       // this lane's K row (A operand of S^T = K.Q^T)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
      const float4* kp = reinterpret_cast<const float4*>(kb + (size_t)(key < a.kv_len ? key : 0) * a.ldk + 32 * h);
      float4 kk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) kk[i] = kp[i];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s = MFMA(kk[i].x, qf[4 * i + 0], s); s = MFMA(kk[i].y, qf[4 * i + 1], s);
        s = MFMA(kk[i].z, qf[4 * i + 2], s); s = MFMA(kk[i].w, qf[4 * i + 3], s);
      }
    }
    // V rows of the 16 keys this lane's accumulator registers hold (A operand of O^T = V^T.P^T: d along lanes)
    float v0[16], v1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = kt * 32 + CROW(r, h);
      const float* vr = vb + (size_t)(kr < a.kv_len ? kr : 0) * a.ldv + c;
      v0[r] = vr[0]; v1[r] = vr[32];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kt * 32 + CROW(r, h) >= a.kv_len) s[r] = -INFINITY;
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float alpha = EXP2(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = EXP2(s[r] - mn);
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
      for (int r = 0; r < 16; ++r)
        s[r] *= drop_scale(rowkey, (uint32_t)(kt * 32 + CROW(r, h)), a.thresh, a.inv_keep);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[0] = MFMA(v0[r], s[r], o[0]);
      o[1] = MFMA(v1[r], s[r], o[1]);
    }
  }
  // merge the waves: every lane holds column q = c of O^T, rows d = 32 t + CROW(r, h); m / lsum are per q (per half)
  const float lw = lsum + __shfl_xor(lsum, 32, 64);
  if (h == 0) { Ms[wave][c] = m; Lsum[wave][c] = lw; }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) Os[wave][32 * t + CROW(r, h)][c] = o[t][r];
  __syncthreads();
  for (int e = tid; e < DH * 32; e += 64 * FEWQ_WAVES) {
    const int d = e >> 5, q = e & 31;
    if (q >= a.Lq) continue;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < FEWQ_WAVES; ++w) M = fmaxf(M, Ms[w][q]);
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < FEWQ_WAVES; ++w) {
      const float sc = Ms[w][q] == -INFINITY ? 0.f : EXP2(Ms[w][q] - M);
      L += Lsum[w][q] * sc;
      acc += Os[w][d][q] * sc;
    }
    if (nsplit == 1) {
      a.out[((size_t)b * a.Lq + q) * a.ldo + head * DH + d] = acc / L;
      if (d == 0 && a.lse) a.lse[(size_t)bh * a.Lq + q] = M + log2f(L);
    } else {
      float* ps = part + ((size_t)bh * nsplit + blockIdx.y) * (DH + 2) * 32;
      ps[d * 32 + q] = acc;
      if (d == 0) { ps[DH * 32 + q] = M; ps[(DH + 1) * 32 + q] = L; }
    }
  }
}
__global__ __launch_bounds__(256) void attn_fewq_merge_kernel(AttnArgs a, const float* __restrict__ part, int nsplit) {
  const int bh = blockIdx.x, b = bh / a.H, head = bh - b * a.H;
  const float* ps = part + (size_t)bh * nsplit * (DH + 2) * 32;
  for (int e = threadIdx.x; e < DH * 32; e += 256) {
    const int d = e >> 5, q = e & 31;
    if (q >= a.Lq) continue;
    float M = -INFINITY;
    for (int g = 0; g < nsplit; ++g) M = fmaxf(M, ps[(size_t)g * (DH + 2) * 32 + DH * 32 + q]);
    float L = 0.f, acc = 0.f;
    for (int g = 0; g < nsplit; ++g) {
      const float* pg = ps + (size_t)g * (DH + 2) * 32;
      const float mg = pg[DH * 32 + q];
      const float sc = mg == -INFINITY ? 0.f : EXP2(mg - M);
      L += pg[(DH + 1) * 32 + q] * sc;
      acc += pg[d * 32 + q] * sc;
    }
    a.out[((size_t)b * a.Lq + q) * a.ldo + head * DH + d] = acc / L;
    if (d == 0 && a.lse) a.lse[(size_t)bh * a.Lq + q] = M + log2f(L);
  }
}

// delta[bh][q] = sum_d dO[q][d] * O[q][d]  (one 16-lane group per (q, head))
// zero_dq: also clears this (b, q, head) slice of dq (the fused backward accumulates dq with atomics; clearing
// it here costs one extra 16-byte store per lane instead of a strided 2-D memset of 85 us)
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs a, float* __restrict__ delta, int zero_dq) {
  const long idx = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;      // (b, q, head)
  const int sub = threadIdx.x & 15;
  const long total = (long)a.B * a.Lq * a.H;
  float s = 0.f;
  int b = 0, q = 0, head = 0;
  if (idx < total) {
    head = (int)(idx % a.H);
    const long bq = idx / a.H;
    q = (int)(bq % a.Lq);
    b = (int)(bq / a.Lq);
    const float4 x = *reinterpret_cast<const float4*>(a.o + ((size_t)b * a.Lq + q) * a.ldo + head * DH + sub * 4);
    const float4 y = *reinterpret_cast<const float4*>(a.dout + ((size_t)b * a.Lq + q) * a.lddo + head * DH + sub * 4);
    s = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    if (zero_dq)
      *reinterpret_cast<float4*>(a.dq + ((size_t)b * a.Lq + q) * a.ldq + head * DH + sub * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (idx < total && sub == 0) delta[((size_t)b * a.H + head) * a.Lq + q] = s;
}

// ============================================================================================
// backward, dK / dV: block = 128 keys (4 waves x 32) of one (b, head); loops over 32-query tiles.
// K fragment in registers, V fragment in a per-wave LDS slab (keeps the kernel at 2 waves/SIMD
// without spills); the next Q / dO tile is prefetched into registers during the 128 MFMAs.
// ============================================================================================
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 32 * PITCH + 64 + 4 * 32 * PITCH];
  float* Qs = lds;                          // [32][PITCH] query tile
  float* Ds = lds + 32 * PITCH;             // [32][PITCH] dO tile
  float* Ls = lds + 2 * 32 * PITCH;         // lse[32]
  float* Es = Ls + 32;                      // delta[32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Vw = Es + 32 + wave * 32 * PITCH;  // this wave's V rows [32][PITCH]
  const int h = lane >> 5, c = lane & 31;
  int ktile, bh;
  if (!attn_block((a.Lk + 127) / 128, a.B * a.H, ktile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int key = ktile * 128 + wave * 32 + c;
  const bool kvalid = key < a.kv_len;
  const float* qb = a.q + (size_t)b * a.Lq * a.ldq + head * DH;
  const float* dob = a.dout + (size_t)b * a.Lq * a.lddo + head * DH;

  float kf[32];
  {
    float4 t[8], u[8];
    if (kvalid) {
      const float4* pk = reinterpret_cast<const float4*>(a.k + ((size_t)b * a.Lk + key) * a.ldk + head * DH + 32 * h);
      const float4* pv = reinterpret_cast<const float4*>(a.v + ((size_t)b * a.Lk + key) * a.ldv + head * DH + 32 * h);
#pragma unroll
      for (int i = 0; i < 8; ++i) { t[i] = pk[i]; u[i] = pv[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { t[i] = make_float4(0, 0, 0, 0); u[i] = make_float4(0, 0, 0, 0); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      kf[4 * i + 0] = t[i].x; kf[4 * i + 1] = t[i].y; kf[4 * i + 2] = t[i].z; kf[4 * i + 3] = t[i].w;
      *reinterpret_cast<float4*>(&Vw[c * PITCH + 32 * h + 4 * i]) = u[i];
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  const bool block_active = ktile * 128 < a.kv_len;
  const int nq = block_active ? (a.Lq + 31) / 32 : 0;
  float4 rq[2], rd[2];
  float rl = INFINITY, re = 0.f;
  if (nq > 0) {
    tile_load<2>(rq, qb, a.ldq, 0, a.Lq, tid);
    tile_load<2>(rd, dob, a.lddo, 0, a.Lq, tid);
    if (tid < 32 && tid < a.Lq) { rl = a.lse_in[(size_t)bh * a.Lq + tid]; re = a.delta_in[(size_t)bh * a.Lq + tid]; }
  }
  for (int qt = 0; qt < nq; ++qt) {
    __syncthreads();                       // previous tile's readers are done (also orders the V slab writes)
    tile_store<2>(rq, Qs, tid);
    tile_store<2>(rd, Ds, tid);
    if (tid < 32) { Ls[tid] = rl; Es[tid] = re; }
    __syncthreads();
    if (qt + 1 < nq) {                     // prefetch the next tile; lands during the MFMAs below
      tile_load<2>(rq, qb, a.ldq, (qt + 1) * 32, a.Lq, tid);
      tile_load<2>(rd, dob, a.lddo, (qt + 1) * 32, a.Lq, tid);
      if (tid < 32) {
        const int q = (qt + 1) * 32 + tid;
        rl = q < a.Lq ? a.lse_in[(size_t)bh * a.Lq + q] : INFINITY;
        re = q < a.Lq ? a.delta_in[(size_t)bh * a.Lq + q] : 0.f;
      }
    }
    // S = Q.K^T and dP = dO.V^T  (rows = queries, cols = this lane's key)
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const float4 qq = *reinterpret_cast<const float4*>(&Qs[c * PITCH + 32 * h + 4 * c4]);
      const float4 dd = *reinterpret_cast<const float4*>(&Ds[c * PITCH + 32 * h + 4 * c4]);
      const float4 vv = *reinterpret_cast<const float4*>(&Vw[c * PITCH + 32 * h + 4 * c4]);
      s = MFMA(qq.x, kf[4 * c4 + 0], s);  dp = MFMA(dd.x, vv.x, dp);
      s = MFMA(qq.y, kf[4 * c4 + 1], s);  dp = MFMA(dd.y, vv.y, dp);
      s = MFMA(qq.z, kf[4 * c4 + 2], s);  dp = MFMA(dd.z, vv.z, dp);
      s = MFMA(qq.w, kf[4 * c4 + 3], s);  dp = MFMA(dd.w, vv.w, dp);
    }
    // P (dropped) and dS, in place: s <- Pd, dp <- dS
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = CROW(r, h);
      const float p = kvalid ? EXP2(s[r] * QSCALE2 - Ls[ql]) : 0.f;
      float dscale = 1.f;
      if (a.drop_p > 0.f)
        dscale = drop_scale(drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qt * 32) + (uint32_t)ql), (uint32_t)key,
                            a.thresh, a.inv_keep);
      s[r] = p * dscale;
      dp[r] = p * (dp[r] * dscale - Es[ql]);
    }
    // dV^T += dO^T . Pd ; dK^T += Q^T . dS   (A from LDS with d along lanes, B from registers)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* dr = &Ds[CROW(r, h) * PITCH + c];
      const float* qr = &Qs[CROW(r, h) * PITCH + c];
      dv[0] = MFMA(dr[0], s[r], dv[0]);
      dv[1] = MFMA(dr[32], s[r], dv[1]);
      dk[0] = MFMA(qr[0], dp[r], dk[0]);
      dk[1] = MFMA(qr[32], dp[r], dk[1]);
    }
  }
  if (key < a.Lk) {
    float* pk = a.dk + ((size_t)b * a.Lk + key) * a.ldk + head * DH;
    float* pv = a.dv + ((size_t)b * a.Lk + key) * a.ldv + head * DH;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(pk + 32 * t + 8 * g + 4 * h) =
            make_float4(dk[t][4 * g + 0] * 0.125f, dk[t][4 * g + 1] * 0.125f, dk[t][4 * g + 2] * 0.125f,
                        dk[t][4 * g + 3] * 0.125f);
        *reinterpret_cast<float4*>(pv + 32 * t + 8 * g + 4 * h) =
            make_float4(dv[t][4 * g + 0], dv[t][4 * g + 1], dv[t][4 * g + 2], dv[t][4 * g + 3]);
      }
  }
}

// ============================================================================================
// backward, fused dK / dV / dQ (5 GEMM-equivalents instead of 7): same key-owner decomposition as the
// dK/dV kernel; in addition every wave drops its dS block into a shared LDS tile T[32 q][128 keys] and,
// after one more barrier, each wave contracts T with the block's K slab over all 128 keys for its own
// 16-wide slice of d (two 16x16x4 MFMA tiles), so the 32 x 64 dQ tile of the block is complete in
// registers and leaves with one float atomic per element (dq pre-zeroed by the host entry; measured
// 0.1 ms per 2048-token layer).  K lives in an LDS slab (B operand of S and of dQ), V in registers.
// LDS 69.4 KB -> 2 workgroups / CU.  The dQ summation order over key blocks is not fixed (atomics).
// (An LDS-atomic reduction of per-wave partial dQ tiles was 2.5x slower: ds_add_f32 ~0.4 lanes/clk.)
// ============================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TP = 132;       // pitch of the block's dS tile T[32 q][128 keys] (= 4 mod 32: 2 lanes / bank)
__global__ __launch_bounds__(256, 2) void attn_bwd_fused_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 32 * PITCH + 64 + 128 * PITCH + 32 * TP];
  float* Qs = lds;                          // [32][PITCH] query tile
  float* Ds = lds + 32 * PITCH;             // [32][PITCH] dO tile
  float* Ls = lds + 2 * 32 * PITCH;         // lse[32]
  float* Es = Ls + 32;                      // delta[32]
  float* Kall = Es + 32;                    // [128][PITCH] this block's keys
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Kw = Kall + wave * 32 * PITCH;     // this wave's K rows
  float* Ts = Kall + 128 * PITCH;           // [32 q][TP] dS of the whole block
  const int h = lane >> 5, c = lane & 31;
  int ktile, bh;
  if (!attn_block((a.Lk + 127) / 128, a.B * a.H, ktile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int key = ktile * 128 + wave * 32 + c;
  const bool kvalid = key < a.kv_len;
  const float* qb = a.q + (size_t)b * a.Lq * a.ldq + head * DH;
  const float* dob = a.dout + (size_t)b * a.Lq * a.lddo + head * DH;
  float* dqb = a.dq + (size_t)b * a.Lq * a.ldq + head * DH;

  float vf[32];
  {
    float4 t[8], u[8];
    if (kvalid) {
      const float4* pk = reinterpret_cast<const float4*>(a.k + ((size_t)b * a.Lk + key) * a.ldk + head * DH + 32 * h);
      const float4* pv = reinterpret_cast<const float4*>(a.v + ((size_t)b * a.Lk + key) * a.ldv + head * DH + 32 * h);
#pragma unroll
      for (int i = 0; i < 8; ++i) { t[i] = pk[i]; u[i] = pv[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { t[i] = make_float4(0, 0, 0, 0); u[i] = make_float4(0, 0, 0, 0); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      vf[4 * i + 0] = u[i].x; vf[4 * i + 1] = u[i].y; vf[4 * i + 2] = u[i].z; vf[4 * i + 3] = u[i].w;
      *reinterpret_cast<float4*>(&Kw[c * PITCH + 32 * h + 4 * i]) = t[i];
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  const bool block_active = ktile * 128 < a.kv_len;
  const int nq = block_active ? (a.Lq + 31) / 32 : 0;
  float4 rq[2], rd[2];
  float rl = INFINITY, re = 0.f;
  if (nq > 0) {
    tile_load<2>(rq, qb, a.ldq, 0, a.Lq, tid);
    tile_load<2>(rd, dob, a.lddo, 0, a.Lq, tid);
    if (tid < 32 && tid < a.Lq) { rl = a.lse_in[(size_t)bh * a.Lq + tid]; re = a.delta_in[(size_t)bh * a.Lq + tid]; }
  }
  for (int qt = 0; qt < nq; ++qt) {
    __syncthreads();                       // previous tile's readers are done (also orders the K slab / R writes)
    tile_store<2>(rq, Qs, tid);
    tile_store<2>(rd, Ds, tid);
    if (tid < 32) { Ls[tid] = rl; Es[tid] = re; }
    __syncthreads();
    // S = Q.K^T and dP = dO.V^T  (rows = queries, cols = this lane's key)
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const float4 qq = *reinterpret_cast<const float4*>(&Qs[c * PITCH + 32 * h + 4 * c4]);
      const float4 dd = *reinterpret_cast<const float4*>(&Ds[c * PITCH + 32 * h + 4 * c4]);
      const float4 kk = *reinterpret_cast<const float4*>(&Kw[c * PITCH + 32 * h + 4 * c4]);
      s = MFMA(qq.x, kk.x, s);  dp = MFMA(dd.x, vf[4 * c4 + 0], dp);
      s = MFMA(qq.y, kk.y, s);  dp = MFMA(dd.y, vf[4 * c4 + 1], dp);
      s = MFMA(qq.z, kk.z, s);  dp = MFMA(dd.z, vf[4 * c4 + 2], dp);
      s = MFMA(qq.w, kk.w, s);  dp = MFMA(dd.w, vf[4 * c4 + 3], dp);
    }
    // P (dropped) and dS, in place: s <- Pd, dp <- dS
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = CROW(r, h);
      const float p = kvalid ? EXP2(s[r] * QSCALE2 - Ls[ql]) : 0.f;
      float dscale = 1.f;
      if (a.drop_p > 0.f)
        dscale = drop_scale(drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qt * 32) + (uint32_t)ql), (uint32_t)key,
                            a.thresh, a.inv_keep);
      s[r] = p * dscale;
      dp[r] = p * (dp[r] * dscale - Es[ql]);
    }
    if (qt + 1 < nq) {                     // prefetch the next tile; issued after the S/dP phase (register peak), lands during the 96 MFMAs below
      tile_load<2>(rq, qb, a.ldq, (qt + 1) * 32, a.Lq, tid);
      tile_load<2>(rd, dob, a.lddo, (qt + 1) * 32, a.Lq, tid);
      if (tid < 32) {
        const int q = (qt + 1) * 32 + tid;
        rl = q < a.Lq ? a.lse_in[(size_t)bh * a.Lq + q] : INFINITY;
        re = q < a.Lq ? a.delta_in[(size_t)bh * a.Lq + q] : 0.f;
      }
    }
    // dV^T += dO^T . Pd ; dK^T += Q^T . dS   (A from LDS with d along lanes, B from registers)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* dr = &Ds[CROW(r, h) * PITCH + c];
      const float* qr = &Qs[CROW(r, h) * PITCH + c];
      dv[0] = MFMA(dr[0], s[r], dv[0]);
      dv[1] = MFMA(dr[32], s[r], dv[1]);
      dk[0] = MFMA(qr[0], dp[r], dk[0]);
      dk[1] = MFMA(qr[32], dp[r], dk[1]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Ts[CROW(r, h) * TP + wave * 32 + c] = dp[r];     // dS[q][key] of this wave's keys
    __syncthreads();
    // dQ[q][d0..d0+15] = sum over the block's 128 keys of dS[q][key] K[key][d]  (16x16x4 MFMA: lane l supplies
    // A[row l%16][k l/16] and B[k l/16][col l%16], holds C[rows 4*(l/16)..+3][col l%16])
    {
      const int l16 = lane & 15, kq = lane >> 4;
      f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
      const float* tp = Ts + l16 * TP + kq;
      const float* kp = Kall + kq * PITCH + wave * 16 + l16;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) {
        const float bv = kp[4 * j * PITCH];
        q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(tp[4 * j], bv, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(tp[16 * TP + 4 * j], bv, q1, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int qa = qt * 32 + 4 * kq + i, qb2 = qa + 16;
        if (qa < a.Lq) atomicAdd(dqb + (size_t)qa * a.ldq + wave * 16 + l16, q0[i] * 0.125f);
        if (qb2 < a.Lq) atomicAdd(dqb + (size_t)qb2 * a.ldq + wave * 16 + l16, q1[i] * 0.125f);
      }
    }
  }
  if (key < a.Lk) {
    float* pk = a.dk + ((size_t)b * a.Lk + key) * a.ldk + head * DH;
    float* pv = a.dv + ((size_t)b * a.Lk + key) * a.ldv + head * DH;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(pk + 32 * t + 8 * g + 4 * h) =
            make_float4(dk[t][4 * g + 0] * 0.125f, dk[t][4 * g + 1] * 0.125f, dk[t][4 * g + 2] * 0.125f,
                        dk[t][4 * g + 3] * 0.125f);
        *reinterpret_cast<float4*>(pv + 32 * t + 8 * g + 4 * h) =
            make_float4(dv[t][4 * g + 0], dv[t][4 * g + 1], dv[t][4 * g + 2], dv[t][4 * g + 3]);
      }
  }
}

// ============================================================================================
// backward, dQ: block = 128 queries (4 waves x 32); loops over 32-key tiles (prefetched)
// ============================================================================================
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 32 * PITCH];   // K tile, V tile
  float* Ks = lds;
  float* Vs = lds + 32 * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  int qtile, bh;
  if (!attn_block((a.Lq + 127) / 128, a.B * a.H, qtile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int qrow = qtile * 128 + wave * 32 + c;
  const bool qvalid = qrow < a.Lq;
  const float* kb = a.k + (size_t)b * a.Lk * a.ldk + head * DH;
  const float* vb = a.v + (size_t)b * a.Lk * a.ldv + head * DH;

  float qf[32], df[32];
  float lse = INFINITY, delta = 0.f;
  if (qvalid) {
    const float4* pq = reinterpret_cast<const float4*>(a.q + ((size_t)b * a.Lq + qrow) * a.ldq + head * DH + 32 * h);
    const float4* pd = reinterpret_cast<const float4*>(a.dout + ((size_t)b * a.Lq + qrow) * a.lddo + head * DH + 32 * h);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 t = pq[i], u = pd[i];
      qf[4 * i + 0] = t.x; qf[4 * i + 1] = t.y; qf[4 * i + 2] = t.z; qf[4 * i + 3] = t.w;
      df[4 * i + 0] = u.x; df[4 * i + 1] = u.y; df[4 * i + 2] = u.z; df[4 * i + 3] = u.w;
    }
    lse = a.lse_in[(size_t)bh * a.Lq + qrow];
    delta = a.delta_in[(size_t)bh * a.Lq + qrow];
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) { qf[i] = 0.f; df[i] = 0.f; }
  }
  const uint32_t rowkey = drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + qrow));
  f32x16 dq[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  const int nk = (a.kv_len + 31) / 32;
  float4 rk[2], rv[2];
  tile_load<2>(rk, kb, a.ldk, 0, a.kv_len, tid);
  tile_load<2>(rv, vb, a.ldv, 0, a.kv_len, tid);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    tile_store<2>(rk, Ks, tid);
    tile_store<2>(rv, Vs, tid);
    __syncthreads();
    if (kt + 1 < nk) {
      tile_load<2>(rk, kb, a.ldk, (kt + 1) * 32, a.kv_len, tid);
      tile_load<2>(rv, vb, a.ldv, (kt + 1) * 32, a.kv_len, tid);
    }
    // S^T = K.Q^T, dP^T = V.dO^T  (rows = keys, cols = this lane's query)
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const float4 kk = *reinterpret_cast<const float4*>(&Ks[c * PITCH + 32 * h + 4 * c4]);
      const float4 vv = *reinterpret_cast<const float4*>(&Vs[c * PITCH + 32 * h + 4 * c4]);
      s = MFMA(kk.x, qf[4 * c4 + 0], s);  dp = MFMA(vv.x, df[4 * c4 + 0], dp);
      s = MFMA(kk.y, qf[4 * c4 + 1], s);  dp = MFMA(vv.y, df[4 * c4 + 1], dp);
      s = MFMA(kk.z, qf[4 * c4 + 2], s);  dp = MFMA(vv.z, df[4 * c4 + 2], dp);
      s = MFMA(kk.w, qf[4 * c4 + 3], s);  dp = MFMA(vv.w, df[4 * c4 + 3], dp);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + CROW(r, h);
      const float p = (key < a.kv_len) ? EXP2(s[r] * QSCALE2 - lse) : 0.f;
      float dscale = 1.f;
      if (a.drop_p > 0.f) dscale = drop_scale(rowkey, (uint32_t)key, a.thresh, a.inv_keep);
      dp[r] = p * (dp[r] * dscale - delta);
    }
    // dQ^T += K^T . dS^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* kr = &Ks[CROW(r, h) * PITCH + c];
      dq[0] = MFMA(kr[0], dp[r], dq[0]);
      dq[1] = MFMA(kr[32], dp[r], dq[1]);
    }
  }
  if (qvalid) {
    float* pq = a.dq + ((size_t)b * a.Lq + qrow) * a.ldq + head * DH;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(pq + 32 * t + 8 * g + 4 * h) =
            make_float4(dq[t][4 * g + 0] * 0.125f, dq[t][4 * g + 1] * 0.125f, dq[t][4 * g + 2] * 0.125f,
                        dq[t][4 * g + 3] * 0.125f);
  }
}

// ============================================================================================
// small masked attention (Lq, Lk <= 64): one wave per (b, head, query); probs saved
// ============================================================================================
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(AttnArgs a, const uint8_t* __restrict__ mask,
                                                             float* __restrict__ probs) {
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long total = (long)a.B * a.H * a.Lq;
  if (w >= total) return;
  const int qi = (int)(w % a.Lq);
  const int bh = (int)(w / a.Lq), b = bh / a.H, head = bh - b * a.H;
  const float* qp = a.q + ((size_t)b * a.Lq + qi) * a.ldq + head * DH;
  float sc = -INFINITY;
  if (lane < a.Lk && !(mask && mask[qi * a.Lk + lane])) {
    const float* kp = a.k + ((size_t)b * a.Lk + lane) * a.ldk + head * DH;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < DH; ++d) s += (qp[d] * 0.125f) * kp[d];
    sc = s;
  }
  const float mx = wave_max(sc);
  const float e = (sc == -INFINITY) ? 0.f : expf(sc - mx);
  const float p = e / wave_sum(e);
  if (lane < a.Lk) probs[(size_t)w * a.Lk + lane] = p;
  float pd = p;
  if (a.drop_p > 0.f && lane < a.Lk) pd *= drop_scale(drop_rowkey(a.seed, (uint32_t)w), (uint32_t)lane, a.thresh, a.inv_keep);
  // out[d = lane] = sum_j pd_j V[j][d]
  float acc = 0.f;
  for (int j = 0; j < a.Lk; ++j) {
    const float pj = __shfl(pd, j, 64);
    acc += pj * a.v[((size_t)b * a.Lk + j) * a.ldv + head * DH + lane];
  }
  a.out[((size_t)b * a.Lq + qi) * a.ldo + head * DH + lane] = acc;
}

// dq written directly (one wave owns a query); dk/dv accumulated with atomics (pre-zeroed)
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(AttnArgs a, const float* __restrict__ probs) {
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long total = (long)a.B * a.H * a.Lq;
  if (w >= total) return;
  const int qi = (int)(w % a.Lq);
  const int bh = (int)(w / a.Lq), b = bh / a.H, head = bh - b * a.H;
  const float* dop = a.dout + ((size_t)b * a.Lq + qi) * a.lddo + head * DH;
  const float p = lane < a.Lk ? probs[(size_t)w * a.Lk + lane] : 0.f;
  float dsc = 1.f;
  if (a.drop_p > 0.f && lane < a.Lk) dsc = drop_scale(drop_rowkey(a.seed, (uint32_t)w), (uint32_t)lane, a.thresh, a.inv_keep);
  // dP_j = sum_d dO[d] V[j][d]  (lane = key j)
  float dpj = 0.f;
  if (lane < a.Lk) {
    const float* vp = a.v + ((size_t)b * a.Lk + lane) * a.ldv + head * DH;
#pragma unroll 8
    for (int d = 0; d < DH; ++d) dpj += dop[d] * vp[d];
  }
  dpj *= dsc;
  const float dot = wave_sum(p * dpj);
  const float dsj = p * (dpj - dot);               // d score_j
  const float pdj = p * dsc;
  // lane = d from here on
  const float dod = dop[lane];
  const float qd = a.q[((size_t)b * a.Lq + qi) * a.ldq + head * DH + lane];
  float dqd = 0.f;
  for (int j = 0; j < a.Lk; ++j) {
    const float ds = __shfl(dsj, j, 64);
    const float pj = __shfl(pdj, j, 64);
    const size_t krow = ((size_t)b * a.Lk + j);
    dqd += ds * a.k[krow * a.ldk + head * DH + lane];
    atomicAdd(&a.dk[krow * a.ldk + head * DH + lane], ds * qd * 0.125f);
    atomicAdd(&a.dv[krow * a.ldv + head * DH + lane], pj * dod);
  }
  a.dq[((size_t)b * a.Lq + qi) * a.ldq + head * DH + lane] = dqd * 0.125f;
}

// Deterministic form of the small backward: one wave per (sample, head) walks the queries in order, so dk / dv have a
// single writer (plain read-modify-write in query order); Lk <= 64 keys, lane = d.
__global__ __launch_bounds__(256) void attn_small_bwd_det_kernel(AttnArgs a, const float* __restrict__ probs) {
  const int lane = threadIdx.x & 63;
  const long bhl = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bhl >= (long)a.B * a.H) return;
  const int bh = (int)bhl, b = bh / a.H, head = bh - b * a.H;
  for (int qi = 0; qi < a.Lq; ++qi) {
    const long w = (long)bh * a.Lq + qi;
    const float* dop = a.dout + ((size_t)b * a.Lq + qi) * a.lddo + head * DH;
    const float p = lane < a.Lk ? probs[(size_t)w * a.Lk + lane] : 0.f;
    float dsc = 1.f;
    if (a.drop_p > 0.f && lane < a.Lk) dsc = drop_scale(drop_rowkey(a.seed, (uint32_t)w), (uint32_t)lane, a.thresh, a.inv_keep);
    float dpj = 0.f;
    if (lane < a.Lk) {
      const float* vp = a.v + ((size_t)b * a.Lk + lane) * a.ldv + head * DH;
#pragma unroll 8
      for (int d = 0; d < DH; ++d) dpj += dop[d] * vp[d];
    }
    dpj *= dsc;
    const float dot = wave_sum(p * dpj);
    const float dsj = p * (dpj - dot);
    const float pdj = p * dsc;
    const float dod = dop[lane];
    const float qd = a.q[((size_t)b * a.Lq + qi) * a.ldq + head * DH + lane];
    float dqd = 0.f;
    for (int j = 0; j < a.Lk; ++j) {
      const float ds = __shfl(dsj, j, 64);
      const float pj = __shfl(pdj, j, 64);
      const size_t krow = ((size_t)b * a.Lk + j);
      dqd += ds * a.k[krow * a.ldk + head * DH + lane];
      float* pk = &a.dk[krow * a.ldk + head * DH + lane];
      float* pv = &a.dv[krow * a.ldv + head * DH + lane];
      *pk = *pk + ds * qd * 0.125f;              // same lane, same address every query: program order = query order
      *pv = *pv + pj * dod;
    }
    a.dq[((size_t)b * a.Lq + qi) * a.ldq + head * DH + lane] = dqd * 0.125f;
  }
}

static int check_attn(const AttnArgs& a, const char* who) {
  HOISDF_REQUIRE(a.q && a.k && a.v, HOISDF_ERR_INVALID, "%s: null pointer", who);
  HOISDF_REQUIRE(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0 && a.kv_len > 0 && a.kv_len <= a.Lk, HOISDF_ERR_INVALID,
                 "%s: bad sizes B=%d H=%d Lq=%d Lk=%d kv_len=%d", who, a.B, a.H, a.Lq, a.Lk, a.kv_len);
  HOISDF_REQUIRE(a.ldq >= a.H * DH && a.ldk >= a.H * DH && a.ldv >= a.H * DH && (a.ldq & 3) == 0 &&
                     (a.ldk & 3) == 0 && (a.ldv & 3) == 0, HOISDF_ERR_INVALID,
                 "%s: leading dims must be multiples of 4 and >= H*64", who);
  HOISDF_REQUIRE((((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) & 15) == 0, HOISDF_ERR_INVALID,
                 "%s: q/k/v must be 16-byte aligned", who);
  HOISDF_REQUIRE(a.drop_p >= 0.f && a.drop_p < 1.f, HOISDF_ERR_INVALID, "%s: drop_p", who);
  return 0;
}

}  // namespace hoisdf

using namespace hoisdf;

extern "C" int hoisdf_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    float* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len,
                                    float drop_p, uint64_t seed, void* stream) {
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.kv_len = kv_len;
  a.drop_p = drop_p; a.inv_keep = 1.f / (1.f - drop_p); a.thresh = drop_threshold(drop_p); a.seed = seed;
  if (int rc = check_attn(a, "attention_fwd")) return rc;
  HOISDF_REQUIRE(o && ldo >= H * DH && (ldo & 3) == 0 && ((uintptr_t)o & 15) == 0, HOISDF_ERR_INVALID,
                 "attention_fwd: bad output");
  if (Lq <= 32) {
    // (b, head) pairs x key splits ~ 256 blocks, a split = at least 16 key tiles (two per wave); HOISDF_FEWQ_SPLIT=0: never split
    static int split_on = -1;
    if (split_on < 0) { const char* e = getenv("HOISDF_FEWQ_SPLIT"); split_on = (e && atoi(e) == 0) ? 0 : 1; }
    const int ntiles = cdiv(kv_len, 32);
    int nsplit = split_on ? min(max(1, 256 / (B * H)), max(1, ntiles / 16)) : 1;
    float* part = nsplit > 1 ? reinterpret_cast<float*>(mag_scratch(as_stream(stream), (long)B * H * nsplit * (DH + 2) * 32)) : nullptr;
    if (!part) nsplit = 1;
    hipLaunchKernelGGL(attn_fwd_fewq_kernel, dim3(B * H, nsplit), dim3(64 * FEWQ_WAVES), 0, as_stream(stream), a, part);
    if (nsplit > 1) hipLaunchKernelGGL(attn_fewq_merge_kernel, dim3(B * H), dim3(256), 0, as_stream(stream), a, part, nsplit);
    return check_launch("attention_fwd_fewq");
  }
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(Lq, 128) * 8 * cdiv(B * H, 8)), dim3(256), 0, as_stream(stream), a);
  return check_launch("attention_fwd");
}

extern "C" int hoisdf_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    const float* o, int ldo, const float* dout, int lddo, const float* lse,
                                    float* delta, float* dq, float* dk, float* dv, int B, int H, int Lq, int Lk,
                                    int kv_len, float drop_p, uint64_t seed, void* stream) {
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout; a.lse_in = lse; a.delta_in = delta;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.kv_len = kv_len;
  a.drop_p = drop_p; a.inv_keep = 1.f / (1.f - drop_p); a.thresh = drop_threshold(drop_p); a.seed = seed;
  if (int rc = check_attn(a, "attention_bwd")) return rc;
  HOISDF_REQUIRE(o && dout && lse && delta && dq && dk && dv, HOISDF_ERR_INVALID, "attention_bwd: null pointer");
  HOISDF_REQUIRE(ldo >= H * DH && lddo >= H * DH && (ldo & 3) == 0 && (lddo & 3) == 0 &&
                     (((uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                 HOISDF_ERR_INVALID, "attention_bwd: bad leading dims / alignment");
  hipStream_t st = as_stream(stream);
  const long ng = (long)B * Lq * H;
  static const int env_mode = [] { const char* e = getenv("HOISDF_ATTN_BWD"); return (e && e[0] == 's') ? 0 : 1; }();
  const int mode = deterministic_mode() ? 0 : env_mode;        // deterministic: the two-kernel form (no dQ atomics)
  // delta = rowsum(dO * O); in fused mode the same pass clears dq, which the fused kernel accumulates with atomics
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((ng * 16 + 255) / 256)), dim3(256), 0, st, a, delta, mode);
  if (int rc = check_launch("attention_delta")) return rc;
  if (mode == 1) {
    hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(cdiv(Lk, 128) * 8 * cdiv(B * H, 8)), dim3(256), 0, st, a);
    return check_launch("attention_bwd_fused");
  }
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(cdiv(Lk, 128) * 8 * cdiv(B * H, 8)), dim3(256), 0, st, a);
  if (int rc = check_launch("attention_bwd_dkv")) return rc;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(cdiv(Lq, 128) * 8 * cdiv(B * H, 8)), dim3(256), 0, st, a);
  return check_launch("attention_bwd_dq");
}

extern "C" int hoisdf_attention_small_fwd(const float* q, int ldq, const float* k, int ldk, const float* v,
                                          int ldv, const uint8_t* mask, float* o, int ldo, float* probs, int B,
                                          int H, int Lq, int Lk, float drop_p, uint64_t seed, void* stream) {
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.out = o;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.kv_len = Lk;
  a.drop_p = drop_p; a.inv_keep = 1.f / (1.f - drop_p); a.thresh = drop_threshold(drop_p); a.seed = seed;
  if (int rc = check_attn(a, "attention_small_fwd")) return rc;
  HOISDF_REQUIRE(o && probs && Lq <= 64 && Lk <= 64, HOISDF_ERR_INVALID, "attention_small_fwd: Lq, Lk must be <= 64");
  const long nw = (long)B * H * Lq;
  hipLaunchKernelGGL(attn_small_fwd_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, as_stream(stream), a, mask,
                     probs);
  return check_launch("attention_small_fwd");
}

extern "C" int hoisdf_attention_small_bwd(const float* q, int ldq, const float* k, int ldk, const float* v,
                                          int ldv, const float* probs, const float* dout, int lddo, float* dq,
                                          float* dk, float* dv, int B, int H, int Lq, int Lk, float drop_p,
                                          uint64_t seed, void* stream) {
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = lddo;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.kv_len = Lk;
  a.drop_p = drop_p; a.inv_keep = 1.f / (1.f - drop_p); a.thresh = drop_threshold(drop_p); a.seed = seed;
  if (int rc = check_attn(a, "attention_small_bwd")) return rc;
  HOISDF_REQUIRE(probs && dout && dq && dk && dv && Lq <= 64 && Lk <= 64, HOISDF_ERR_INVALID,
                 "attention_small_bwd: bad arguments");
  const long nw = (long)B * H * Lq;
  if (deterministic_mode()) {
    hipLaunchKernelGGL(attn_small_bwd_det_kernel, dim3((unsigned)((B * H + 3) / 4)), dim3(256), 0, as_stream(stream), a, probs);
    return check_launch("attention_small_bwd_det");
  }
  hipLaunchKernelGGL(attn_small_bwd_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, as_stream(stream), a, probs);
  return check_launch("attention_small_bwd");
}
