// Inference-only attention with f16 MFMA operands (BASELINE.json configs[4]: "fp16 MFMA attention"):
// Q, K, V are rounded to f16, S = QK^T and O = PV accumulate in f32 on v_mfma_f32_32x32x16_f16, the
// softmax state (max, sum, rescale) is f32 in the log2 domain, P is rounded to f16 for the PV product.
// Scores keep ~21 bits: Q and K are split into f16 hi + lo parts and S = Kh.Qh + Kh.Ql + Kl.Qh (3 MFMAs; a single
// f16 rounding of Q and K perturbs logits of magnitude 50 by 2.5e-2, i.e. 2.5 % in the probabilities - measured
// 4.9e-4 m on the joints of the golden model, over the 1e-4 bar).  The kernel is softmax-VALU bound, so the
// two extra MFMAs per k-step are nearly free.  The same hi + lo split is applied to P and V in the PV product
// (O = Vh.Ph + Vl.Ph + Vh.Pl): with single-rounded P and V the golden joints were still 1.7e-4 m off.
// Same decomposition as the f32 kernel (attention.hip): block = 128 queries of one (b, head), S^T = K.Q^T
// so a lane owns one query column, 64-key tiles double-buffered in LDS, every tile of a (b, head) on one XCD.
// A conversion pre-pass writes K as f16 [b*H+h][Lk][64] and V transposed, f16 [b*H+h][64][Lkp] (Lkp = Lk
// rounded up to 64), so both MFMA A operands are contiguous 8- / 16-byte LDS reads:
//   S^T: A = K[key = lane&31][d = 16 j + 8 h .. +7]        B = Q^T from registers (same d)
//   O^T: A = V^T[d = lane&31][8 keys in accumulator order]  B = P^T = the 8 accumulator registers of that k-step
// The f32 path remains the parity configuration; this one is opt-in (cfg.attention_f16_eval) and never used
// when a gradient is required.
#include "common.h"
#include <hip/hip_fp16.h>

namespace hoisdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int DH16 = 64;
constexpr int KP = 72;     // halves per K row in LDS (144 B: 16-byte reads of 32 consecutive rows spread over all banks)
constexpr int VP = 72;     // halves per V^T row (d) in LDS
constexpr float QSCALE2H = 0.125f * 1.4426950408889634f;
#define CROW16(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct F16Args {
  const float* q; const _Float16* kh; const _Float16* kl; const _Float16* vt; const _Float16* vl; float* out;
  int ldq, ldo, B, H, Lq, Lk, Lkp, kv_len;
};

// K -> f16 rows, V -> f16 transposed.  One wave per (bh, 64-key block): lane = key.
__global__ __launch_bounds__(256) void attn_f16_convert_kernel(const float* __restrict__ k, int ldk,
                                                               const float* __restrict__ v, int ldv,
                                                               _Float16* __restrict__ kh, _Float16* __restrict__ kl,
                                                               _Float16* __restrict__ vt, _Float16* __restrict__ vl, int B, int H, int Lk,
                                                               int Lkp) {
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nkb = Lkp / 64;
  if (w >= (long)B * H * nkb) return;
  const int kb = (int)(w % nkb), bh = (int)(w / nkb), b = bh / H, head = bh - b * H;
  const int key = kb * 64 + lane;
  const bool valid = key < Lk;
  const float* kr = k + ((size_t)b * Lk + (valid ? key : 0)) * ldk + head * DH16;
  const float* vr = v + ((size_t)b * Lk + (valid ? key : 0)) * ldv + head * DH16;
  _Float16* ko = kh + ((size_t)bh * Lkp + key) * DH16;
  _Float16* lo = kl + ((size_t)bh * Lkp + key) * DH16;
  _Float16* vo = vt + (size_t)bh * DH16 * Lkp + key;
  _Float16* wo = vl + (size_t)bh * DH16 * Lkp + key;
#pragma unroll
  for (int d4 = 0; d4 < 16; ++d4) {
    const float4 a = valid ? *reinterpret_cast<const float4*>(kr + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 c = valid ? *reinterpret_cast<const float4*>(vr + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
    f16x4 h4 = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w};
    *reinterpret_cast<f16x4*>(ko + 4 * d4) = h4;
    f16x4 l4 = {(_Float16)(a.x - (float)h4[0]), (_Float16)(a.y - (float)h4[1]), (_Float16)(a.z - (float)h4[2]),
                (_Float16)(a.w - (float)h4[3])};
    *reinterpret_cast<f16x4*>(lo + 4 * d4) = l4;
    const float cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {                             // 64 lanes -> 64 consecutive keys of one d row
      const _Float16 hi = (_Float16)cv[i];
      vo[(size_t)(4 * d4 + i) * Lkp] = hi;
      wo[(size_t)(4 * d4 + i) * Lkp] = (_Float16)(cv[i] - (float)hi);
    }
  }
}

__device__ __forceinline__ bool attn16_block(int nx, int nbh, int& tile, int& bh) {
  const int L = blockIdx.x, slot = L >> 3;
  bh = (slot / nx) * 8 + (L & 7);
  tile = slot % nx;
  return bh < nbh;
}

__global__ __launch_bounds__(256, 2) void attn_fwd_f16_kernel(F16Args a) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * (2 * 64 * KP + 2 * 64 * VP)];    // [buf][Kh 64xKP | Kl 64xKP | Vth 64xVP | Vtl 64xVP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  int qtile, bh;
  if (!attn16_block((a.Lq + 127) / 128, a.B * a.H, qtile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int qrow = qtile * 128 + wave * 32 + c;
  const _Float16* kb = a.kh + (size_t)bh * a.Lkp * DH16;
  const _Float16* klb = a.kl + (size_t)bh * a.Lkp * DH16;
  const _Float16* vb = a.vt + (size_t)bh * DH16 * a.Lkp;
  const _Float16* vlb = a.vl + (size_t)bh * DH16 * a.Lkp;

  // Q^T fragment: for k-step j the lane supplies d = 16 j + 8 h .. + 7 of its query
  f16x8 qf[4], ql[4];
  {
    const float* qp = a.q + ((size_t)b * a.Lq + (qrow < a.Lq ? qrow : 0)) * a.ldq + head * DH16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 x = *reinterpret_cast<const float4*>(qp + 16 * j + 8 * h);
      const float4 y = *reinterpret_cast<const float4*>(qp + 16 * j + 8 * h + 4);
      const float s = qrow < a.Lq ? QSCALE2H : 0.f;
      const float e[8] = {x.x * s, x.y * s, x.z * s, x.w * s, y.x * s, y.y * s, y.z * s, y.w * s};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        qf[j][i] = (_Float16)e[i];
        ql[j][i] = (_Float16)(e[i] - (float)qf[j][i]);
      }
    }
  }
  float m = -INFINITY, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

  // cooperative tile load: 64 rows x 128 B for K and for V^T -> 2 x (256 threads x 2 x 16 B)
  const int lrow = tid >> 3, lcol = (tid & 7) * 8;          // row 0..31 (+32), 8 halves at lcol
  uint4 rk[2], rl[2], rv[2], rw[2];
  auto load = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = lrow + 32 * i;
      rk[i] = *reinterpret_cast<const uint4*>(kb + ((size_t)kt * 64 + row) * DH16 + lcol);
      rl[i] = *reinterpret_cast<const uint4*>(klb + ((size_t)kt * 64 + row) * DH16 + lcol);
      rv[i] = *reinterpret_cast<const uint4*>(vb + (size_t)row * a.Lkp + kt * 64 + lcol);
      rw[i] = *reinterpret_cast<const uint4*>(vlb + (size_t)row * a.Lkp + kt * 64 + lcol);
    }
  };
  auto store = [&](_Float16* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = lrow + 32 * i;
      *reinterpret_cast<uint4*>(buf + row * KP + lcol) = rk[i];
      *reinterpret_cast<uint4*>(buf + 64 * KP + row * KP + lcol) = rl[i];
      *reinterpret_cast<uint4*>(buf + 2 * 64 * KP + row * VP + lcol) = rv[i];
      *reinterpret_cast<uint4*>(buf + 2 * 64 * KP + 64 * VP + row * VP + lcol) = rw[i];
    }
  };
  const int ntiles = (a.kv_len + 63) / 64;
  load(0);
  store(lds);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ntiles) load(kt + 1);
    const _Float16* Ks = lds + cur * (2 * 64 * KP + 2 * 64 * VP);
    const _Float16* Ls = Ks + 64 * KP;
    const _Float16* Vs = Ks + 2 * 64 * KP;
    const _Float16* Ws = Vs + 64 * VP;

    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f16x8 kk = *reinterpret_cast<const f16x8*>(&Ks[(t * 32 + c) * KP + 16 * j + 8 * h]);
        const f16x8 kl8 = *reinterpret_cast<const f16x8*>(&Ls[(t * 32 + c) * KP + 16 * j + 8 * h]);
        s[t] = MFMA16(kl8, qf[j], s[t]);          // small terms first
        s[t] = MFMA16(kk, ql[j], s[t]);
        s[t] = MFMA16(kk, qf[j], s[t]);
      }
    }
    if (kt == ntiles - 1) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 64 + t * 32 + CROW16(r, h) >= a.kv_len) s[t][r] = -INFINITY;
    }
    float mt = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[t][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[t][r] - mn);
        ps += p;
        s[t][r] = p;
      }
    lsum = lsum * alpha + ps;
    m = mn;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // O^T += V^T . P^T: k-step (t, jj) covers the keys held by accumulator registers 8 jj .. 8 jj + 7 of
    // sub-tile t, i.e. keys t*32 + 16 jj + 4 h + {0..3} and + 8 + {0..3}
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        f16x8 pf, pl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          pf[i] = (_Float16)s[t][8 * jj + i];
          pl[i] = (_Float16)(s[t][8 * jj + i] - (float)pf[i]);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int off = (dt * 32 + c) * VP + t * 32 + 16 * jj + 4 * h;
          const f16x4 v0 = *reinterpret_cast<const f16x4*>(&Vs[off]);
          const f16x4 v1 = *reinterpret_cast<const f16x4*>(&Vs[off + 8]);
          const f16x4 w0 = *reinterpret_cast<const f16x4*>(&Ws[off]);
          const f16x4 w1 = *reinterpret_cast<const f16x4*>(&Ws[off + 8]);
          const f16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          const f16x8 ww = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
          o[dt] = MFMA16(ww, pf, o[dt]);
          o[dt] = MFMA16(vv, pl, o[dt]);
          o[dt] = MFMA16(vv, pf, o[dt]);
        }
      }
    if (kt + 1 < ntiles) store(lds + (cur ^ 1) * (2 * 64 * KP + 2 * 64 * VP));
    __syncthreads();
  }

  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  if (qrow < a.Lq) {
    const float inv = 1.f / ltot;
    float* op = a.out + ((size_t)b * a.Lq + qrow) * a.ldo + head * DH16;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(op + 32 * t + 8 * g + 4 * h) =
            make_float4(o[t][4 * g + 0] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
  }
}

}  // namespace hoisdf

using namespace hoisdf;

extern "C" long hoisdf_attention_f16_workspace(int B, int H, int Lk) {
  if (B <= 0 || H <= 0 || Lk <= 0) return 0;
  const long Lkp = ((long)Lk + 63) / 64 * 64;
  return 4L * B * H * Lkp * DH16 * (long)sizeof(_Float16);      // K hi, K lo, V^T hi, V^T lo
}

extern "C" int hoisdf_attention_fwd_f16(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                        float* o, int ldo, int B, int H, int Lq, int Lk, int kv_len,
                                        void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(q && k && v && o && workspace, HOISDF_ERR_INVALID, "attention_fwd_f16: null pointer");
  HOISDF_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && kv_len > 0 && kv_len <= Lk, HOISDF_ERR_INVALID,
                 "attention_fwd_f16: bad sizes B=%d H=%d Lq=%d Lk=%d kv_len=%d", B, H, Lq, Lk, kv_len);
  HOISDF_REQUIRE(ldq >= H * DH16 && ldk >= H * DH16 && ldv >= H * DH16 && ldo >= H * DH16 && ((ldq | ldk | ldv | ldo) & 3) == 0 &&
                     (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)workspace) & 15) == 0,
                 HOISDF_ERR_INVALID, "attention_fwd_f16: leading dims must be multiples of 4 and >= H*64, pointers 16-byte aligned");
  HOISDF_REQUIRE(workspace_bytes >= hoisdf_attention_f16_workspace(B, H, Lk), HOISDF_ERR_WORKSPACE,
                 "attention_fwd_f16: workspace %ld < %ld bytes", workspace_bytes, hoisdf_attention_f16_workspace(B, H, Lk));
  const int Lkp = (Lk + 63) / 64 * 64;
  _Float16* kh = reinterpret_cast<_Float16*>(workspace);
  _Float16* kl = kh + (size_t)B * H * Lkp * DH16;
  _Float16* vt = kl + (size_t)B * H * Lkp * DH16;
  _Float16* vl = vt + (size_t)B * H * Lkp * DH16;
  hipStream_t st = as_stream(stream);
  const long nw = (long)B * H * (Lkp / 64);
  hipLaunchKernelGGL(attn_f16_convert_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, st, k, ldk, v, ldv, kh, kl, vt,
                     vl, B, H, Lk, Lkp);
  if (int rc = check_launch("attention_f16_convert")) return rc;
  F16Args a{q, kh, kl, vt, vl, o, ldq, ldo, B, H, Lq, Lk, Lkp, kv_len};
  hipLaunchKernelGGL(attn_fwd_f16_kernel, dim3(cdiv(Lq, 128) * 8 * cdiv(B * H, 8)), dim3(256), 0, st, a);
  return check_launch("attention_fwd_f16");
}
