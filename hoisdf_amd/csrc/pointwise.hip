// HBM-bound row kernels of the hot path: NeRF positional encoding (K3), weight-norm fold,
// the 512->1 SDF head with tanh/clamp (tail of K4), sigma-gated token assembly (K8),
// and residual+dropout+LayerNorm.
// All are one-pass, coalesced (16-byte lanes where the row width allows), with wave-level
// reductions; none is reshaped into a GEMM.
#include <stdlib.h>

#include "common.h"

namespace hoisdf {

// ------------------------------------------------------------------------------------------
// posenc: x0[r][col0 + 6k + {0,1,2}] = sin(2^k p), [.. + 3 + {0,1,2}] = cos(2^k p), k=0..4,
// then xyz; common/utils/sdf_utils.py:113-126 (freq bands 2^linspace(0,4,5) = 1,2,4,8,16).
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ pts, long n_rows,
                                                     float* __restrict__ x0, int ldx0, int col0,
                                                     float* __restrict__ pe, uint32_t* __restrict__ x0_mag) {
  long idx = (long)blockIdx.x * 256 + threadIdx.x;       // one thread per (row, slot<36)
  const long total = n_rows * 36;
  for (; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / 36;
    const int s = (int)(idx - r * 36);
    float val = 0.f;
    if (s < 30) {
      const int k = s / 6, w = s - k * 6, d = w % 3;
      const float f = (float)(1 << k);
      const float arg = pts[r * 3 + d] * f;
      val = (w < 3) ? sinf(arg) : cosf(arg);
      if (pe) pe[r * 30 + s] = val;
    } else if (s < 33) {
      val = pts[r * 3 + (s - 30)];
    }
    if (x0 && col0 + s < ldx0) x0[(size_t)r * ldx0 + col0 + s] = val;   // s in [33,36) zero-fills the pad
    // row magnitudes (common.h) of what goes into x0, when wanted: an upper bound per row - the sines / cosines are <= 1, the three
    // coordinates as they are (one atomic per row instead of 36)
    if (x0_mag && s == 0)
      atomicMax(x0_mag + r, max(max(mag_bits(1.f), mag_bits(pts[r * 3])), max(mag_bits(pts[r * 3 + 1]), mag_bits(pts[r * 3 + 2]))));
  }
}

// ------------------------------------------------------------------------------------------
// weight norm: one wave per output row
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(const float* __restrict__ v,
                                                             const float* __restrict__ g, float* __restrict__ W,
                                                             int ldw, float* __restrict__ norms, int out, int in) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= out) return;
  const float* vr = v + (size_t)r * in;
  float s = 0.f;
  for (int c = lane; c < in; c += 64) s += vr[c] * vr[c];
  const float nrm = sqrtf(wave_sum(s));
  const float sc = g[r] / nrm;
  for (int c = lane; c < ldw; c += 64) W[(size_t)r * ldw + c] = c < in ? vr[c] * sc : 0.f;
  if (norms && lane == 0) norms[r] = nrm;
}

__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(const float* __restrict__ v,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ dW, int ldw,
                                                             float* __restrict__ dv, float* __restrict__ dg,
                                                             int out, int in) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= out) return;
  const float* vr = v + (size_t)r * in;
  const float* dr = dW + (size_t)r * ldw;
  float s = 0.f, d = 0.f;
  for (int c = lane; c < in; c += 64) { s += vr[c] * vr[c]; d += dr[c] * vr[c]; }
  s = wave_sum(s); d = wave_sum(d);
  const float nrm = sqrtf(s);
  const float gn = g[r] / nrm;
  const float coef = d / s;
  for (int c = lane; c < in; c += 64) dv[(size_t)r * in + c] = gn * (dr[c] - vr[c] * coef);
  if (lane == 0) dg[r] = d / nrm;
}

// ------------------------------------------------------------------------------------------
// SDF head: one wave per row; K = 512
__global__ __launch_bounds__(256) void sdf_head_fwd_kernel(const float* __restrict__ h, int ldh,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ b,
                                                           float* __restrict__ sdf_raw, float* __restrict__ sdf,
                                                           long n_rows, int K, float clampv) {
  const int lane = threadIdx.x & 63;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (; r < n_rows; r += (long)gridDim.x * 4) {
    const float* hr = h + (size_t)r * ldh;
    float s = 0.f;
    for (int c = lane * 4; c < K; c += 256) {
      const float4 a = *reinterpret_cast<const float4*>(hr + c);
      const float4 ww = *reinterpret_cast<const float4*>(w + c);
      s += a.x * ww.x + a.y * ww.y + a.z * ww.z + a.w * ww.w;
    }
    s = wave_sum(s);
    if (lane == 0) {
      const float t = tanhf(s + b[0]);
      if (sdf_raw) sdf_raw[r] = t;
      if (sdf) sdf[r] = fminf(fmaxf(t, -clampv), clampv);
    }
  }
}

__global__ __launch_bounds__(256) void sdf_head_bwd_kernel(const float* __restrict__ dsdf,
                                                           const float* __restrict__ sdf_raw,
                                                           const float* __restrict__ h, int ldh,
                                                           const float* __restrict__ w, float* __restrict__ dh,
                                                           int lddh, float* __restrict__ dw, float* __restrict__ db,
                                                           long n_rows, int K, float clampv, DetScratch ds, uint32_t* __restrict__ dh_mag) {
  // each wave walks rows with a grid stride (two rows in flight) and keeps a private dw accumulator per lane
  // slot; the four waves of a block are summed in LDS so every block issues ONE atomic per column (the same 512
  // addresses are hit by every block: per-wave atomics serialised 2048 adds per address, 238 us -> see DESIGN.md)
  __shared__ float red[4][512 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 dwa[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};   // K <= 512: 2 float4 per lane
  float dba = 0.f;
  const long stride = (long)gridDim.x * 4;
  for (long r0 = (long)blockIdx.x * 4 + wave; r0 < n_rows; r0 += 2 * stride) {
    const long r1 = r0 + stride;
    const bool two = r1 < n_rows;
    const float t0 = sdf_raw[r0], t1 = two ? sdf_raw[r1] : 0.f;
    // clamp passes gradient only inside [-clamp, clamp] (torch.clamp backward)
    const float g0 = (t0 >= -clampv && t0 <= clampv) ? dsdf[r0] * (1.f - t0 * t0) : 0.f;
    const float g1 = (two && t1 >= -clampv && t1 <= clampv) ? dsdf[r1] * (1.f - t1 * t1) : 0.f;
    const float* h0 = h + (size_t)r0 * ldh;
    const float* h1 = h + (size_t)(two ? r1 : r0) * ldh;
    uint32_t hmax0 = 0u, hmax1 = 0u;                     // dh's row magnitudes (common.h), when wanted
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = lane * 4 + i * 256;
      if (c < K) {
        const float4 a0 = *reinterpret_cast<const float4*>(h0 + c);
        const float4 a1 = *reinterpret_cast<const float4*>(h1 + c);
        const float4 ww = *reinterpret_cast<const float4*>(w + c);
        const float4 d0 = make_float4(g0 * ww.x, g0 * ww.y, g0 * ww.z, g0 * ww.w), d1 = make_float4(g1 * ww.x, g1 * ww.y, g1 * ww.z, g1 * ww.w);
        *reinterpret_cast<float4*>(dh + (size_t)r0 * lddh + c) = d0;
        if (two) *reinterpret_cast<float4*>(dh + (size_t)r1 * lddh + c) = d1;
        hmax0 = max(hmax0, mag_bits4(d0)); hmax1 = max(hmax1, mag_bits4(d1));
        dwa[i].x += g0 * a0.x + g1 * a1.x; dwa[i].y += g0 * a0.y + g1 * a1.y;
        dwa[i].z += g0 * a0.z + g1 * a1.z; dwa[i].w += g0 * a0.w + g1 * a1.w;
      }
    }
    dba += g0 + g1;
    rowmag_publish_wave(dh_mag, r0, hmax0);
    if (two) rowmag_publish_wave(dh_mag, r1, hmax1);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(&red[wave][lane * 4 + i * 256]) = dwa[i];
  if (lane == 0) red[wave][512] = dba;
  __syncthreads();
  // column sums of the block: red[0][0..K) = dw, red[0][K] = db
  for (int c = threadIdx.x; c < K; c += 256) red[0][c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
  if (threadIdx.x == 0) red[1][0] = red[0][512] + red[1][512] + red[2][512] + red[3][512];
  __syncthreads();
  if (threadIdx.x == 0) red[0][K] = red[1][0];
  __syncthreads();
  __shared__ unsigned s_last;
  block_column_sum(dw, K, db, 1, red[0], &s_last, ds);
}

// ------------------------------------------------------------------------------------------
// token assembly: one wave per token row (D = 256 -> 64 float4 lanes)
__global__ __launch_bounds__(256) void token_build_fwd_kernel(const float* __restrict__ cam,
                                                              const float* __restrict__ center,
                                                              const float* __restrict__ pe,
                                                              const float* __restrict__ feat, int ldfeat,
                                                              const float* __restrict__ sdf,
                                                              const float* __restrict__ beta_ptr,
                                                              float* __restrict__ tok, int B, int P, int S, int row0,
                                                              int D) {
  const int lane = threadIdx.x & 63;
  const float beta = fmaxf(beta_ptr[0], 2e-3f);
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long n = (long)B * P;
  for (; r < n; r += (long)gridDim.x * 4) {
    const int b = (int)(r / P), p = (int)(r - (long)b * P);
    const float sig = (1.f / (1.f + expf(-sdf[r] / beta))) / beta;
    float* o = tok + ((size_t)b * S + row0 + p) * D;
    for (int c = lane; c < D; c += 64) {
      float v;
      if (c < 3) v = cam[r * 3 + c] - center[b * 3 + c];
      else if (c < 33) v = pe[r * 30 + (c - 3)];
      else v = feat[(size_t)r * ldfeat + (c - 33)] * sig;
      o[c] = v;
    }
  }
}

__global__ __launch_bounds__(256) void token_build_bwd_kernel(const float* __restrict__ dtok,
                                                              const float* __restrict__ feat, int ldfeat,
                                                              const float* __restrict__ sdf,
                                                              const float* __restrict__ beta_ptr,
                                                              float* __restrict__ dfeat, int lddfeat,
                                                              float* __restrict__ dbeta, int B, int P, int S,
                                                              int row0, int D, DetScratch ds, float* __restrict__ partials) {
  const int lane = threadIdx.x & 63;
  const float braw = beta_ptr[0];
  const float beta = fmaxf(braw, 2e-3f);
  float acc = 0.f;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long n = (long)B * P;
  for (; r < n; r += (long)gridDim.x * 4) {
    const int b = (int)(r / P), p = (int)(r - (long)b * P);
    const float z = sdf[r] / beta;
    const float sg = 1.f / (1.f + expf(-z));
    const float sig = sg / beta;
    // d sigma / d beta = -sg(1-sg) z / beta^2 - sg / beta^2
    const float dsig = -(sg * (1.f - sg) * z + sg) / (beta * beta);
    const float* d = dtok + ((size_t)b * S + row0 + p) * D;
    for (int c = 33 + lane; c < D; c += 64) {
      const float dv = d[c];
      const float f = feat[(size_t)r * ldfeat + (c - 33)];
      dfeat[(size_t)r * lddfeat + (c - 33)] = dv * sig;
      acc += dv * f * dsig;
    }
  }
  acc = wave_sum(acc);
  if (!ds.on && !partials) {
    if (lane == 0 && dbeta) atomicAdd(dbeta, acc);
    return;
  }
  __shared__ float wsum[4];
  if (lane == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) wsum[0] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  if (partials) {                               // (ordered form) the block's sum; token_beta_finish_kernel adds them in block order
    if (threadIdx.x == 0) partials[blockIdx.x] = wsum[0];
    return;
  }
  __shared__ unsigned s_last;
  if (dbeta) block_column_sum(dbeta, 1, dbeta, 0, wsum, &s_last, ds);
}

// dbeta[0] += sum of the per-block partials, in block order (one wave: lane l adds blocks l, l + 64, ... in order, then the
// 64 lane sums are combined by the fixed shuffle tree)
__global__ __launch_bounds__(64) void token_beta_finish_kernel(const float* __restrict__ partials, int n, float* __restrict__ dbeta) {
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) a += partials[i];
  a = wave_sum(a);
  if (threadIdx.x == 0) dbeta[0] += a;
}

// ------------------------------------------------------------------------------------------
// y = LN(x + dropout(r)); one wave per row, D <= 1024 (D/4 float4 units, <= 4 per lane)
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd, long M,
                                                         int D, float eps, float drop_p, float inv_keep,
                                                         uint64_t seed, uint32_t thresh, int rpg, int take, uint32_t* __restrict__ y_mag) {
  // rpg > 0 (hoisdf_layernorm_rows_fwd): only the first `take` rows of every group of `rpg` input rows are normalised;
  // y / mean / rstd are compact ([groups * take]), M counts the compact rows
  const int lane = threadIdx.x & 63;
  const int nu = D >> 2;
  long orow = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (; orow < M; orow += (long)gridDim.x * 4) {
    uint32_t ymax = 0u;                             // y's row magnitude (common.h), when wanted
    const long row = rpg > 0 ? (orow / take) * rpg + orow % take : orow;       // input row
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = lane + 64 * i;
      v[i] = make_float4(0, 0, 0, 0);
      if (u < nu) {
        float4 a = *reinterpret_cast<const float4*>(x + (size_t)row * D + u * 4);
        if (r) {
          float4 b = *reinterpret_cast<const float4*>(r + (size_t)row * D + u * 4);
          if (drop_p > 0.f) {
            const uint32_t rk = drop_rowkey(seed, (uint32_t)row), e = (uint32_t)(u * 4);
            b.x *= drop_scale(rk, e + 0, thresh, inv_keep);
            b.y *= drop_scale(rk, e + 1, thresh, inv_keep);
            b.z *= drop_scale(rk, e + 2, thresh, inv_keep);
            b.w *= drop_scale(rk, e + 3, thresh, inv_keep);
          }
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        v[i] = a;
        s += a.x + a.y + a.z + a.w;
      }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = lane + 64 * i;
      if (u < nu) {
        const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
        q += a * a + b * b + c * c + d * d;
      }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = lane + 64 * i;
      if (u < nu) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + u * 4);
        const float4 bb = *reinterpret_cast<const float4*>(beta + u * 4);
        float4 o;
        o.x = (v[i].x - mu) * rs * g.x + bb.x;
        o.y = (v[i].y - mu) * rs * g.y + bb.y;
        o.z = (v[i].z - mu) * rs * g.z + bb.z;
        o.w = (v[i].w - mu) * rs * g.w + bb.w;
        *reinterpret_cast<float4*>(y + (size_t)orow * D + u * 4) = o;
        ymax = max(ymax, mag_bits4(o));
      }
    }
    if (lane == 0) {
      if (mean) mean[orow] = mu;
      if (rstd) rstd[orow] = rs;
    }
    rowmag_publish_wave(y_mag, orow, ymax);
  }
}

// backward: dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); dr = dx * mask/(1-p).
// dgamma/dbeta: per-wave register partials over a grid-stride of rows, then LDS + atomics.
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ r,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ dx,
                                                         float* __restrict__ dr, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, long M, int D, float drop_p,
                                                         float inv_keep, uint64_t seed, uint32_t thresh, DetScratch ds, const float* __restrict__ dx_add,
                                                         int rpg, int take, uint32_t* __restrict__ dx_mag, uint32_t* __restrict__ dr_mag) {
  // rpg > 0 (hoisdf_layernorm_rows_bwd): M counts ALL input rows; dy / mean / rstd are compact - only rows t < take of a
  // group carry a gradient, the others just pass dx_add through (or get 0)
  const int lane = threadIdx.x & 63;
  const int nu = D >> 2;
  float4 dg[4], db[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dg[i] = make_float4(0, 0, 0, 0); db[i] = make_float4(0, 0, 0, 0); }
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (; row < M; row += (long)gridDim.x * 4) {
    uint32_t xmax = 0u, rmax = 0u;          // row magnitudes of dx / dr (common.h), when wanted
    long crow = row;                        // row of dy / mean / rstd
    if (rpg > 0) {
      const long grp = row / rpg;
      const int t = (int)(row - grp * rpg);
      if (t >= take) {                      // wave-uniform: no gradient through this row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int u = lane + 64 * i;
          if (u < nu) {
            const float4 e = dx_add ? *reinterpret_cast<const float4*>(dx_add + (size_t)row * D + u * 4) : make_float4(0, 0, 0, 0);
            *reinterpret_cast<float4*>(dx + (size_t)row * D + u * 4) = e;
            xmax = max(xmax, mag_bits4(e));
          }
        }
        rowmag_publish_wave(dx_mag, row, xmax);
        continue;
      }
      crow = grp * take + t;
    }
    const float mu = mean[crow], rs = rstd[crow];
    float4 xh[4], gd[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = lane + 64 * i;
      xh[i] = make_float4(0, 0, 0, 0); gd[i] = make_float4(0, 0, 0, 0);
      if (u < nu) {
        float4 a = *reinterpret_cast<const float4*>(x + (size_t)row * D + u * 4);
        if (r) {
          float4 b = *reinterpret_cast<const float4*>(r + (size_t)row * D + u * 4);
          if (drop_p > 0.f) {
            const uint32_t rk = drop_rowkey(seed, (uint32_t)row), e = (uint32_t)(u * 4);
            b.x *= drop_scale(rk, e + 0, thresh, inv_keep);
            b.y *= drop_scale(rk, e + 1, thresh, inv_keep);
            b.z *= drop_scale(rk, e + 2, thresh, inv_keep);
            b.w *= drop_scale(rk, e + 3, thresh, inv_keep);
          }
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const float4 d = *reinterpret_cast<const float4*>(dy + (size_t)crow * D + u * 4);
        const float4 g = *reinterpret_cast<const float4*>(gamma + u * 4);
        xh[i] = make_float4((a.x - mu) * rs, (a.y - mu) * rs, (a.z - mu) * rs, (a.w - mu) * rs);
        gd[i] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
        dg[i].x += d.x * xh[i].x; dg[i].y += d.y * xh[i].y; dg[i].z += d.z * xh[i].z; dg[i].w += d.w * xh[i].w;
        db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
        s1 += gd[i].x + gd[i].y + gd[i].z + gd[i].w;
        s2 += gd[i].x * xh[i].x + gd[i].y * xh[i].y + gd[i].z * xh[i].z + gd[i].w * xh[i].w;
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = lane + 64 * i;
      if (u < nu) {
        float4 o;
        o.x = rs * (gd[i].x - s1 - xh[i].x * s2);
        o.y = rs * (gd[i].y - s1 - xh[i].y * s2);
        o.z = rs * (gd[i].z - s1 - xh[i].z * s2);
        o.w = rs * (gd[i].w - s1 - xh[i].w * s2);
        float4 ox = o;
        if (dx_add) {                       // gradient of the same tensor arriving from another consumer: summed here
          const float4 e = *reinterpret_cast<const float4*>(dx_add + (size_t)row * D + u * 4);
          ox.x += e.x; ox.y += e.y; ox.z += e.z; ox.w += e.w;
        }
        *reinterpret_cast<float4*>(dx + (size_t)row * D + u * 4) = ox;
        xmax = max(xmax, mag_bits4(ox));
        if (dr) {
          if (drop_p > 0.f) {
            const uint32_t rk = drop_rowkey(seed, (uint32_t)row), e = (uint32_t)(u * 4);
            o.x *= drop_scale(rk, e + 0, thresh, inv_keep);
            o.y *= drop_scale(rk, e + 1, thresh, inv_keep);
            o.z *= drop_scale(rk, e + 2, thresh, inv_keep);
            o.w *= drop_scale(rk, e + 3, thresh, inv_keep);
          }
          *reinterpret_cast<float4*>(dr + (size_t)row * D + u * 4) = o;
          rmax = max(rmax, mag_bits4(o));
        }
      }
    }
    rowmag_publish_wave(dx_mag, row, xmax);
    rowmag_publish_wave(dr_mag, row, rmax);
  }
  // reduce the 4 waves of the block through LDS, then one atomic per column per block
  __shared__ float red[2][4][1024];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = lane + 64 * i;
    if (u < nu) {
      *reinterpret_cast<float4*>(&red[0][w][u * 4]) = dg[i];
      *reinterpret_cast<float4*>(&red[1][w][u * 4]) = db[i];
    }
  }
  __syncthreads();
  // block column sums, contiguous: [0, D) = dgamma, [D, 2D) = dbeta  (red[0] is [4][1024] floats: D <= 1024)
  float* flat = &red[0][0][0];
  float g[4], bsum[4];
  int nc = 0;
  for (int c = threadIdx.x; c < D; c += 256, ++nc) {
    g[nc] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    bsum[nc] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
  }
  __syncthreads();
  nc = 0;
  for (int c = threadIdx.x; c < D; c += 256, ++nc) { flat[c] = g[nc]; flat[D + c] = bsum[nc]; }
  __syncthreads();
  __shared__ unsigned s_last;
  block_column_sum(dgamma, D, dbeta, D, flat, &s_last, ds);
}

// ------------------------------------------------------------------------------------------
// Round 6: the two kernels above for D <= 256 (one float4 per lane: every LayerNorm of the model, E = 256) with FOUR rows of a wave in
// flight - the row loop above is a chain (loads -> two wave reductions -> stores) that the next row's loads wait behind: the largest
// launches ran at 3.8 - 4.1 TB/s where the BatchNorm passes reach 5.7.  Per row the arithmetic is the same, expression by expression.
constexpr int LN_NR = 4;
__device__ __forceinline__ void ln_drop4(float4& b, uint64_t seed, long row, int lane, uint32_t thresh, float inv_keep) {
  const uint32_t rk = drop_rowkey(seed, (uint32_t)row), e = (uint32_t)(lane * 4);
  b.x *= drop_scale(rk, e + 0, thresh, inv_keep);
  b.y *= drop_scale(rk, e + 1, thresh, inv_keep);
  b.z *= drop_scale(rk, e + 2, thresh, inv_keep);
  b.w *= drop_scale(rk, e + 3, thresh, inv_keep);
}
__global__ __launch_bounds__(256) void add_ln_fwd256_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long M,
                                                            int D, float eps, float drop_p, float inv_keep, uint64_t seed, uint32_t thresh,
                                                            int rpg, int take, uint32_t* __restrict__ y_mag) {
  // rpg > 0 (hoisdf_layernorm_rows_fwd): as add_ln_fwd_kernel - M counts the compact output rows, input row = in_row(output row)
  const int lane = threadIdx.x & 63;
  const int nu = D >> 2;
  const bool act = lane < nu;
  auto in_row = [&](long orow) -> long { return rpg > 0 ? (orow / take) * rpg + orow % take : orow; };
  float4 g = make_float4(0, 0, 0, 0), bb = make_float4(0, 0, 0, 0);
  if (act) { g = *reinterpret_cast<const float4*>(gamma + lane * 4); bb = *reinterpret_cast<const float4*>(beta + lane * 4); }
  long base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_NR;
  const long stride = (long)gridDim.x * 4 * LN_NR;
  for (; base < M; base += stride) {
    float4 v[LN_NR], b[LN_NR];
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) {
      const long row = in_row(base + k);
      v[k] = make_float4(0, 0, 0, 0); b[k] = make_float4(0, 0, 0, 0);
      if (act && base + k < M) {
        v[k] = *reinterpret_cast<const float4*>(x + (size_t)row * D + lane * 4);
        if (r) b[k] = *reinterpret_cast<const float4*>(r + (size_t)row * D + lane * 4);
      }
    }
    float s[LN_NR], q[LN_NR], mu[LN_NR], rs[LN_NR];
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) {
      s[k] = 0.f;
      if (act && base + k < M) {
        float4 a = v[k];
        if (r) {
          float4 t = b[k];
          if (drop_p > 0.f) ln_drop4(t, seed, in_row(base + k), lane, thresh, inv_keep);
          a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        v[k] = a;
        s[k] += a.x + a.y + a.z + a.w;
      }
    }
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) mu[k] = wave_sum(s[k]) / (float)D;
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) {
      q[k] = 0.f;
      if (act && base + k < M) {
        const float a = v[k].x - mu[k], b_ = v[k].y - mu[k], c = v[k].z - mu[k], d = v[k].w - mu[k];
        q[k] += a * a + b_ * b_ + c * c + d * d;
      }
    }
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) rs[k] = rsqrtf(wave_sum(q[k]) / (float)D + eps);
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) {
      const long row = base + k;
      if (row >= M) break;
      uint32_t ymax = 0u;
      if (act) {
        float4 o;
        o.x = (v[k].x - mu[k]) * rs[k] * g.x + bb.x;
        o.y = (v[k].y - mu[k]) * rs[k] * g.y + bb.y;
        o.z = (v[k].z - mu[k]) * rs[k] * g.z + bb.z;
        o.w = (v[k].w - mu[k]) * rs[k] * g.w + bb.w;
        *reinterpret_cast<float4*>(y + (size_t)row * D + lane * 4) = o;
        ymax = mag_bits4(o);
      }
      if (lane == 0) {
        if (mean) mean[row] = mu[k];
        if (rstd) rstd[row] = rs[k];
      }
      rowmag_publish_wave(y_mag, row, ymax);
    }
  }
}

__global__ __launch_bounds__(256) void add_ln_bwd256_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ r,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dr,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, long M, int D, float drop_p,
                                                            float inv_keep, uint64_t seed, uint32_t thresh, DetScratch ds,
                                                            const float* __restrict__ dx_add, int rpg, int take,
                                                            uint32_t* __restrict__ dx_mag, uint32_t* __restrict__ dr_mag) {
  // rpg > 0 (hoisdf_layernorm_rows_bwd): as add_ln_bwd_kernel - M counts ALL input rows; dy / mean / rstd are compact: only rows
  // t < take of a group carry a gradient, the others pass dx_add through (or get 0)
  const int lane = threadIdx.x & 63;
  const int nu = D >> 2;
  const bool act = lane < nu;
  float4 dg = make_float4(0, 0, 0, 0), db = make_float4(0, 0, 0, 0), g = make_float4(0, 0, 0, 0);
  if (act) g = *reinterpret_cast<const float4*>(gamma + lane * 4);
  long base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_NR;
  const long stride = (long)gridDim.x * 4 * LN_NR;
  for (; base < M; base += stride) {
    float4 a[LN_NR], b[LN_NR], d[LN_NR], e[LN_NR];
    float mu[LN_NR], rs[LN_NR];
    bool live[LN_NR];                           // the row carries a gradient through the LayerNorm (wave-uniform)
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) {
      const long row = base + k;
      a[k] = b[k] = d[k] = e[k] = make_float4(0, 0, 0, 0);
      mu[k] = 0.f; rs[k] = 0.f;
      live[k] = false;
      if (row < M) {
        long crow = row;
        live[k] = true;
        if (rpg > 0) {
          const long grp = row / rpg;
          const int t = (int)(row - grp * rpg);
          live[k] = t < take;
          crow = grp * take + t;
        }
        if (live[k]) { mu[k] = mean[crow]; rs[k] = rstd[crow]; }
        if (act) {
          if (live[k]) {
            a[k] = *reinterpret_cast<const float4*>(x + (size_t)row * D + lane * 4);
            if (r) b[k] = *reinterpret_cast<const float4*>(r + (size_t)row * D + lane * 4);
            d[k] = *reinterpret_cast<const float4*>(dy + (size_t)crow * D + lane * 4);
          }
          if (dx_add) e[k] = *reinterpret_cast<const float4*>(dx_add + (size_t)row * D + lane * 4);
        }
      }
    }
    float4 xh[LN_NR], gd[LN_NR];
    float s1[LN_NR], s2[LN_NR];
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) {
      const long row = base + k;
      xh[k] = gd[k] = make_float4(0, 0, 0, 0);
      s1[k] = 0.f; s2[k] = 0.f;
      if (act && live[k]) {
        float4 aa = a[k];
        if (r) {
          float4 t = b[k];
          if (drop_p > 0.f) ln_drop4(t, seed, row, lane, thresh, inv_keep);
          aa.x += t.x; aa.y += t.y; aa.z += t.z; aa.w += t.w;
        }
        xh[k] = make_float4((aa.x - mu[k]) * rs[k], (aa.y - mu[k]) * rs[k], (aa.z - mu[k]) * rs[k], (aa.w - mu[k]) * rs[k]);
        gd[k] = make_float4(d[k].x * g.x, d[k].y * g.y, d[k].z * g.z, d[k].w * g.w);
        dg.x += d[k].x * xh[k].x; dg.y += d[k].y * xh[k].y; dg.z += d[k].z * xh[k].z; dg.w += d[k].w * xh[k].w;
        db.x += d[k].x; db.y += d[k].y; db.z += d[k].z; db.w += d[k].w;
        s1[k] += gd[k].x + gd[k].y + gd[k].z + gd[k].w;
        s2[k] += gd[k].x * xh[k].x + gd[k].y * xh[k].y + gd[k].z * xh[k].z + gd[k].w * xh[k].w;
      }
    }
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) { s1[k] = wave_sum(s1[k]) / (float)D; s2[k] = wave_sum(s2[k]) / (float)D; }
#pragma unroll
    for (int k = 0; k < LN_NR; ++k) {
      const long row = base + k;
      if (row >= M) break;
      uint32_t xmax = 0u, rmax = 0u;
      if (!live[k]) {                         // (no gradient through this row: dx = what arrives from the other consumer, or 0)
        if (act) { *reinterpret_cast<float4*>(dx + (size_t)row * D + lane * 4) = e[k]; xmax = mag_bits4(e[k]); }
        rowmag_publish_wave(dx_mag, row, xmax);
        continue;
      }
      if (act) {
        float4 o;
        o.x = rs[k] * (gd[k].x - s1[k] - xh[k].x * s2[k]);
        o.y = rs[k] * (gd[k].y - s1[k] - xh[k].y * s2[k]);
        o.z = rs[k] * (gd[k].z - s1[k] - xh[k].z * s2[k]);
        o.w = rs[k] * (gd[k].w - s1[k] - xh[k].w * s2[k]);
        float4 ox = o;
        if (dx_add) { ox.x += e[k].x; ox.y += e[k].y; ox.z += e[k].z; ox.w += e[k].w; }
        *reinterpret_cast<float4*>(dx + (size_t)row * D + lane * 4) = ox;
        xmax = mag_bits4(ox);
        if (dr) {
          if (drop_p > 0.f) ln_drop4(o, seed, row, lane, thresh, inv_keep);
          *reinterpret_cast<float4*>(dr + (size_t)row * D + lane * 4) = o;
          rmax = mag_bits4(o);
        }
      }
      rowmag_publish_wave(dx_mag, row, xmax);
      rowmag_publish_wave(dr_mag, row, rmax);
    }
  }
  // the block's column sums as in add_ln_bwd_kernel: the four waves through LDS, then one atomic per column per block (or the ordered fold)
  __shared__ float red[2][4][256];
  __shared__ float flat[512];
  const int w = threadIdx.x >> 6;
  if (act) {
    *reinterpret_cast<float4*>(&red[0][w][lane * 4]) = dg;
    *reinterpret_cast<float4*>(&red[1][w][lane * 4]) = db;
  }
  __syncthreads();
  if ((int)threadIdx.x < D) {
    const int c = threadIdx.x;
    flat[c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    flat[D + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
  }
  __syncthreads();
  __shared__ unsigned s_last;
  block_column_sum(dgamma, D, dbeta, D, flat, &s_last, ds);
}

// ------------------------------------------------------------------------------------------
// y = x + dropout(r)  (x == nullptr: y = dropout(r), the backward's dr = dy * mask / keep): the residual of a PRE-norm layer
// (common/nets/transformer.py:304-331,397-437; the post-norm layers fuse this into add_ln_*).  Same (seed, row, column) mask
// function as add_ln_fwd_kernel.  One float4 per thread.
__global__ __launch_bounds__(256) void residual_dropout_kernel(const float* __restrict__ x, const float* __restrict__ r, float* __restrict__ y,
                                                               long n4, int D4, float drop_p, float inv_keep, uint64_t seed, uint32_t thresh) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 b = reinterpret_cast<const float4*>(r)[i];
  if (drop_p > 0.f) {
    const long row = i / D4;
    const uint32_t rk = drop_rowkey(seed, (uint32_t)row), e = (uint32_t)((i - row * D4) * 4);
    b.x *= drop_scale(rk, e + 0, thresh, inv_keep);
    b.y *= drop_scale(rk, e + 1, thresh, inv_keep);
    b.z *= drop_scale(rk, e + 2, thresh, inv_keep);
    b.w *= drop_scale(rk, e + 3, thresh, inv_keep);
  }
  if (x) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
  }
  reinterpret_cast<float4*>(y)[i] = b;
}

}  // namespace hoisdf

using namespace hoisdf;

// HOISDF_LN_ROWS=1: add + LayerNorm with one row per wave in flight for every width (the round-1 kernels; A/B runs)
static bool ln_one_row_form() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("HOISDF_LN_ROWS"); v = (e && atoi(e) == 1) ? 1 : 0; }
  return v == 1;
}
static int row_grid(long n_rows) {
  long blocks = (n_rows + 3) / 4;
  if (blocks > 256L * 8) blocks = 256L * 8;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

extern "C" int hoisdf_posenc_fwd(const float* points, long n_rows, float* x0, int ldx0, int col0, float* pe,
                                 void* stream) {
  return posenc_fwd_mag(points, n_rows, x0, ldx0, col0, pe, nullptr, stream);
}
int hoisdf::posenc_fwd_mag(const float* points, long n_rows, float* x0, int ldx0, int col0, float* pe, uint32_t* x0_mag, void* stream) {
  HOISDF_REQUIRE(points && (x0 || pe) && n_rows >= 0, HOISDF_ERR_INVALID, "posenc_fwd: bad arguments");
  HOISDF_REQUIRE(!x0 || (col0 >= 0 && col0 + 33 <= ldx0), HOISDF_ERR_INVALID,
                 "posenc_fwd: col0=%d + 33 exceeds ldx0=%d", col0, ldx0);
  if (n_rows == 0) return HOISDF_OK;
  long total = n_rows * 36;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), points, n_rows, x0,
                     ldx0, col0, pe, x0_mag);
  return check_launch("posenc");
}

extern "C" int hoisdf_weightnorm_fwd(const float* v, const float* g, float* W, int ldw, float* norms, int out,
                                     int in, void* stream) {
  HOISDF_REQUIRE(v && g && W && out > 0 && in > 0 && ldw >= in, HOISDF_ERR_INVALID, "weightnorm_fwd: bad arguments");
  hipLaunchKernelGGL(weightnorm_fwd_kernel, dim3(cdiv(out, 4)), dim3(256), 0, as_stream(stream), v, g, W, ldw,
                     norms, out, in);
  return check_launch("weightnorm_fwd");
}

extern "C" int hoisdf_weightnorm_bwd(const float* v, const float* g, const float* dW, int ldw, float* dv,
                                     float* dg, int out, int in, void* stream) {
  HOISDF_REQUIRE(v && g && dW && dv && dg && out > 0 && in > 0 && ldw >= in, HOISDF_ERR_INVALID,
                 "weightnorm_bwd: bad arguments");
  hipLaunchKernelGGL(weightnorm_bwd_kernel, dim3(cdiv(out, 4)), dim3(256), 0, as_stream(stream), v, g, dW, ldw,
                     dv, dg, out, in);
  return check_launch("weightnorm_bwd");
}

extern "C" int hoisdf_sdf_head_fwd(const float* h, int ldh, const float* w, const float* b, float* sdf_raw,
                                   float* sdf, long n_rows, int K, float clamp, void* stream) {
  HOISDF_REQUIRE(h && w && b && (sdf_raw || sdf) && n_rows >= 0, HOISDF_ERR_INVALID, "sdf_head_fwd: null pointer");
  HOISDF_REQUIRE(K > 0 && (K & 3) == 0 && (ldh & 3) == 0 && ldh >= K && ((uintptr_t)h & 15) == 0 &&
                     ((uintptr_t)w & 15) == 0, HOISDF_ERR_INVALID, "sdf_head_fwd: K/ldh must be multiples of 4, 16-byte aligned");
  if (n_rows == 0) return HOISDF_OK;
  hipLaunchKernelGGL(sdf_head_fwd_kernel, dim3(row_grid(n_rows)), dim3(256), 0, as_stream(stream), h, ldh, w, b,
                     sdf_raw, sdf, n_rows, K, clamp);
  return check_launch("sdf_head_fwd");
}

extern "C" int hoisdf_sdf_head_bwd(const float* dsdf, const float* sdf_raw, const float* h, int ldh,
                                   const float* w, float* dh, int lddh, float* dw, float* db, long n_rows, int K,
                                   float clamp, void* stream) {
  return sdf_head_bwd_mag(dsdf, sdf_raw, h, ldh, w, dh, lddh, dw, db, n_rows, K, clamp, nullptr, stream);
}
int hoisdf::sdf_head_bwd_mag(const float* dsdf, const float* sdf_raw, const float* h, int ldh, const float* w, float* dh, int lddh, float* dw,
                             float* db, long n_rows, int K, float clamp, uint32_t* dh_mag, void* stream) {
  HOISDF_REQUIRE(dsdf && sdf_raw && h && w && dh && dw && db && n_rows >= 0, HOISDF_ERR_INVALID,
                 "sdf_head_bwd: null pointer");
  HOISDF_REQUIRE(K > 0 && K <= 512 && (K & 3) == 0 && (ldh & 3) == 0 && (lddh & 3) == 0 && ldh >= K && lddh >= K,
                 HOISDF_ERR_INVALID, "sdf_head_bwd: K must be a multiple of 4 and <= 512");
  if (n_rows == 0) return HOISDF_OK;
  int blocks = row_grid(n_rows);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sdf_head_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dsdf, sdf_raw, h, ldh, w,
                     dh, lddh, dw, db, n_rows, K, clamp, det_scratch((size_t)blocks * (K + 1)), dh_mag);
  return check_launch("sdf_head_bwd");
}

extern "C" int hoisdf_token_build_fwd(const float* cam, const float* center, const float* pe, const float* feat,
                                      int ldfeat, const float* sdf, const float* beta_ptr, float* tok, int B, int P,
                                      int S, int row0, int D, void* stream) {
  HOISDF_REQUIRE(cam && center && pe && feat && sdf && beta_ptr && tok, HOISDF_ERR_INVALID,
                 "token_build_fwd: null pointer");
  HOISDF_REQUIRE(B > 0 && P >= 0 && row0 >= 0 && row0 + P <= S && D > 33 && ldfeat >= D - 33, HOISDF_ERR_INVALID,
                 "token_build_fwd: bad sizes B=%d P=%d S=%d row0=%d D=%d", B, P, S, row0, D);
  if (P == 0) return HOISDF_OK;
  hipLaunchKernelGGL(token_build_fwd_kernel, dim3(row_grid((long)B * P)), dim3(256), 0, as_stream(stream), cam,
                     center, pe, feat, ldfeat, sdf, beta_ptr, tok, B, P, S, row0, D);
  return check_launch("token_build_fwd");
}

extern "C" int hoisdf_token_build_bwd(const float* dtok, const float* feat, int ldfeat, const float* sdf,
                                      const float* beta_ptr, float* dfeat, int lddfeat, float* dbeta, int B, int P,
                                      int S, int row0, int D, void* stream) {
  HOISDF_REQUIRE(dtok && feat && sdf && beta_ptr && dfeat, HOISDF_ERR_INVALID, "token_build_bwd: null pointer");
  HOISDF_REQUIRE(B > 0 && P >= 0 && row0 >= 0 && row0 + P <= S && D > 33 && ldfeat >= D - 33 && lddfeat >= D - 33,
                 HOISDF_ERR_INVALID, "token_build_bwd: bad sizes");
  if (P == 0) return HOISDF_OK;
  int blocks = row_grid((long)B * P);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(token_build_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dtok, feat, ldfeat,
                     sdf, beta_ptr, dfeat, lddfeat, dbeta, B, P, S, row0, D, det_scratch((size_t)blocks), (float*)nullptr);
  return check_launch("token_build_bwd");
}

extern "C" int hoisdf_token_build_bwd_partials(void) { return 1024; }

extern "C" int hoisdf_token_build_bwd_ordered(const float* dtok, const float* feat, int ldfeat, const float* sdf,
                                              const float* beta_ptr, float* dfeat, int lddfeat, float* dbeta, float* partials,
                                              int B, int P, int S, int row0, int D, void* stream) {
  HOISDF_REQUIRE(dtok && feat && sdf && beta_ptr && dfeat && dbeta && partials, HOISDF_ERR_INVALID, "token_build_bwd_ordered: null pointer");
  HOISDF_REQUIRE(B > 0 && P >= 0 && row0 >= 0 && row0 + P <= S && D > 33 && ldfeat >= D - 33 && lddfeat >= D - 33,
                 HOISDF_ERR_INVALID, "token_build_bwd_ordered: bad sizes");
  if (P == 0) return HOISDF_OK;
  int blocks = row_grid((long)B * P);
  if (blocks > 1024) blocks = 1024;
  DetScratch off{};
  hipLaunchKernelGGL(token_build_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dtok, feat, ldfeat,
                     sdf, beta_ptr, dfeat, lddfeat, dbeta, B, P, S, row0, D, off, partials);
  hipLaunchKernelGGL(token_beta_finish_kernel, dim3(1), dim3(64), 0, as_stream(stream), partials, blocks, dbeta);
  return check_launch("token_build_bwd_ordered");
}

extern "C" int hoisdf_add_layernorm_fwd(const float* x, const float* r, const float* gamma, const float* beta,
                                        float* y, float* mean, float* rstd, long M, int D, float eps, float drop_p,
                                        uint64_t seed, void* stream) {
  return add_layernorm_fwd_mag(x, r, gamma, beta, y, mean, rstd, M, D, eps, drop_p, seed, nullptr, stream);
}
int hoisdf::add_layernorm_fwd_mag(const float* x, const float* r, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                  long M, int D, float eps, float drop_p, uint64_t seed, uint32_t* y_mag, void* stream) {
  HOISDF_REQUIRE(x && gamma && beta && y && M >= 0, HOISDF_ERR_INVALID, "add_layernorm_fwd: null pointer");
  HOISDF_REQUIRE(D > 0 && D <= 1024 && (D & 3) == 0 && drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID,
                 "add_layernorm_fwd: D=%d must be a multiple of 4 and <= 1024", D);
  if (M == 0) return HOISDF_OK;
  if (D <= 256 && !ln_one_row_form()) {
    int blocks = cdiv(M, 4 * LN_NR);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(add_ln_fwd256_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, r, gamma, beta, y, mean, rstd, M, D, eps, drop_p,
                       1.f / (1.f - drop_p), seed, drop_threshold(drop_p), 0, 0, y_mag);
    return check_launch("add_layernorm_fwd");
  }
  hipLaunchKernelGGL(add_ln_fwd_kernel, dim3(row_grid(M)), dim3(256), 0, as_stream(stream), x, r, gamma, beta, y,
                     mean, rstd, M, D, eps, drop_p, 1.f / (1.f - drop_p), seed, drop_threshold(drop_p), 0, 0, y_mag);
  return check_launch("add_layernorm_fwd");
}

extern "C" int hoisdf_residual_dropout(const float* x, const float* r, float* y, long M, int D, float drop_p, uint64_t seed, void* stream) {
  HOISDF_REQUIRE(r && y && M >= 0, HOISDF_ERR_INVALID, "residual_dropout: null pointer");
  HOISDF_REQUIRE(D > 0 && (D & 3) == 0 && drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID, "residual_dropout: D=%d must be a multiple of 4, drop_p=%f", D, drop_p);
  HOISDF_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, HOISDF_ERR_INVALID,
                 "residual_dropout: buffers must be 16-byte aligned");
  const long n4 = M * (D / 4);
  if (n4 == 0) return HOISDF_OK;
  hipLaunchKernelGGL(residual_dropout_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), x, r, y, n4, D / 4, drop_p,
                     1.f / (1.f - drop_p), seed, drop_threshold(drop_p));
  return check_launch("residual_dropout");
}

extern "C" int hoisdf_layernorm_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                         float* rstd, long groups, int rows_per_group, int take, int D, float eps,
                                         void* stream) {
  HOISDF_REQUIRE(x && gamma && beta && y && groups >= 0, HOISDF_ERR_INVALID, "layernorm_rows_fwd: null pointer");
  HOISDF_REQUIRE(D > 0 && D <= 1024 && (D & 3) == 0 && rows_per_group > 0 && take > 0 && take <= rows_per_group,
                 HOISDF_ERR_INVALID, "layernorm_rows_fwd: D=%d rows_per_group=%d take=%d", D, rows_per_group, take);
  const long M = groups * take;
  if (M == 0) return HOISDF_OK;
  if (D <= 256 && !ln_one_row_form()) {
    int blocks = cdiv(M, 4 * LN_NR);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(add_ln_fwd256_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, (const float*)nullptr, gamma, beta, y, mean, rstd, M, D,
                       eps, 0.f, 1.f, (uint64_t)0, 0u, rows_per_group, take, (uint32_t*)nullptr);
    return check_launch("layernorm_rows_fwd");
  }
  hipLaunchKernelGGL(add_ln_fwd_kernel, dim3(row_grid(M)), dim3(256), 0, as_stream(stream), x, (const float*)nullptr, gamma,
                     beta, y, mean, rstd, M, D, eps, 0.f, 1.f, (uint64_t)0, 0u, rows_per_group, take, (uint32_t*)nullptr);
  return check_launch("layernorm_rows_fwd");
}

extern "C" int hoisdf_add_layernorm_bwd(const float* dy, const float* x, const float* r, const float* gamma,
                                        const float* mean, const float* rstd, const float* dx_add, float* dx, float* dr,
                                        float* dgamma, float* dbeta, long M, int D, float drop_p, uint64_t seed,
                                        void* stream) {
  return add_layernorm_bwd_mag(dy, x, r, gamma, mean, rstd, dx_add, dx, dr, dgamma, dbeta, M, D, drop_p, seed, nullptr, nullptr, stream);
}
int hoisdf::add_layernorm_bwd_mag(const float* dy, const float* x, const float* r, const float* gamma, const float* mean, const float* rstd,
                                  const float* dx_add, float* dx, float* dr, float* dgamma, float* dbeta, long M, int D, float drop_p,
                                  uint64_t seed, uint32_t* dx_mag, uint32_t* dr_mag, void* stream) {
  HOISDF_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && M >= 0, HOISDF_ERR_INVALID,
                 "add_layernorm_bwd: null pointer");
  HOISDF_REQUIRE(D > 0 && D <= 1024 && (D & 3) == 0 && drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID,
                 "add_layernorm_bwd: D=%d must be a multiple of 4 and <= 1024", D);
  if (M == 0) return HOISDF_OK;
  int blocks = row_grid(M);
  static int cap = -1;                          // HOISDF_LN_BWD_BLOCKS: blocks of the backward (A/B runs; default 1024 = 4 per CU)
  if (cap < 0) { const char* e = getenv("HOISDF_LN_BWD_BLOCKS"); cap = e && atoi(e) > 0 ? atoi(e) : 1024; }
  if (blocks > cap) blocks = cap;
  if (D <= 256 && !ln_one_row_form()) {
    blocks = cdiv(M, 4 * LN_NR);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(add_ln_bwd256_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dy, x, r, gamma, mean, rstd, dx, dr, dgamma, dbeta, M, D,
                       drop_p, 1.f / (1.f - drop_p), seed, drop_threshold(drop_p), det_scratch((size_t)blocks * 2 * D), dx_add, 0, 0, dx_mag, dr_mag);
    return check_launch("add_layernorm_bwd");
  }
  hipLaunchKernelGGL(add_ln_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dy, x, r, gamma, mean, rstd,
                     dx, dr, dgamma, dbeta, M, D, drop_p, 1.f / (1.f - drop_p), seed, drop_threshold(drop_p),
                     det_scratch((size_t)blocks * 2 * D), dx_add, 0, 0, dx_mag, dr_mag);
  return check_launch("add_layernorm_bwd");
}

extern "C" int hoisdf_layernorm_rows_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                         const float* rstd, const float* dx_add, float* dx, float* dgamma, float* dbeta,
                                         long groups, int rows_per_group, int take, int D, void* stream) {
  HOISDF_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && groups >= 0, HOISDF_ERR_INVALID,
                 "layernorm_rows_bwd: null pointer");
  HOISDF_REQUIRE(D > 0 && D <= 1024 && (D & 3) == 0 && rows_per_group > 0 && take > 0 && take <= rows_per_group,
                 HOISDF_ERR_INVALID, "layernorm_rows_bwd: D=%d rows_per_group=%d take=%d", D, rows_per_group, take);
  const long M = groups * rows_per_group;
  if (M == 0) return HOISDF_OK;
  int blocks = row_grid(M);
  if (blocks > 512) blocks = 512;
  if (D <= 256 && !ln_one_row_form()) {
    blocks = cdiv(M, 4 * LN_NR);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(add_ln_bwd256_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dy, x, (const float*)nullptr, gamma, mean, rstd, dx,
                       (float*)nullptr, dgamma, dbeta, M, D, 0.f, 1.f, (uint64_t)0, 0u, det_scratch((size_t)blocks * 2 * D), dx_add, rows_per_group,
                       take, (uint32_t*)nullptr, (uint32_t*)nullptr);
    return check_launch("layernorm_rows_bwd");
  }
  hipLaunchKernelGGL(add_ln_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dy, x, (const float*)nullptr, gamma,
                     mean, rstd, dx, (float*)nullptr, dgamma, dbeta, M, D, 0.f, 1.f, (uint64_t)0, 0u,
                     det_scratch((size_t)blocks * 2 * D), dx_add, rows_per_group, take, (uint32_t*)nullptr, (uint32_t*)nullptr);
  return check_launch("layernorm_rows_bwd");
}
