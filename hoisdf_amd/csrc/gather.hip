// K1: pinhole projection + multi-level bilinear gather from an NHWC pyramid, its scatter-add
// backward, and K5: dense-lattice generation with the strict bbox filter (stream compaction).
//
// Layout: the pyramid is channels-last, so the 4 taps of a point are 4 contiguous C_l-float
// rows per level; a wave owns one point and its lanes sweep the concatenated channel axis in
// float4 units -> every global access of the wave is a run of full 16-byte lanes.
// HBM-bound by construction (15.9 KB read + 4 KB written per point at C = 992); the pyramid
// (4 MB / sample) is L2 / Infinity-Cache resident across the ~2000-6000 points of a sample.
#include <stdlib.h>

#include "common.h"

namespace hoisdf {

struct PyrDev {
  int n_levels;
  int C4;                          // total channels / 4
  const float* data[HOISDF_MAX_LEVELS];
  float* grad[HOISDF_MAX_LEVELS];
  int C[HOISDF_MAX_LEVELS], H[HOISDF_MAX_LEVELS], W[HOISDF_MAX_LEVELS];
  int off4[HOISDF_MAX_LEVELS + 1]; // channel offset of each level, in float4 units
};

struct ProjArgs {
  const float* points;
  const int32_t* sample_idx;
  long n_rows;
  int rows_per_sample;
  const float* center;
  const float* cam_intr;
  float scale, nx, ny;
};

// cam = p/scale + c ; q = K cam ; uv = q_xy / q_z ; g = (uv - n)/n      (main/model.py:148-157)
__device__ __forceinline__ void project_row(const ProjArgs& a, long r, int& b, float (&cam)[3],
                                            float (&uv)[2], float (&g)[2]) {
  b = a.sample_idx ? a.sample_idx[r] : (int)(r / a.rows_per_sample);
  const float* p = a.points + r * 3;
  const float* c = a.center + (size_t)b * 3;
  const float* K = a.cam_intr + (size_t)b * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) cam[i] = __fadd_rn(__fdiv_rn(p[i], a.scale), c[i]);
  float q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = cam[0] * K[i * 3 + 0] + cam[1] * K[i * 3 + 1] + cam[2] * K[i * 3 + 2];
  uv[0] = __fdiv_rn(q[0], q[2]);
  uv[1] = __fdiv_rn(q[1], q[2]);
  g[0] = __fdiv_rn(uv[0] - a.nx, a.nx);
  g[1] = __fdiv_rn(uv[1] - a.ny, a.ny);
}

struct Taps {
  int o00, o01, o10, o11;   // pixel offsets (in pixels) of nw, ne, sw, se; -1 = out of range
  float w00, w01, w10, w11;
};

// ATen grid_sampler_2d, bilinear / border / align_corners=True: unnormalise, clip, floor.
__device__ __forceinline__ Taps make_taps(float gx, float gy, int W, int H) {
  float x = ((gx + 1.f) * 0.5f) * (float)(W - 1);
  float y = ((gy + 1.f) * 0.5f) * (float)(H - 1);
  x = fminf(fmaxf(x, 0.f), (float)(W - 1));
  y = fminf(fmaxf(y, 0.f), (float)(H - 1));
  float x0f = floorf(x), y0f = floorf(y);
  int x0 = (int)x0f, y0 = (int)y0f;
  float wx1 = x - x0f, wy1 = y - y0f;
  float wx0 = (x0f + 1.f) - x, wy0 = (y0f + 1.f) - y;
  Taps t;
  bool xin = x0 + 1 <= W - 1, yin = y0 + 1 <= H - 1;
  t.o00 = y0 * W + x0;
  t.o01 = xin ? y0 * W + x0 + 1 : -1;
  t.o10 = yin ? (y0 + 1) * W + x0 : -1;
  t.o11 = (xin && yin) ? (y0 + 1) * W + x0 + 1 : -1;
  t.w00 = wx0 * wy0; t.w01 = wx1 * wy0; t.w10 = wx0 * wy1; t.w11 = wx1 * wy1;
  return t;
}

__device__ __forceinline__ int level_of(const PyrDev& P, int u) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < HOISDF_MAX_LEVELS; ++i)
    if (i < P.n_levels && u >= P.off4[i]) l = i;
  return l;
}

__global__ __launch_bounds__(256) void gather_fwd_kernel(PyrDev P, ProjArgs a, float* __restrict__ feat,
                                                         int ldf, float* __restrict__ cam_out,
                                                         float* __restrict__ uv_out, uint32_t* __restrict__ feat_mag) {
  const int lane = threadIdx.x & 63;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long stride = (long)gridDim.x * 4;
  for (; r < a.n_rows; r += stride) {
    uint32_t fmax = 0u;                             // feat's row magnitude (common.h), when wanted
    int b;
    float cam[3], uv[2], g[2];
    project_row(a, r, b, cam, uv, g);
    if (lane < 3 && cam_out) cam_out[r * 3 + lane] = cam[lane];
    if (lane < 2 && uv_out) uv_out[r * 2 + lane] = uv[lane];
    float4* out = reinterpret_cast<float4*>(feat + (size_t)r * ldf);
    for (int u = lane; u < P.C4; u += 64) {
      const int l = level_of(P, u);
      const int C = P.C[l], H = P.H[l], W = P.W[l];
      const Taps t = make_taps(g[0], g[1], W, H);
      const float* base = P.data[l] + (size_t)b * H * W * C + (size_t)(u - P.off4[l]) * 4;
      float4 v = *reinterpret_cast<const float4*>(base + (size_t)t.o00 * C);
      float4 acc = make_float4(v.x * t.w00, v.y * t.w00, v.z * t.w00, v.w * t.w00);
      if (t.o01 >= 0) {
        v = *reinterpret_cast<const float4*>(base + (size_t)t.o01 * C);
        acc.x += v.x * t.w01; acc.y += v.y * t.w01; acc.z += v.z * t.w01; acc.w += v.w * t.w01;
      }
      if (t.o10 >= 0) {
        v = *reinterpret_cast<const float4*>(base + (size_t)t.o10 * C);
        acc.x += v.x * t.w10; acc.y += v.y * t.w10; acc.z += v.z * t.w10; acc.w += v.w * t.w10;
      }
      if (t.o11 >= 0) {
        v = *reinterpret_cast<const float4*>(base + (size_t)t.o11 * C);
        acc.x += v.x * t.w11; acc.y += v.y * t.w11; acc.z += v.z * t.w11; acc.w += v.w * t.w11;
      }
      out[u] = acc;
      fmax = max(fmax, mag_bits4(acc));
    }
    rowmag_publish_wave(feat_mag, r, fmax);
  }
}

// Round 6: the same gather for C <= 1024 channels (every pyramid of the model: 992 / 3968 -> the second takes the loop above) with the
// memory-level parallelism written out: a lane's (up to) four float4 columns belong to fixed levels (constants hoisted out of the row
// loop), all SIXTEEN tap loads of a row are issued before the first is used (a border tap re-reads the nw pixel and is not added - the
// sums are those of gather_fwd_kernel bit for bit), and the next row's point and sample index are requested while this row's taps are
// in flight.  Lattice points of one sdf_infer call (320 000 rows): 948 -> 691 us in configs[3]'s trace (3.2 TB/s, profiles/r06_pmc.json).
__global__ __launch_bounds__(256) void gather_fwd4_kernel(PyrDev P, ProjArgs a, float* __restrict__ feat, int ldf, float* __restrict__ cam_out,
                                                          float* __restrict__ uv_out, uint32_t* __restrict__ feat_mag) {
  const int lane = threadIdx.x & 63;
  const float* lbase[4];
  long lsz[4];
  int LC[4], LH[4], LW[4];
  bool ok[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int u = lane + 64 * it;
    ok[it] = u < P.C4;
    const int l = level_of(P, ok[it] ? u : 0);
    LC[it] = P.C[l]; LH[it] = P.H[l]; LW[it] = P.W[l];
    lsz[it] = (long)P.H[l] * P.W[l] * P.C[l];
    lbase[it] = P.data[l] + (size_t)((ok[it] ? u : 0) - P.off4[l]) * 4;
  }
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long stride = (long)gridDim.x * 4;
  if (r >= a.n_rows) return;
  // the row's inputs, one row ahead
  float pn[3]; int bn;
#define GF4_FETCH(R_)                                                                               \
  do {                                                                                              \
    bn = a.sample_idx ? a.sample_idx[(R_)] : (int)((R_) / a.rows_per_sample);                       \
    pn[0] = a.points[(R_) * 3]; pn[1] = a.points[(R_) * 3 + 1]; pn[2] = a.points[(R_) * 3 + 2];     \
  } while (0)
  GF4_FETCH(r);
  for (; r < a.n_rows; r += stride) {
    const int b = bn;
    const float p0 = pn[0], p1 = pn[1], p2 = pn[2];
    if (r + stride < a.n_rows) GF4_FETCH(r + stride);          // (issued first: back before this row's taps are)
    // cam = p/scale + c ; q = K cam ; uv = q_xy / q_z ; g = (uv - n)/n      (project_row, on the prefetched inputs)
    const float* c = a.center + (size_t)b * 3;
    const float* K = a.cam_intr + (size_t)b * 9;
    float cam[3], q[3], uv[2], g[2];
    cam[0] = __fadd_rn(__fdiv_rn(p0, a.scale), c[0]);
    cam[1] = __fadd_rn(__fdiv_rn(p1, a.scale), c[1]);
    cam[2] = __fadd_rn(__fdiv_rn(p2, a.scale), c[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = cam[0] * K[i * 3 + 0] + cam[1] * K[i * 3 + 1] + cam[2] * K[i * 3 + 2];
    uv[0] = __fdiv_rn(q[0], q[2]);
    uv[1] = __fdiv_rn(q[1], q[2]);
    g[0] = __fdiv_rn(uv[0] - a.nx, a.nx);
    g[1] = __fdiv_rn(uv[1] - a.ny, a.ny);
    Taps t[4];
    float4 v[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      t[it] = make_taps(g[0], g[1], LW[it], LH[it]);
      if (ok[it]) {
        const float* base = lbase[it] + (size_t)b * lsz[it];
        const int C = LC[it];
        v[it][0] = *reinterpret_cast<const float4*>(base + (size_t)t[it].o00 * C);
        v[it][1] = *reinterpret_cast<const float4*>(base + (size_t)(t[it].o01 >= 0 ? t[it].o01 : t[it].o00) * C);
        v[it][2] = *reinterpret_cast<const float4*>(base + (size_t)(t[it].o10 >= 0 ? t[it].o10 : t[it].o00) * C);
        v[it][3] = *reinterpret_cast<const float4*>(base + (size_t)(t[it].o11 >= 0 ? t[it].o11 : t[it].o00) * C);
      }
    }
    if (lane < 3 && cam_out) cam_out[r * 3 + lane] = cam[lane];
    if (lane < 2 && uv_out) uv_out[r * 2 + lane] = uv[lane];
    float4* out = reinterpret_cast<float4*>(feat + (size_t)r * ldf);
    uint32_t fmax = 0u;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (!ok[it]) continue;
      const Taps& tt = t[it];
      float4 x = v[it][0];
      float4 acc = make_float4(x.x * tt.w00, x.y * tt.w00, x.z * tt.w00, x.w * tt.w00);
      if (tt.o01 >= 0) { x = v[it][1]; acc.x += x.x * tt.w01; acc.y += x.y * tt.w01; acc.z += x.z * tt.w01; acc.w += x.w * tt.w01; }
      if (tt.o10 >= 0) { x = v[it][2]; acc.x += x.x * tt.w10; acc.y += x.y * tt.w10; acc.z += x.z * tt.w10; acc.w += x.w * tt.w10; }
      if (tt.o11 >= 0) { x = v[it][3]; acc.x += x.x * tt.w11; acc.y += x.y * tt.w11; acc.z += x.z * tt.w11; acc.w += x.w * tt.w11; }
      out[lane + 64 * it] = acc;
      fmax = max(fmax, mag_bits4(acc));
    }
    rowmag_publish_wave(feat_mag, r, fmax);
  }
#undef GF4_FETCH
}

__global__ __launch_bounds__(256) void gather_bwd_kernel(PyrDev P, ProjArgs a, const float* __restrict__ dfeat,
                                                         int ldf, int skip_mask) {
  const int lane = threadIdx.x & 63;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long stride = (long)gridDim.x * 4;
  for (; r < a.n_rows; r += stride) {
    int b;
    float cam[3], uv[2], g[2];
    project_row(a, r, b, cam, uv, g);
    // lanes sweep single channels, so one atomic instruction covers runs of consecutive floats (whole 64-byte
    // lines per tap) instead of one float of every fourth (the L2 atomic units work per line)
    const float* din = dfeat + (size_t)r * ldf;
    for (int c = lane; c < P.C4 * 4; c += 64) {
      const int l = level_of(P, c >> 2);
      if ((skip_mask >> l) & 1) continue;          // coarse level: handled by the LDS-privatised kernel
      const int C = P.C[l], H = P.H[l], W = P.W[l];
      const Taps t = make_taps(g[0], g[1], W, H);
      float* base = P.grad[l] + (size_t)b * H * W * C + (size_t)(c - P.off4[l] * 4);
      const float d = din[c];
      atomicAdd(base + (size_t)t.o00 * C, d * t.w00);
      if (t.o01 >= 0) atomicAdd(base + (size_t)t.o01 * C, d * t.w01);
      if (t.o10 >= 0) atomicAdd(base + (size_t)t.o10 * C, d * t.w10);
      if (t.o11 >= 0) atomicAdd(base + (size_t)t.o11 * C, d * t.w11);
    }
  }
}

// Coarse levels (<= 256 pixels): every point of a sample hits the same few pixels (128 adds per
// address at stride 32), and those levels carry ~80 % of all channel-taps.  One single-wave
// workgroup owns (sample, level, 64-channel chunk, point slice): the lanes are the 64 channels, the
// wave walks its slice of the sample's points and accumulates into a wave-private LDS image with
// plain read-modify-write (LDS ops of one wave execute in order, so no atomics: ds_add_f32 measured
// ~150 cycles per wave instruction on gfx950), then adds the image to the level gradient with one
// float atomic per element (COARSE_SLICES slices per image).
struct CoarseJob { int level, chunk; };
struct CoarseArgs { int n_jobs; CoarseJob job[96]; };
constexpr int COARSE_SLICES = 4;

__global__ __launch_bounds__(64) void gather_bwd_coarse_kernel(PyrDev P, ProjArgs a, CoarseArgs J,
                                                               const float* __restrict__ dfeat, int ldf) {
  extern __shared__ __attribute__((aligned(16))) float img[];       // [H*W][64] image, then 64 x 8 tap records
  const int lane = threadIdx.x;
  const int b = blockIdx.y, slice = blockIdx.z;
  const CoarseJob jb = J.job[blockIdx.x];
  const int l = jb.level, C = P.C[l], H = P.H[l], W = P.W[l];
  const int npix = H * W;
  float* tapw = img + npix * 64;                                    // [64][4] weights
  int* tapo = reinterpret_cast<int*>(tapw + 256);                   // [64][4] pixel offsets
  for (int i = lane; i < npix * 64; i += 64) img[i] = 0.f;
  const int ch0 = P.off4[l] * 4 + jb.chunk * 64;                    // column in dfeat rows
  const int Pn = a.rows_per_sample;
  const int nmine = (Pn - slice + COARSE_SLICES - 1) / COARSE_SLICES;   // points p = slice + k * SLICES, k < nmine
  const float* drow = dfeat + ((size_t)b * Pn + slice) * ldf + ch0 + lane;
  for (int k0 = 0; k0 < nmine; k0 += 64) {
    // projection + taps of 64 points at once (one per lane) instead of 64 redundant copies per point
    if (k0 + lane < nmine) {
      int bb;
      float cam[3], uv[2], g[2];
      project_row(a, (long)b * Pn + slice + (long)(k0 + lane) * COARSE_SLICES, bb, cam, uv, g);
      const Taps t = make_taps(g[0], g[1], W, H);
      // out-of-range taps alias the (always valid) nw pixel with weight 0
      *reinterpret_cast<float4*>(&tapw[lane * 4]) =
          make_float4(t.w00, t.o01 >= 0 ? t.w01 : 0.f, t.o10 >= 0 ? t.w10 : 0.f, t.o11 >= 0 ? t.w11 : 0.f);
      *reinterpret_cast<int4*>(&tapo[lane * 4]) =
          make_int4(t.o00, t.o01 >= 0 ? t.o01 : t.o00, t.o10 >= 0 ? t.o10 : t.o00, t.o11 >= 0 ? t.o11 : t.o00);
    }
    __builtin_amdgcn_wave_barrier();
    const int kn = min(64, nmine - k0);
    // this lane's channel of 16 points at a time: 16 independent global loads in flight (a one-deep
    // prefetch exposed ~1 us of L2/HBM latency per point), next group requested before this one is used
    float dv[16], dn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dn[i] = i < kn ? drow[(size_t)(k0 + i) * COARSE_SLICES * ldf] : 0.f;
    for (int j0 = 0; j0 < kn; j0 += 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) dv[i] = dn[i];
      if (j0 + 16 < kn) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          dn[i] = j0 + 16 + i < kn ? drow[(size_t)(k0 + j0 + 16 + i) * COARSE_SLICES * ldf] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int j = j0 + i;
        if (j < kn) {
          const float d = dv[i];
          const float4 w = *reinterpret_cast<const float4*>(&tapw[j * 4]);      // broadcast reads
          const int4 o = *reinterpret_cast<const int4*>(&tapo[j * 4]);
          float* q00 = &img[o.x * 64 + lane];
          float* q01 = &img[o.y * 64 + lane];
          float* q10 = &img[o.z * 64 + lane];
          float* q11 = &img[o.w * 64 + lane];
          // read all four, write nw LAST: an aliased (weight-0) tap writes the old nw value back first
          const float v00 = *q00, v01 = *q01, v10 = *q10, v11 = *q11;
          *q11 = v11 + d * w.w;
          *q10 = v10 + d * w.z;
          *q01 = v01 + d * w.y;
          *q00 = v00 + d * w.x;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  float* out = P.grad[l] + (size_t)b * npix * C + jb.chunk * 64 + lane;
  for (int px = 0; px < npix; ++px) {
    const float v = img[px * 64 + lane];
    if (v != 0.f) atomicAdd(out + (size_t)px * C, v);
  }
}

// Deterministic backward (HOISDF_DETERMINISTIC): a single-wave workgroup OWNS a 16 x 16-pixel tile of one (sample, level,
// 64-channel chunk) - 64 / 16 / 4 / 1 / 1 tiles for the 128^2 ... 8^2 levels - and walks ALL points of the sample in
// order: 64 projections at a time (one per lane), a ballot keeps the points with at least one bilinear tap inside the tile,
// and those are applied in ascending point order to a wave-private LDS image (plain in-order read-modify-write, lane =
// channel).  The tile leaves with plain stores: no atomics anywhere, every sum has a fixed order.
struct DetJob { short level, chunk, tx, ty; };
struct DetArgs { int n_jobs; DetJob job[160]; };
constexpr int DTILE = 16;

__global__ __launch_bounds__(64) void gather_bwd_det_kernel(PyrDev P, ProjArgs a, DetArgs J,
                                                            const float* __restrict__ dfeat, int ldf) {
  __shared__ __attribute__((aligned(16))) float img[(DTILE * DTILE + 1) * 64];   // + one dummy pixel for out-of-tile taps
  __shared__ __attribute__((aligned(16))) float tapw[64 * 4];
  __shared__ __attribute__((aligned(16))) int tapo[64 * 4];
  const int lane = threadIdx.x;
  const int b = blockIdx.y;
  const DetJob jb = J.job[blockIdx.x];
  const int l = jb.level, C = P.C[l], H = P.H[l], W = P.W[l];
  const int tx = jb.tx, ty = jb.ty;
  const int tw = min(DTILE, W - tx), th = min(DTILE, H - ty);
  const int npix = tw * th;
  const int cw = min(64, C - jb.chunk * 64);
  const bool active = lane < cw;
  for (int i = lane; i < (npix + 1) * 64; i += 64) img[i] = 0.f;
  const int ch0 = P.off4[l] * 4 + jb.chunk * 64;
  const int Pn = a.rows_per_sample;
  const float* drow = dfeat + (size_t)b * Pn * ldf + ch0 + lane;
  for (int k0 = 0; k0 < Pn; k0 += 64) {
    bool hit = false;
    if (k0 + lane < Pn) {
      int bb;
      float cam[3], uv[2], g[2];
      project_row(a, (long)b * Pn + k0 + lane, bb, cam, uv, g);
      const Taps t = make_taps(g[0], g[1], W, H);
      const int o[4] = {t.o00, t.o01, t.o10, t.o11};
      const float w[4] = {t.w00, t.w01, t.w10, t.w11};
      int lo[4];
      float lw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lo[i] = npix;                       // dummy pixel, weight 0
        lw[i] = 0.f;
        if (o[i] >= 0) {
          const int y = o[i] / W, x = o[i] - y * W;
          if (x >= tx && x < tx + tw && y >= ty && y < ty + th) {
            lo[i] = (y - ty) * tw + (x - tx);
            lw[i] = w[i];
            hit = true;
          }
        }
      }
      *reinterpret_cast<float4*>(&tapw[lane * 4]) = make_float4(lw[0], lw[1], lw[2], lw[3]);
      *reinterpret_cast<int4*>(&tapo[lane * 4]) = make_int4(lo[0], lo[1], lo[2], lo[3]);
    }
    unsigned long long mask = __ballot(hit);
    __builtin_amdgcn_wave_barrier();
    while (mask) {                                      // ascending point order
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float d = active ? drow[(size_t)(k0 + j) * ldf] : 0.f;
      const float4 w = *reinterpret_cast<const float4*>(&tapw[j * 4]);
      const int4 o = *reinterpret_cast<const int4*>(&tapo[j * 4]);
      // the four taps of one point are four DIFFERENT pixels (or the dummy): sequential read-modify-write is exact
      img[o.x * 64 + lane] += d * w.x;
      img[o.y * 64 + lane] += d * w.y;
      img[o.z * 64 + lane] += d * w.z;
      img[o.w * 64 + lane] += d * w.w;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (active) {
    float* out = P.grad[l] + (size_t)b * H * W * C + jb.chunk * 64 + lane;
    for (int py = 0; py < th; ++py)
      for (int px = 0; px < tw; ++px) out[((size_t)(ty + py) * W + tx + px) * C] = img[(py * tw + px) * 64 + lane];
  }
}

// ---- dense lattice (main/model.py:257-273): sheared, float32 index arithmetic --------------
__device__ __forceinline__ void lattice_point(int idx, int n, float v32, float (&p)[3]) {
  const float fn = (float)n;
  const float a = (float)idx;                 // exact for idx < 2^24
  const float zl = (float)(idx % n);
  const float q = __fdiv_rn(a, fn);           // true (float) division, as int64 / int does in torch
  const float yl = fmodf(q, fn);
  const float xl = fmodf(__fdiv_rn(q, fn), fn);
  p[0] = __fadd_rn(__fmul_rn(xl, v32), -1.f);
  p[1] = __fadd_rn(__fmul_rn(yl, v32), -1.f);
  p[2] = __fadd_rn(__fmul_rn(zl, v32), -1.f);
}

__device__ __forceinline__ bool lattice_keep(const float (&p)[3], const float* c, const float* K,
                                             const float* bb, float scale) {
  float cam[3], q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) cam[i] = __fadd_rn(__fdiv_rn(p[i], scale), c[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = cam[0] * K[i * 3 + 0] + cam[1] * K[i * 3 + 1] + cam[2] * K[i * 3 + 2];
  const float u = __fdiv_rn(q[0], q[2]), v = __fdiv_rn(q[1], q[2]);
  return (u > bb[0]) && (u < bb[2]) && (v > bb[1]) && (v < bb[3]);      // strict (main/model.py:293-300)
}

__global__ __launch_bounds__(256) void lattice_count_kernel(const float* __restrict__ center,
                                                            const float* __restrict__ cam_intr,
                                                            const float* __restrict__ bbox, float scale,
                                                            int n, float v32, int32_t* __restrict__ counts) {
  const int b = blockIdx.y;
  const int total = n * n * n;
  int idx = blockIdx.x * 256 + threadIdx.x;
  bool keep = false;
  if (idx < total) {
    float p[3];
    lattice_point(idx, n, v32, p);
    keep = lattice_keep(p, center + b * 3, cam_intr + b * 9, bbox + b * 4, scale);
  }
  unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counts[b], __popcll(m));
}

// one 1024-thread workgroup per sample walks the lattice in order -> deterministic ascending
// lattice order of the survivors (what boolean indexing yields in the reference).
__global__ __launch_bounds__(1024) void lattice_fill_kernel(const float* __restrict__ center,
                                                            const float* __restrict__ cam_intr,
                                                            const float* __restrict__ bbox, float scale,
                                                            int n, float v32,
                                                            const int32_t* __restrict__ offsets,
                                                            float* __restrict__ points,
                                                            int32_t* __restrict__ sample_idx,
                                                            int32_t* __restrict__ lattice_idx) {
  __shared__ int wave_cnt[16];
  __shared__ int running;
  const int b = blockIdx.x;
  const int total = n * n * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = offsets[b];
  __syncthreads();
  for (int base = 0; base < total; base += 1024) {
    const int idx = base + threadIdx.x;
    float p[3] = {0.f, 0.f, 0.f};
    bool keep = false;
    if (idx < total) {
      lattice_point(idx, n, v32, p);
      keep = lattice_keep(p, center + b * 3, cam_intr + b * 9, bbox + b * 4, scale);
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int pre = running;
    for (int w = 0; w < wave; ++w) pre += wave_cnt[w];
    if (keep) {
      const int dst = pre + __popcll(m & ((1ULL << lane) - 1ULL));
      points[(size_t)dst * 3 + 0] = p[0];
      points[(size_t)dst * 3 + 1] = p[1];
      points[(size_t)dst * 3 + 2] = p[2];
      if (sample_idx) sample_idx[dst] = b;
      if (lattice_idx) lattice_idx[dst] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int s = 0;
      for (int w = 0; w < 16; ++w) s += wave_cnt[w];
      running += s;
    }
    __syncthreads();
  }
}

// Round 6: the same two passes with the lattice cut into CHUNKS of 4096 consecutive indices per sample (64 chunks at 64^3): one 256-thread
// block per (chunk, sample) instead of one 1024-thread block per sample (16 of the 256 CUs at B = 16: 434 us per call) and instead of
// one atomic per WAVE on the sample's counter (4096 adds per address: 134 us per call).  The order of the survivors is the ascending
// lattice order as before (chunks ascending, ascending inside a chunk): outputs bit-identical to the kernels above.
constexpr int LCH = 4096;
// survivors of this block's chunk (all threads return it); keep[i] = candidate chunk_base + 256 i + tid
__device__ __forceinline__ int lattice_chunk_eval(const float* c, const float* K, const float* bb, float scale, int n, float v32, int chunk,
                                                  int total, int* red4, unsigned& keepbits) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float cc[3] = {c[0], c[1], c[2]}, KK[9], bbox[4] = {bb[0], bb[1], bb[2], bb[3]};
#pragma unroll
  for (int i = 0; i < 9; ++i) KK[i] = K[i];
  keepbits = 0u;
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < LCH / 256; ++i) {
    const int idx = chunk * LCH + i * 256 + threadIdx.x;
    bool keep = false;
    if (idx < total) {
      float p[3];
      lattice_point(idx, n, v32, p);
      keep = lattice_keep(p, cc, KK, bbox, scale);
    }
    keepbits |= (keep ? 1u : 0u) << i;
    cnt += keep ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) red4[wave] = cnt;
  __syncthreads();
  return red4[0] + red4[1] + red4[2] + red4[3];
}
__global__ __launch_bounds__(256) void lattice_chunk_count_kernel(const float* __restrict__ center, const float* __restrict__ cam_intr,
                                                                  const float* __restrict__ bbox, float scale, int n, float v32,
                                                                  int32_t* __restrict__ counts, int32_t* __restrict__ chunk_counts) {
  __shared__ int red4[4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  unsigned kb;
  const int s = lattice_chunk_eval(center + b * 3, cam_intr + b * 9, bbox + b * 4, scale, n, v32, chunk, n * n * n, red4, kb);
  if (threadIdx.x == 0) {
    if (chunk_counts) chunk_counts[(size_t)b * gridDim.x + chunk] = s;
    if (counts && s) atomicAdd(&counts[b], s);
  }
}
__global__ __launch_bounds__(256) void lattice_chunk_fill_kernel(const float* __restrict__ center, const float* __restrict__ cam_intr,
                                                                 const float* __restrict__ bbox, float scale, int n, float v32,
                                                                 const int32_t* __restrict__ offsets, const int32_t* __restrict__ chunk_counts,
                                                                 float* __restrict__ points, int32_t* __restrict__ sample_idx,
                                                                 int32_t* __restrict__ lattice_idx) {
  __shared__ int red4[4];
  __shared__ int wcnt[LCH / 256][4];
  __shared__ int start;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // where this chunk's survivors start: the sample's offset + the chunks before it (<= 4096 of them: bins_n <= 256)
  int pre = 0;
  for (int c = threadIdx.x; c < chunk; c += 256) pre += chunk_counts[(size_t)b * nchunk + c];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) pre += __shfl_xor(pre, o, 64);
  if (lane == 0) red4[wave] = pre;
  __syncthreads();
  if (threadIdx.x == 0) start = offsets[b] + red4[0] + red4[1] + red4[2] + red4[3];
  __syncthreads();
  const float* c = center + b * 3;
  const float* K = cam_intr + b * 9;
  const float* bb = bbox + b * 4;
  const int total = n * n * n;
  float pts[LCH / 256][3];
  unsigned long long masks[LCH / 256];
#pragma unroll
  for (int i = 0; i < LCH / 256; ++i) {
    const int idx = chunk * LCH + i * 256 + threadIdx.x;
    bool keep = false;
    pts[i][0] = pts[i][1] = pts[i][2] = 0.f;
    if (idx < total) {
      lattice_point(idx, n, v32, pts[i]);
      keep = lattice_keep(pts[i], c, K, bb, scale);
    }
    masks[i] = __ballot(keep);
    if (lane == 0) wcnt[i][wave] = __popcll(masks[i]);
  }
  __syncthreads();
  int run = start;
#pragma unroll
  for (int i = 0; i < LCH / 256; ++i) {
    int at = run;
    for (int w = 0; w < wave; ++w) at += wcnt[i][w];
    if ((masks[i] >> lane) & 1ULL) {
      const int dst = at + __popcll(masks[i] & ((1ULL << lane) - 1ULL));
      points[(size_t)dst * 3 + 0] = pts[i][0];
      points[(size_t)dst * 3 + 1] = pts[i][1];
      points[(size_t)dst * 3 + 2] = pts[i][2];
      if (sample_idx) sample_idx[dst] = b;
      if (lattice_idx) lattice_idx[dst] = chunk * LCH + i * 256 + threadIdx.x;
    }
    run += wcnt[i][0] + wcnt[i][1] + wcnt[i][2] + wcnt[i][3];
  }
}

static int fill_pyr(PyrDev& P, int n_levels, const int* C, const int* H, const int* W) {
  P.n_levels = n_levels;
  int off = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (C[l] <= 0 || (C[l] & 3) || H[l] <= 0 || W[l] <= 0) return -1;
    P.C[l] = C[l]; P.H[l] = H[l]; P.W[l] = W[l];
    P.off4[l] = off;
    off += C[l] / 4;
  }
  for (int l = n_levels; l <= HOISDF_MAX_LEVELS; ++l) P.off4[l] = off;
  P.C4 = off;
  return 0;
}

static int grid_for_rows(long n_rows) {
  long blocks = (n_rows + 3) / 4;
  if (blocks > 256L * 16) blocks = 256L * 16;
  return (int)blocks;
}

}  // namespace hoisdf

using namespace hoisdf;

static int check_proj(const float* points, long n_rows, int rows_per_sample, const int32_t* sample_idx,
                      const float* center, const float* cam_intr, float scale, const char* who) {
  HOISDF_REQUIRE(points && center && cam_intr, HOISDF_ERR_INVALID, "%s: null pointer", who);
  HOISDF_REQUIRE(n_rows >= 0 && (sample_idx || rows_per_sample > 0) && scale != 0.f, HOISDF_ERR_INVALID,
                 "%s: bad sizes", who);
  return 0;
}

extern "C" int hoisdf_project_gather_fwd(const hoisdf_pyramid* pyr, const float* points,
                                         const int32_t* sample_idx, long n_rows, int rows_per_sample,
                                         const float* center, const float* cam_intr, float scale, int img_h,
                                         int img_w, float* feat, int ldf, float* cam_out, float* uv_out,
                                         void* stream) {
  return project_gather_fwd_mag(pyr, points, sample_idx, n_rows, rows_per_sample, center, cam_intr, scale, img_h, img_w, feat, ldf, cam_out,
                                uv_out, nullptr, stream);
}
int hoisdf::project_gather_fwd_mag(const hoisdf_pyramid* pyr, const float* points, const int32_t* sample_idx, long n_rows,
                                   int rows_per_sample, const float* center, const float* cam_intr, float scale, int img_h, int img_w,
                                   float* feat, int ldf, float* cam_out, float* uv_out, uint32_t* feat_mag, void* stream) {
  HOISDF_REQUIRE(pyr && feat, HOISDF_ERR_INVALID, "project_gather_fwd: null pointer");
  if (int rc = check_proj(points, n_rows, rows_per_sample, sample_idx, center, cam_intr, scale,
                          "project_gather_fwd")) return rc;
  HOISDF_REQUIRE(pyr->n_levels > 0 && pyr->n_levels <= HOISDF_MAX_LEVELS, HOISDF_ERR_INVALID,
                 "project_gather_fwd: n_levels=%d", pyr->n_levels);
  PyrDev P{};
  HOISDF_REQUIRE(fill_pyr(P, pyr->n_levels, pyr->C, pyr->H, pyr->W) == 0, HOISDF_ERR_INVALID,
                 "project_gather_fwd: level channels must be positive multiples of 4");
  for (int l = 0; l < pyr->n_levels; ++l) {
    HOISDF_REQUIRE(pyr->data[l] && ((uintptr_t)pyr->data[l] & 15) == 0, HOISDF_ERR_INVALID,
                   "project_gather_fwd: level %d pointer null or not 16-byte aligned", l);
    P.data[l] = pyr->data[l];
  }
  HOISDF_REQUIRE(ldf >= P.C4 * 4 && (ldf & 3) == 0 && ((uintptr_t)feat & 15) == 0, HOISDF_ERR_INVALID,
                 "project_gather_fwd: feat must be 16-byte aligned with ldf %% 4 == 0 and ldf >= C");
  if (n_rows == 0) return HOISDF_OK;
  ProjArgs a{points, sample_idx, n_rows, rows_per_sample, center, cam_intr, scale,
             (float)(img_w - 1) * 0.5f, (float)(img_h - 1) * 0.5f};
  static int old_form = -1;                       // HOISDF_GATHER_FWD=1: the round-1 loop for every width (A/B runs)
  if (old_form < 0) { const char* e = getenv("HOISDF_GATHER_FWD"); old_form = (e && atoi(e) == 1) ? 1 : 0; }
  if (P.C4 <= 256 && !old_form)
    hipLaunchKernelGGL(gather_fwd4_kernel, dim3(grid_for_rows(n_rows)), dim3(256), 0, as_stream(stream), P, a, feat, ldf, cam_out, uv_out, feat_mag);
  else
    hipLaunchKernelGGL(gather_fwd_kernel, dim3(grid_for_rows(n_rows)), dim3(256), 0, as_stream(stream), P, a, feat, ldf, cam_out, uv_out, feat_mag);
  return check_launch("gather_fwd");
}

extern "C" int hoisdf_project_gather_bwd(const hoisdf_pyramid_grad* dpyr, const float* points,
                                         const int32_t* sample_idx, long n_rows, int rows_per_sample,
                                         const float* center, const float* cam_intr, float scale, int img_h,
                                         int img_w, const float* dfeat, int ldf, void* stream) {
  HOISDF_REQUIRE(dpyr && dfeat, HOISDF_ERR_INVALID, "project_gather_bwd: null pointer");
  if (int rc = check_proj(points, n_rows, rows_per_sample, sample_idx, center, cam_intr, scale,
                          "project_gather_bwd")) return rc;
  HOISDF_REQUIRE(dpyr->n_levels > 0 && dpyr->n_levels <= HOISDF_MAX_LEVELS, HOISDF_ERR_INVALID,
                 "project_gather_bwd: n_levels=%d", dpyr->n_levels);
  PyrDev P{};
  HOISDF_REQUIRE(fill_pyr(P, dpyr->n_levels, dpyr->C, dpyr->H, dpyr->W) == 0, HOISDF_ERR_INVALID,
                 "project_gather_bwd: level channels must be positive multiples of 4");
  for (int l = 0; l < dpyr->n_levels; ++l) {
    HOISDF_REQUIRE(dpyr->data[l], HOISDF_ERR_INVALID, "project_gather_bwd: level %d pointer null", l);
    P.grad[l] = dpyr->data[l];
  }
  HOISDF_REQUIRE(ldf >= P.C4 * 4 && (ldf & 3) == 0 && ((uintptr_t)dfeat & 15) == 0, HOISDF_ERR_INVALID,
                 "project_gather_bwd: dfeat must be 16-byte aligned with ldf %% 4 == 0");
  if (n_rows == 0) return HOISDF_OK;
  ProjArgs a{points, sample_idx, n_rows, rows_per_sample, center, cam_intr, scale,
             (float)(img_w - 1) * 0.5f, (float)(img_h - 1) * 0.5f};
  if (deterministic_mode() && !sample_idx && rows_per_sample > 0 && n_rows % rows_per_sample == 0) {
    // tile owners, no atomics (the pyramid gradient is fully overwritten tile by tile)
    const int B = (int)(n_rows / rows_per_sample);
    DetArgs J{};
    for (int l = 0; l < P.n_levels; ++l)
      for (int ty = 0; ty < P.H[l]; ty += DTILE)
        for (int tx = 0; tx < P.W[l]; tx += DTILE)
          for (int ch = 0; ch < cdiv(P.C[l], 64); ++ch) {
            if (J.n_jobs == 160) {
              hipLaunchKernelGGL(gather_bwd_det_kernel, dim3(J.n_jobs, B), dim3(64), 0, as_stream(stream), P, a, J, dfeat, ldf);
              if (int rc = check_launch("gather_bwd_det")) return rc;
              J.n_jobs = 0;
            }
            J.job[J.n_jobs++] = DetJob{(short)l, (short)ch, (short)tx, (short)ty};
          }
    if (J.n_jobs) {
      hipLaunchKernelGGL(gather_bwd_det_kernel, dim3(J.n_jobs, B), dim3(64), 0, as_stream(stream), P, a, J, dfeat, ldf);
      if (int rc = check_launch("gather_bwd_det")) return rc;
    }
    return HOISDF_OK;
  }
  int skip_mask = 0;
  if (!sample_idx && rows_per_sample > 0 && n_rows % rows_per_sample == 0) {
    const int B = (int)(n_rows / rows_per_sample);
    // group the coarse levels by LDS image size so one launch uses one dynamic-LDS size
    for (int npix_cap : {64, 256}) {
      CoarseArgs J{};
      int max_pix = 0;
      for (int l = 0; l < P.n_levels; ++l) {
        const int npix = P.H[l] * P.W[l];
        if (npix > npix_cap || (npix_cap == 256 && npix <= 64) || (P.C[l] & 63)) continue;
        if (J.n_jobs + P.C[l] / 64 > 96) continue;
        for (int ch = 0; ch < P.C[l] / 64; ++ch) J.job[J.n_jobs++] = CoarseJob{l, ch};
        skip_mask |= 1 << l;
        if (npix > max_pix) max_pix = npix;
      }
      if (J.n_jobs == 0) continue;
      hipLaunchKernelGGL(gather_bwd_coarse_kernel, dim3(J.n_jobs, B, COARSE_SLICES), dim3(64),
                         (size_t)max_pix * 64 * 4 + 64 * 8 * 4, as_stream(stream), P, a, J, dfeat, ldf);
      if (int rc = check_launch("gather_bwd_coarse")) return rc;
    }
  }
  hipLaunchKernelGGL(gather_bwd_kernel, dim3(grid_for_rows(n_rows)), dim3(256), 0, as_stream(stream), P, a,
                     dfeat, ldf, skip_mask);
  return check_launch("gather_bwd");
}

// HOISDF_LATTICE=1: the round-1 kernels (one block per sample; A/B runs)
static bool lattice_old_form() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("HOISDF_LATTICE"); v = (e && atoi(e) == 1) ? 1 : 0; }
  return v == 1;
}

extern "C" int hoisdf_lattice_count(const float* center, const float* cam_intr, const float* bbox, float scale,
                                    int bins_n, int B, int32_t* counts, void* stream) {
  HOISDF_REQUIRE(center && cam_intr && bbox && counts, HOISDF_ERR_INVALID, "lattice_count: null pointer");
  HOISDF_REQUIRE(bins_n >= 2 && bins_n <= 256 && B > 0, HOISDF_ERR_INVALID, "lattice_count: bins_n=%d B=%d",
                 bins_n, B);
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(counts, 0, sizeof(int32_t) * B, st) != hipSuccess) {
    set_error("lattice_count: memset failed");
    return HOISDF_ERR_LAUNCH;
  }
  const int total = bins_n * bins_n * bins_n;
  const float v32 = (float)(2.0 / (double)(bins_n - 1));
  if (lattice_old_form())
    hipLaunchKernelGGL(lattice_count_kernel, dim3(cdiv(total, 256), B), dim3(256), 0, st, center, cam_intr, bbox, scale, bins_n, v32, counts);
  else
    hipLaunchKernelGGL(lattice_chunk_count_kernel, dim3(cdiv(total, LCH), B), dim3(256), 0, st, center, cam_intr, bbox, scale, bins_n, v32, counts,
                       (int32_t*)nullptr);
  return check_launch("lattice_count");
}

extern "C" int hoisdf_lattice_fill(const float* center, const float* cam_intr, const float* bbox, float scale,
                                   int bins_n, int B, const int32_t* offsets, float* points,
                                   int32_t* sample_idx, int32_t* lattice_idx, void* stream) {
  HOISDF_REQUIRE(center && cam_intr && bbox && offsets && points, HOISDF_ERR_INVALID,
                 "lattice_fill: null pointer");
  HOISDF_REQUIRE(bins_n >= 2 && bins_n <= 256 && B > 0, HOISDF_ERR_INVALID, "lattice_fill: bins_n=%d B=%d",
                 bins_n, B);
  const float v32 = (float)(2.0 / (double)(bins_n - 1));
  hipStream_t st = as_stream(stream);
  const int nchunk = cdiv((long)bins_n * bins_n * bins_n, LCH);
  // per-(sample, chunk) survivor counts: stream-ordered library scratch (the words live until the next call on this stream)
  int32_t* cc = lattice_old_form() ? nullptr : reinterpret_cast<int32_t*>(mag_scratch(st, (long)B * nchunk));
  if (!cc) {
    hipLaunchKernelGGL(lattice_fill_kernel, dim3(B), dim3(1024), 0, st, center, cam_intr, bbox, scale, bins_n, v32, offsets, points, sample_idx,
                       lattice_idx);
    return check_launch("lattice_fill");
  }
  hipLaunchKernelGGL(lattice_chunk_count_kernel, dim3(nchunk, B), dim3(256), 0, st, center, cam_intr, bbox, scale, bins_n, v32, (int32_t*)nullptr, cc);
  hipLaunchKernelGGL(lattice_chunk_fill_kernel, dim3(nchunk, B), dim3(256), 0, st, center, cam_intr, bbox, scale, bins_n, v32, offsets, cc, points,
                     sample_idx, lattice_idx);
  return check_launch("lattice_fill");
}
