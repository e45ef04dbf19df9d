// (f4) encoder-side fusion: BatchNorm2d (+ residual add) (+ ReLU) of a channels_last map as two HBM passes per direction.
// reference: the BatchNorm2d -> ReLU pairs and the `out += identity; relu(out)` tails of common/nets/resnet.py (torchvision
// Bottleneck / BasicBlock as the reference imports them), common/nets/layer.py:23-63 (make_conv_layers / make_deconv_layers:
// Conv -> BatchNorm2d -> ReLU), i.e. torch.nn.functional.batch_norm(training) [+ add] + relu.  The convolutions stay MIOpen's.
//
// A channels_last (N, C, H, W) map IS a row-major [M = N H W][C] matrix (row stride ld >= C: a channel slice of a concatenation
// keeps the parent's stride).  Every pass is HBM-bound streaming; what this file removes against the library sequence
// (MIOpen BatchNorm: mean/variance, final, normalise | ATen add | ATen clamp; backward: ATen threshold, MIOpen dscale/dbias, final,
// dx) are whole passes over the map:
//   forward   stats pass (1 read) + apply pass (1-2 reads, 1 write, + 1 BIT per element: the ReLU sign map)      [was 5-8 passes' worth]
//   backward  reduce pass (dy, x, bits) + dx pass (dy, x, bits -> dx [, d residual])                              [was 8]
// A thread owns 8 consecutive channels (two 16-byte loads per row; its byte of the sign map), four rows in flight per thread.
// Statistics: per-thread f32 sums of (x - x[0][c]) and its square (shifted: no cancellation for |mean| >> sigma), per-block
// partials, combined in f64 IN SLICE ORDER by a finishing kernel of one block per channel chunk (queued by the same C call) -
// bit-reproducible run to run, no float atomics.  (A last-block-to-arrive fold inside the pass was 6x slower than the pass: the
// agent-scope release fence every block needs writes the XCD's whole L2 back.)  Same for the backward's column sums.
#include "common.h"

namespace hoisdf {
namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BN_T = 256;

struct Geom { int tpr, rpi; };     // threads per row (C / 8), rows per block iteration: the streaming passes (whole rows per block)
__host__ __device__ __forceinline__ Geom geom(int C) { Geom g; g.tpr = C >> 3; g.rpi = BN_T / g.tpr; return g; }
// the reducing passes: a block owns a CHUNK of 64 (32 / 16 / 8: the largest that divides C) channels of a row slice - 8 threads per
// row, 32 rows per iteration - so that wide maps with few rows still make hundreds of blocks and the partials of a chunk
// (slices x 2 x chunk floats) stay small enough for ONE block of the finishing kernel to fold
__host__ __device__ __forceinline__ int chunk_of(int C) { return C % 64 == 0 ? 64 : C % 32 == 0 ? 32 : C % 16 == 0 ? 16 : 8; }

__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// the block's 16 per-thread sums (s[0..7] -> plane 0, s[8..15] -> plane 1) -> part[chunk][slice][plane][chunk channels]; row groups added in order
__device__ __forceinline__ void block_partials(const float (&s)[16], float* red, int tpr, int rpi, int chunk, float* __restrict__ part) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 16; ++j) red[j * BN_T + tid] = s[j];
  __syncthreads();
  float* dst = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * chunk;
  for (int o = tid; o < 2 * chunk; o += BN_T) {        // output o = plane * chunk + channel
    const int pl = o / chunk, ch = o - pl * chunk;
    const int j = pl * 8 + (ch & 7), t0 = ch >> 3;
    float a = 0.f;
    for (int g = 0; g < rpi; ++g) a += red[j * BN_T + g * tpr + t0];
    dst[o] = a;
  }
}
// (finishing kernels: one block per chunk) f64 sums over the S slices' partials of this chunk, in slice order (8+ groups of slices in
// parallel, combined in order): thread ch < chunk ends with the two plane sums of channel ch.  All threads call.
__device__ __forceinline__ void sum_partials(const float* __restrict__ part, int S, int chunk, double* dred, double& a1, double& a2) {
  const int tid = threadIdx.x;
  const int lanes = chunk >> 1;                         // 16-byte lanes of one slice's [2][chunk] floats
  const int G = BN_T / lanes, lane = tid % lanes, g = tid / lanes;
  const f32x4* src = reinterpret_cast<const f32x4*>(part + (size_t)blockIdx.x * S * 2 * chunk) + lane;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  int sl = g;
  for (; sl + 3 * G < S; sl += 4 * G) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(sl + u * G) * lanes];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += (double)v[u][i];
  }
  for (; sl < S; sl += G) {
    const f32x4 v = src[(size_t)sl * lanes];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += (double)v[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) dred[(g * lanes + lane) * 4 + i] = acc[i];
  __syncthreads();
  a1 = 0.0; a2 = 0.0;
  if (tid < chunk) {
    const int l1 = tid >> 2, l2 = (chunk >> 2) + (tid >> 2), i = tid & 3;
    for (int g2 = 0; g2 < G; ++g2) { a1 += dred[(g2 * lanes + l1) * 4 + i]; a2 += dred[(g2 * lanes + l2) * 4 + i]; }
  }
}

__global__ __launch_bounds__(BN_T) void bn_stats_kernel(const float* __restrict__ x, long ldx, long M, int C, long rows_per_block,
                                                        float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float red[16 * BN_T];
  const int chunk = chunk_of(C), tpr = chunk >> 3, rpi = BN_T / tpr;
  const int tid = threadIdx.x, cg = tid % tpr, rg = tid / tpr;
  const int c0 = blockIdx.y * chunk + 8 * cg;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float s[16], sh[8];
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = 0.f;
  ld8(x + c0, sh);
  {
    const long step = rpi;
    long r = r0 + rg;
    for (; r + 3 * step < r1; r += 4 * step) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld8(x + (r + u * step) * ldx + c0, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[u][j] - sh[j]; s[j] += d; s[8 + j] = __builtin_fmaf(d, d, s[8 + j]); }
    }
    for (; r < r1; r += step) {
      float v[8];
      ld8(x + r * ldx + c0, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[j] - sh[j]; s[j] += d; s[8 + j] = __builtin_fmaf(d, d, s[8 + j]); }
    }
  }
  block_partials(s, red, tpr, rpi, chunk, part);
}
__global__ __launch_bounds__(BN_T) void bn_stats_finish_kernel(const float* __restrict__ x, const float* __restrict__ part, int S, long M, int C,
                                                               float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                                                               float* __restrict__ run_var, float momentum, float eps) {
  __shared__ __attribute__((aligned(16))) double dred[4 * BN_T];
  const int chunk = chunk_of(C), tid = threadIdx.x;
  double a1, a2;
  sum_partials(part, S, chunk, dred, a1, a2);
  if (tid < chunk) {
    const int c = blockIdx.x * chunk + tid;
    const double m1 = a1 / (double)M;
    const double mu = (double)x[c] + m1;
    double var = a2 / (double)M - m1 * m1;
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * mu);
    if (run_var) run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * (M > 1 ? var * (double)M / (double)(M - 1) : var));
  }
}

// y = relu?((x - mean) * (gamma * invstd) + beta (+ res)); bits: the sign map of y (one byte per thread and row), null = not wanted
template <bool RES, bool RELU>
__global__ __launch_bounds__(BN_T) void bn_apply_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ res, long ldr,
                                                            const float* __restrict__ mean, const float* __restrict__ stat2, int is_var, float eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, uint8_t* __restrict__ bits, long M, int C) {
  const Geom gm = geom(C);
  const int tid = threadIdx.x, cg = tid % gm.tpr, rg = tid / gm.tpr;
  if (rg >= gm.rpi) return;
  float mu[8], k[8], b[8];
  ld8(mean + 8 * cg, mu); ld8(stat2 + 8 * cg, k);
  if (gamma) { float g[8]; ld8(gamma + 8 * cg, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = (is_var ? 1.f / sqrtf(k[j] + eps) : k[j]) * g[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = is_var ? 1.f / sqrtf(k[j] + eps) : k[j];
  }
  if (beta) ld8(beta + 8 * cg, b);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = 0.f;
  }
  const long step = (long)gridDim.x * gm.rpi;
  for (long r = (long)blockIdx.x * gm.rpi + rg; r < M; r += 4 * step) {
    float v[4][8], q[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + u * step < M) { ld8(x + (r + u * step) * ldx + 8 * cg, v[u]); if (RES) ld8(res + (r + u * step) * ldr + 8 * cg, q[u]); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + u * step >= M) break;
      unsigned m = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = __builtin_fmaf(v[u][j] - mu[j], k[j], b[j]);
        if (RES) t += q[u][j];
        if (RELU) { m |= (t > 0.f ? 1u : 0u) << j; t = t > 0.f ? t : 0.f; }
        v[u][j] = t;
      }
      st8(y + (r + u * step) * (long)C + 8 * cg, v[u]);
      if (RELU && bits) bits[(r + u * step) * gm.tpr + cg] = (uint8_t)m;
    }
  }
}

// column sums of g = dy * sign and of g * (x - mean) per slice; the finishing kernel -> dbeta, dgamma, and the two per-channel
// coefficients of the dx pass
template <bool RELU>
__global__ __launch_bounds__(BN_T) void bn_bwd_reduce_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                             const uint8_t* __restrict__ bits, const float* __restrict__ mean,
                                                             long M, int C, long rows_per_block, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float red[16 * BN_T];
  const int chunk = chunk_of(C), tpr = chunk >> 3, rpi = BN_T / tpr;
  const int tid = threadIdx.x, cg = tid % tpr, rg = tid / tpr;
  const int c0 = blockIdx.y * chunk + 8 * cg, b0 = c0 >> 3, ldb = C >> 3;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float s[16], mu[8];
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = 0.f;
  ld8(mean + c0, mu);
  {
    const long step = rpi;
    for (long r = r0 + rg; r < r1; r += 4 * step) {
      float g[4][8], v[4][8]; unsigned m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u * step < r1) {
          ld8(dy + (r + u * step) * lddy + c0, g[u]); ld8(x + (r + u * step) * ldx + c0, v[u]);
          m[u] = RELU ? bits[(r + u * step) * ldb + b0] : 0xffu;
        }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r + u * step >= r1) break;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gg = (m[u] >> j) & 1u ? g[u][j] : 0.f;
          s[j] += gg; s[8 + j] = __builtin_fmaf(gg, v[u][j] - mu[j], s[8 + j]);
        }
      }
    }
  }
  block_partials(s, red, tpr, rpi, chunk, part);
}
__global__ __launch_bounds__(BN_T) void bn_bwd_finish_kernel(const float* __restrict__ part, int S, long M, int C, const float* __restrict__ invstd,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
  __shared__ __attribute__((aligned(16))) double dred[4 * BN_T];
  const int chunk = chunk_of(C), tid = threadIdx.x;
  double a1, a2;
  sum_partials(part, S, chunk, dred, a1, a2);
  if (tid < chunk) {
    const int c = blockIdx.x * chunk + tid;
    const double is = (double)invstd[c];
    if (dbeta) dbeta[c] = (float)a1;
    if (dgamma) dgamma[c] = (float)(a2 * is);
    coef[c] = (float)(a1 / (double)M);
    coef[C + c] = (float)(a2 * is * is / (double)M);
  }
}

// dx = gamma invstd (g - mean(g) - (x - mean) invstd^2 mean(g (x - mean))); d residual = g
template <bool RES, bool RELU>
__global__ __launch_bounds__(BN_T) void bn_bwd_dx_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                         const uint8_t* __restrict__ bits, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ coef, float* __restrict__ dx, float* __restrict__ dres,
                                                         long M, int C) {
  const Geom gm = geom(C);
  const int tid = threadIdx.x, cg = tid % gm.tpr, rg = tid / gm.tpr;
  if (rg >= gm.rpi) return;
  float mu[8], k[8], ca[8], cb[8];
  ld8(mean + 8 * cg, mu); ld8(invstd + 8 * cg, k); ld8(coef + 8 * cg, ca); ld8(coef + C + 8 * cg, cb);
  if (gamma) { float g[8]; ld8(gamma + 8 * cg, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] *= g[j];
  }
  const long step = (long)gridDim.x * gm.rpi;
  for (long r = (long)blockIdx.x * gm.rpi + rg; r < M; r += 4 * step) {
    float g[4][8], v[4][8]; unsigned m[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + u * step < M) {
        ld8(dy + (r + u * step) * lddy + 8 * cg, g[u]); ld8(x + (r + u * step) * ldx + 8 * cg, v[u]);
        m[u] = RELU ? bits[(r + u * step) * gm.tpr + cg] : 0xffu;
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + u * step >= M) break;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = (m[u] >> j) & 1u ? g[u][j] : 0.f;
        g[u][j] = gg;
        v[u][j] = k[j] * (gg - ca[j] - (v[u][j] - mu[j]) * cb[j]);
      }
      st8(dx + (r + u * step) * (long)C + 8 * cg, v[u]);
      if (RES) st8(dres + (r + u * step) * (long)C + 8 * cg, g[u]);
    }
  }
}

inline bool shape_ok(long M, int C, long ld) { return M > 0 && C >= 8 && C % 8 == 0 && C <= 8 * BN_T && ld >= C && ld % 4 == 0; }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// row slices of the reducing passes: slices x chunks ~ 1024 blocks, <= 256 slices (what one block folds), >= 4 iterations per block
inline void reduce_grid(long M, int C, long& rows_per_block, int& slices, int& chunks) {
  const int chunk = chunk_of(C), rpi = BN_T / (chunk >> 3);
  chunks = C / chunk;
  int smax = 1024 / chunks;
  smax = smax < 16 ? 16 : smax > 256 ? 256 : smax;
  const long unit = (long)rpi * 4;
  const long k = (M + unit * smax - 1) / (unit * smax);
  rows_per_block = unit * (k < 1 ? 1 : k);
  slices = cdiv(M, rows_per_block);
}
inline int stream_grid(long M, int C) {
  const Geom gm = geom(C);
  const long it = (M + gm.rpi - 1) / gm.rpi;          // block iterations in all
  long g = (it + 3) / 4;                              // four rows per thread and pass
  return (int)(g < 1 ? 1 : g > 4096 ? 4096 : g);
}
}  // namespace
}  // namespace hoisdf

using namespace hoisdf;

extern "C" long hoisdf_bn_workspace_floats(long M, int C) {
  if (!shape_ok(M, C, C)) return 0;
  long rpb; int slices, chunks;
  reduce_grid(M, C, rpb, slices, chunks);
  return (long)slices * 2 * C + 2L * C;               // partials | the dx pass's two coefficient rows
}

extern "C" int hoisdf_bn_stats(const float* x, long ldx, long M, int C, float* mean, float* invstd, float* running_mean, float* running_var,
                               float momentum, float eps, float* workspace, long workspace_floats, void* stream) {
  HOISDF_REQUIRE(x && mean && invstd && workspace, HOISDF_ERR_INVALID, "bn_stats: null pointer");
  HOISDF_REQUIRE(shape_ok(M, C, ldx) && al16(x) && al16(mean) && al16(invstd), HOISDF_ERR_INVALID,
                 "bn_stats: M=%ld C=%d ld=%ld (C %% 8 == 0, C <= %d, 16-byte aligned rows)", M, C, ldx, 8 * BN_T);
  HOISDF_REQUIRE(workspace_floats >= hoisdf_bn_workspace_floats(M, C), HOISDF_ERR_WORKSPACE, "bn_stats: workspace of %ld floats, need %ld",
                 workspace_floats, hoisdf_bn_workspace_floats(M, C));
  long rpb; int slices, chunks;
  reduce_grid(M, C, rpb, slices, chunks);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(slices, chunks), dim3(BN_T), 0, as_stream(stream), x, ldx, M, C, rpb, workspace);
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(chunks), dim3(BN_T), 0, as_stream(stream), x, workspace, slices, M, C, mean, invstd, running_mean,
                     running_var, momentum, eps);
  return check_launch("bn_stats");
}

extern "C" int hoisdf_bn_apply_fwd(const float* x, long ldx, const float* residual, long ldr, const float* mean, const float* invstd_or_var,
                                   int second_is_variance, float eps, const float* gamma, const float* beta, int relu, float* y,
                                   uint8_t* sign_bits, long M, int C, void* stream) {
  HOISDF_REQUIRE(x && mean && invstd_or_var && y, HOISDF_ERR_INVALID, "bn_apply_fwd: null pointer");
  HOISDF_REQUIRE(shape_ok(M, C, ldx) && (!residual || (shape_ok(M, C, ldr) && al16(residual))) && al16(x) && al16(y) && al16(mean) &&
                     al16(invstd_or_var) && (!gamma || al16(gamma)) && (!beta || al16(beta)),
                 HOISDF_ERR_INVALID, "bn_apply_fwd: M=%ld C=%d ld=%ld (C %% 8 == 0, C <= %d, 16-byte aligned rows)", M, C, ldx, 8 * BN_T);
  const dim3 grid(stream_grid(M, C)), block(BN_T);
  hipStream_t st = as_stream(stream);
#define BN_FWD(RES_, RELU_) hipLaunchKernelGGL((bn_apply_fwd_kernel<RES_, RELU_>), grid, block, 0, st, x, ldx, residual, ldr, mean, invstd_or_var, \
                                               second_is_variance, eps, gamma, beta, y, sign_bits, M, C)
  if (residual) { if (relu) BN_FWD(true, true); else BN_FWD(true, false); }
  else { if (relu) BN_FWD(false, true); else BN_FWD(false, false); }
#undef BN_FWD
  return check_launch("bn_apply_fwd");
}

extern "C" int hoisdf_bn_bwd(const float* dy, long lddy, const float* x, long ldx, const uint8_t* sign_bits, const float* mean,
                             const float* invstd, const float* gamma, float* dx, float* d_residual, float* dgamma, float* dbeta, long M, int C,
                             float* workspace, long workspace_floats, void* stream) {
  HOISDF_REQUIRE(dy && x && mean && invstd && dx && workspace, HOISDF_ERR_INVALID, "bn_bwd: null pointer");
  HOISDF_REQUIRE(shape_ok(M, C, ldx) && shape_ok(M, C, lddy) && al16(dy) && al16(x) && al16(dx) && al16(mean) && al16(invstd) &&
                     (!gamma || al16(gamma)) && (!d_residual || al16(d_residual)) && al16(workspace),
                 HOISDF_ERR_INVALID, "bn_bwd: M=%ld C=%d ldx=%ld lddy=%ld (C %% 8 == 0, C <= %d, 16-byte aligned rows)", M, C, ldx, lddy, 8 * BN_T);
  HOISDF_REQUIRE(workspace_floats >= hoisdf_bn_workspace_floats(M, C), HOISDF_ERR_WORKSPACE, "bn_bwd: workspace of %ld floats, need %ld",
                 workspace_floats, hoisdf_bn_workspace_floats(M, C));
  long rpb; int slices, chunks;
  reduce_grid(M, C, rpb, slices, chunks);
  float* coef = workspace + (long)slices * 2 * C;
  hipStream_t st = as_stream(stream);
  if (sign_bits) hipLaunchKernelGGL((bn_bwd_reduce_kernel<true>), dim3(slices, chunks), dim3(BN_T), 0, st, dy, lddy, x, ldx, sign_bits, mean, M, C, rpb,
                                    workspace);
  else hipLaunchKernelGGL((bn_bwd_reduce_kernel<false>), dim3(slices, chunks), dim3(BN_T), 0, st, dy, lddy, x, ldx, sign_bits, mean, M, C, rpb, workspace);
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(chunks), dim3(BN_T), 0, st, workspace, slices, M, C, invstd, dgamma, dbeta, coef);
  if (int rc = check_launch("bn_bwd (reduce)")) return rc;
  const dim3 grid(stream_grid(M, C)), block(BN_T);
#define BN_DX(RES_, RELU_) hipLaunchKernelGGL((bn_bwd_dx_kernel<RES_, RELU_>), grid, block, 0, st, dy, lddy, x, ldx, sign_bits, mean, invstd, gamma, coef, \
                                              dx, d_residual, M, C)
  if (d_residual) { if (sign_bits) BN_DX(true, true); else BN_DX(true, false); }
  else { if (sign_bits) BN_DX(false, true); else BN_DX(false, false); }
#undef BN_DX
  return check_launch("bn_bwd (dx)");
}
