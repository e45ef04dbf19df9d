// a15: the scalar point losses of the step as HIP kernels.
//   kind 0  L1 of the (already clamped) SDF prediction against the ground-truth SDF clamped to +-clamp:
//           common/nets/loss.py:64-78 (SepSDFLoss, L1Loss(mean)) with the target clamp of main/model.py:393-400;
//   kind 1  SmoothL1 (beta = 1): main/model.py:35-36,656-662 (obj_rot / obj_trans, mean over (L, B, P, 3) against a (B, 3)
//           target broadcast over depth and points) and common/nets/loss.py:57-59 (loss_all_joint_3d: joints * 1000 against
//           the (B, J, 3) ground truth broadcast over depth).
// Element i of pred [n] is compared with target[((i / (rep * C)) % Bt) * C + i % C]: a (Bt, C) target broadcast over leading
// dimensions and `rep` repeats between Bt and C (sdf: Bt = n, rep = C = 1).  HBM-bound, n <= a few 100 k: up to 128 blocks each
// reduce a contiguous slice in a fixed order into partials[block]; the second launch (one wave) adds the partials in block
// order, so the scalar is bit-reproducible run to run.  loss[0] = out_scale * sum (out_scale = 1 / n for the mean).
#include "common.h"

namespace hoisdf {
namespace {
constexpr int PL_THREADS = 256, PL_PER_BLOCK = 4096, PL_MAX_BLOCKS = 128;

__device__ __forceinline__ float pl_target(const float* __restrict__ target, long i, long repC, int C, long Bt, float clamp) {
  const long b = (i / repC) % Bt;
  float t = target[b * C + (int)(i % C)];
  if (clamp > 0.f) t = fminf(fmaxf(t, -clamp), clamp);
  return t;
}

__global__ __launch_bounds__(PL_THREADS) void point_loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                                    long n, long repC, int C, long Bt, int kind, float clamp,
                                                                    float pred_scale, long per_block, float* __restrict__ partials) {
  __shared__ float red[PL_THREADS / 64];
  const long beg = (long)blockIdx.x * per_block, end = beg + per_block < n ? beg + per_block : n;
  float s = 0.f;
  for (long i = beg + threadIdx.x; i < end; i += PL_THREADS) {
    const float d = pred[i] * pred_scale - pl_target(target, i, repC, C, Bt, clamp);
    const float a = fabsf(d);
    s += kind == 0 ? a : (a < 1.f ? 0.5f * d * d : a - 0.5f);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = red[0];
#pragma unroll
    for (int w = 1; w < PL_THREADS / 64; ++w) t += red[w];
    partials[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(64) void point_loss_finish_kernel(const float* __restrict__ partials, int nblk, float out_scale,
                                                               float* __restrict__ loss) {
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int b = 0; b < nblk; ++b) t += partials[b];
    loss[0] = t * out_scale;
  }
}

__global__ __launch_bounds__(PL_THREADS) void point_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                                    long n, long repC, int C, long Bt, int kind, float clamp,
                                                                    float pred_scale, float out_scale, const float* __restrict__ g,
                                                                    float* __restrict__ dpred) {
  const long i = (long)blockIdx.x * PL_THREADS + threadIdx.x;
  if (i >= n) return;
  const float d = pred[i] * pred_scale - pl_target(target, i, repC, C, Bt, clamp);
  // torch: sign(0) = 0 for L1; SmoothL1: d inside (-1, 1), sign(d) outside
  float ds = kind == 0 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : (fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f));
  dpred[i] = ds * (g[0] * out_scale * pred_scale);
}

inline int pl_blocks(long n) {
  long b = (n + PL_PER_BLOCK - 1) / PL_PER_BLOCK;
  return (int)(b < 1 ? 1 : (b > PL_MAX_BLOCKS ? PL_MAX_BLOCKS : b));
}
}  // namespace
}  // namespace hoisdf

using namespace hoisdf;

extern "C" int hoisdf_point_loss_blocks(long n) { return n > 0 ? pl_blocks(n) : 0; }

extern "C" int hoisdf_point_loss_fwd(const float* pred, const float* target, long n, long rep, int C, long Bt, int kind,
                                     float clamp, float pred_scale, float out_scale, float* partials, float* loss, void* stream) {
  HOISDF_REQUIRE(pred && target && partials && loss, HOISDF_ERR_INVALID, "point_loss_fwd: null pointer");
  HOISDF_REQUIRE(n > 0 && rep > 0 && C > 0 && Bt > 0 && (kind == 0 || kind == 1), HOISDF_ERR_INVALID,
                 "point_loss_fwd: bad sizes n=%ld rep=%ld C=%d Bt=%ld kind=%d", n, rep, C, Bt, kind);
  const int nblk = pl_blocks(n);
  const long per_block = ((n + nblk - 1) / nblk + PL_THREADS - 1) / PL_THREADS * PL_THREADS;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(point_loss_fwd_kernel, dim3(nblk), dim3(PL_THREADS), 0, st, pred, target, n, rep * C, C, Bt, kind, clamp,
                     pred_scale, per_block, partials);
  hipLaunchKernelGGL(point_loss_finish_kernel, dim3(1), dim3(64), 0, st, partials, nblk, out_scale, loss);
  return check_launch("point_loss_fwd");
}

extern "C" int hoisdf_point_loss_bwd(const float* pred, const float* target, long n, long rep, int C, long Bt, int kind,
                                     float clamp, float pred_scale, float out_scale, const float* g, float* dpred, void* stream) {
  HOISDF_REQUIRE(pred && target && g && dpred, HOISDF_ERR_INVALID, "point_loss_bwd: null pointer");
  HOISDF_REQUIRE(n > 0 && rep > 0 && C > 0 && Bt > 0 && (kind == 0 || kind == 1), HOISDF_ERR_INVALID, "point_loss_bwd: bad sizes");
  hipLaunchKernelGGL(point_loss_bwd_kernel, dim3((unsigned)((n + PL_THREADS - 1) / PL_THREADS)), dim3(PL_THREADS), 0,
                     as_stream(stream), pred, target, n, rep * C, C, Bt, kind, clamp, pred_scale, out_scale, g, dpred);
  return check_launch("point_loss_bwd");
}
