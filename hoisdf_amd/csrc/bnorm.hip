// (f4, encoder side) training-mode BatchNorm2d fused with its ReLU and the residual add of a ResNet block, on
// channels-last activations viewed as rows [R = N*H*W][C] (C contiguous):
//     y = relu?( gamma * (x - mean_c) * invstd_c + beta  (+ residual) )
// reference: nn.BatchNorm2d + nn.ReLU(inplace) (+ `out += identity`) in common/nets/resnet.py / layer.py:23-63 (the CNN stays
// PyTorch / MIOpen for the convolutions; MIOpen runs BN as 3 kernels forward + 3 backward and PyTorch adds one pass each for
// ReLU, ReLU backward, the residual add and its gradient accumulation - this file does the same arithmetic in 3 + 3 passes).
// Statistics: per-block shifted sums (shift = the block's first row) merged in block order with Chan's formula - order-fixed,
// no atomics; biased variance for the normalisation, unbiased for running_var (momentum update), as torch does.
#include "common.h"

namespace hoisdf {
namespace {

constexpr int BN_T = 256;

// A block reduces a [rows slab] x [column chunk] tile: CW = min(C/4, 64) float4 columns, 256 / CW thread-rows; grid =
// (row slabs, C/4 / CW column chunks), sized for >= ~2048 blocks also when R is small and C large (layer4: 2048 x 2048).
struct BnGeom { int C4, CW, tpr, ncol; };
__host__ __device__ inline BnGeom bn_geom(int C) {
  BnGeom g;
  g.C4 = C / 4;
  g.CW = g.C4 < 64 ? g.C4 : 64;
  g.tpr = BN_T / g.CW;
  g.ncol = g.C4 / g.CW;
  return g;
}

// per block: rows [blockIdx.x * rows_per_block, ...): shifted sum / sum of squares per channel
__global__ __launch_bounds__(BN_T) void bn_stats_kernel(const float* __restrict__ x, long R, int C, int rows_per_block,
                                                        float* __restrict__ part /* [nblk][3][C]: shift, s1, s2 */) {
  const BnGeom g = bn_geom(C);
  const int tid = threadIdx.x;
  const int c4 = tid % g.CW, tr = tid / g.CW;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  __shared__ float4 red[2][BN_T];
  const int col = (blockIdx.y * g.CW + c4) * 4;
  const float4 k = *reinterpret_cast<const float4*>(x + (size_t)r0 * C + col);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  for (long r = r0 + tr; r < r1; r += g.tpr) {
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + col);
    const float a = v.x - k.x, b = v.y - k.y, c = v.z - k.z, d = v.w - k.w;
    s1.x += a; s1.y += b; s1.z += c; s1.w += d;
    s2.x += a * a; s2.y += b * b; s2.z += c * c; s2.w += d * d;
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  if (tr == 0) {
    for (int j = 1; j < g.tpr; ++j) {                  // fixed order over the thread-rows
      const float4 a = red[0][c4 + j * g.CW], b = red[1][c4 + j * g.CW];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
    }
    float* p = part + (size_t)blockIdx.x * 3 * C;
    *reinterpret_cast<float4*>(p + col) = k;
    *reinterpret_cast<float4*>(p + C + col) = s1;
    *reinterpret_cast<float4*>(p + 2 * C + col) = s2;
  }
}

// 16 channels per block x 16 partial lanes: lane l merges the block partials l, l + 16, ... (Chan's formula, ascending), the
// 16 lane results are merged in lane order by lane 0 - a fixed order, and 64 instead of 1024 serial steps per channel.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblk, long R, int C,
                                                          int rows_per_block, float eps, float momentum,
                                                          float* __restrict__ mean, float* __restrict__ invstd,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var) {
  __shared__ double sn[16][17], sm[16][17], sM[16][17];
  const int cl = threadIdx.x & 15, l = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double n = 0.0, m = 0.0, M2 = 0.0;
  if (c < C) {
    for (int b = l; b < nblk; b += 16) {
      const float* p = part + (size_t)b * 3 * C;
      const long rb0 = (long)b * rows_per_block;
      const double nb = (double)((rb0 + rows_per_block < R ? rb0 + rows_per_block : R) - rb0);
      const double s1 = p[C + c], s2 = p[2 * C + c];
      const double mb = (double)p[c] + s1 / nb, M2b = s2 - s1 * s1 / nb;
      const double d = mb - m, nn = n + nb;
      m += d * nb / nn;
      M2 += M2b + d * d * n * nb / nn;
      n = nn;
    }
  }
  sn[cl][l] = n; sm[cl][l] = m; sM[cl][l] = M2;
  __syncthreads();
  if (l == 0 && c < C) {
    for (int j = 1; j < 16; ++j) {
      const double nb = sn[cl][j];
      if (nb > 0.0) {
        const double d = sm[cl][j] - m, nn = n + nb;
        m += d * nb / nn;
        M2 += sM[cl][j] + d * d * n * nb / nn;
        n = nn;
      }
    }
    const double var = M2 / n;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(n > 1.0 ? M2 / (n - 1.0) : var);
    }
  }
}

__global__ __launch_bounds__(BN_T) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, long R, int C, int relu) {
  const long n4 = R * (long)(C / 4);
  for (long i = (long)blockIdx.x * BN_T + threadIdx.x; i < n4; i += (long)gridDim.x * BN_T) {
    const int col = (int)(i % (C / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + col), is = *reinterpret_cast<const float4*>(invstd + col);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + col), be = *reinterpret_cast<const float4*>(beta + col);
    float4 o = make_float4((v.x - mu.x) * is.x * ga.x + be.x, (v.y - mu.y) * is.y * ga.y + be.y,
                           (v.z - mu.z) * is.z * ga.z + be.z, (v.w - mu.w) * is.w * ga.w + be.w);
    if (res) {
      const float4 e = *reinterpret_cast<const float4*>(res + i * 4);
      o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4*>(y + i * 4) = o;
  }
}

// backward pass 1: per block column sums of g = dy * [y > 0] and of g * xhat
__global__ __launch_bounds__(BN_T) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, long R, int C,
                                                             int rows_per_block, int relu, float* __restrict__ part /* [nblk][2][C] */) {
  const BnGeom g = bn_geom(C);
  const int tid = threadIdx.x;
  const int c4 = tid % g.CW, tr = tid / g.CW;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  __shared__ float4 red[2][BN_T];
  const int col = (blockIdx.y * g.CW + c4) * 4;
  const float4 mu = *reinterpret_cast<const float4*>(mean + col), is = *reinterpret_cast<const float4*>(invstd + col);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  for (long r = r0 + tr; r < r1; r += g.tpr) {
    float4 d = *reinterpret_cast<const float4*>(dy + (size_t)r * C + col);
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + col);
    if (relu) {
      const float4 o = *reinterpret_cast<const float4*>(y + (size_t)r * C + col);
      d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
    }
    s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
    s2.x += d.x * (v.x - mu.x) * is.x; s2.y += d.y * (v.y - mu.y) * is.y;
    s2.z += d.z * (v.z - mu.z) * is.z; s2.w += d.w * (v.w - mu.w) * is.w;
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  if (tr == 0) {
    for (int j = 1; j < g.tpr; ++j) {
      const float4 a = red[0][c4 + j * g.CW], b = red[1][c4 + j * g.CW];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
    }
    float* p = part + (size_t)blockIdx.x * 2 * C;
    *reinterpret_cast<float4*>(p + col) = s1;
    *reinterpret_cast<float4*>(p + C + col) = s2;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C,
                                                              float* __restrict__ sums /* [2][C] */,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double sa[16][17], sb[16][17];
  const int cl = threadIdx.x & 15, l = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int k = l; k < nblk; k += 16) { a += part[(size_t)k * 2 * C + c]; b += part[(size_t)k * 2 * C + C + c]; }
  sa[cl][l] = a; sb[cl][l] = b;
  __syncthreads();
  if (l == 0 && c < C) {
    for (int j = 1; j < 16; ++j) { a += sa[cl][j]; b += sb[cl][j]; }
    sums[c] = (float)a;
    sums[C + c] = (float)b;
    if (dbeta) dbeta[c] = (float)a;
    if (dgamma) dgamma[c] = (float)b;
  }
}

// backward pass 2: dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)); dres = g
__global__ __launch_bounds__(BN_T) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ sums, float* __restrict__ dx,
                                                            float* __restrict__ dres, long R, int C, int relu) {
  const long n4 = R * (long)(C / 4);
  const float invR = 1.f / (float)R;
  for (long i = (long)blockIdx.x * BN_T + threadIdx.x; i < n4; i += (long)gridDim.x * BN_T) {
    const int col = (int)(i % (C / 4)) * 4;
    float4 d = *reinterpret_cast<const float4*>(dy + i * 4);
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    if (relu) {
      const float4 o = *reinterpret_cast<const float4*>(y + i * 4);
      d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
    }
    if (dres) *reinterpret_cast<float4*>(dres + i * 4) = d;
    const float4 mu = *reinterpret_cast<const float4*>(mean + col), is = *reinterpret_cast<const float4*>(invstd + col);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + col);
    const float4 sg = *reinterpret_cast<const float4*>(sums + col), sx = *reinterpret_cast<const float4*>(sums + C + col);
    float4 o;
    o.x = ga.x * is.x * (d.x - sg.x * invR - (v.x - mu.x) * is.x * sx.x * invR);
    o.y = ga.y * is.y * (d.y - sg.y * invR - (v.y - mu.y) * is.y * sx.y * invR);
    o.z = ga.z * is.z * (d.z - sg.z * invR - (v.z - mu.z) * is.z * sx.z * invR);
    o.w = ga.w * is.w * (d.w - sg.w * invR - (v.w - mu.w) * is.w * sx.w * invR);
    *reinterpret_cast<float4*>(dx + i * 4) = o;
  }
}

inline int bn_blocks(long R, int C, int& rows_per_block) {
  const BnGeom g = bn_geom(C);
  long nblk = 2048 / g.ncol;                                   // row slabs so that slabs x column chunks ~ 2048 blocks
  const long max_slabs = (R + 4 * g.tpr - 1) / (4 * g.tpr);   // >= 4 rows per thread-row
  if (nblk > max_slabs) nblk = max_slabs;
  if (nblk > 1024) nblk = 1024;
  if (nblk < 1) nblk = 1;
  rows_per_block = (int)((R + nblk - 1) / nblk);
  return (int)((R + rows_per_block - 1) / rows_per_block);
}
inline bool bn_ok(int C) {
  const int c4 = C / 4;
  return C > 0 && (C & 3) == 0 && C <= 2048 && (c4 <= 64 ? BN_T % c4 == 0 : c4 % 64 == 0);
}

}  // namespace
}  // namespace hoisdf

using namespace hoisdf;

extern "C" long hoisdf_batchnorm_workspace(long R, int C) {
  if (R <= 0 || C <= 0) return 0;
  int rpb;
  const int nblk = bn_blocks(R, C, rpb);
  return ((long)nblk * 3 * C + 2 * C) * (long)sizeof(float);
}

extern "C" int hoisdf_batchnorm_relu_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float momentum, float eps, int relu,
                                         float* y, float* mean, float* invstd, long R, int C, void* workspace,
                                         long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(x && gamma && beta && y && mean && invstd && workspace, HOISDF_ERR_INVALID, "batchnorm_relu_fwd: null pointer");
  HOISDF_REQUIRE(R > 1 && bn_ok(C), HOISDF_ERR_INVALID, "batchnorm_relu_fwd: R=%ld C=%d (C must be a multiple of 4 dividing / divisible by 1024, <= 2048)", R, C);
  HOISDF_REQUIRE(workspace_bytes >= hoisdf_batchnorm_workspace(R, C), HOISDF_ERR_WORKSPACE, "batchnorm_relu_fwd: workspace too small");
  hipStream_t st = as_stream(stream);
  int rpb;
  const int nblk = bn_blocks(R, C, rpb);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk, bn_geom(C).ncol), dim3(BN_T), 0, st, x, R, C, rpb, part);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, part, nblk, R, C, rpb, eps, momentum, mean, invstd,
                     running_mean, running_var);
  long blocks = (R * (C / 4) + BN_T - 1) / BN_T;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(BN_T), 0, st, x, residual, mean, invstd, gamma, beta, y, R, C,
                     relu);
  return check_launch("batchnorm_relu_fwd");
}

extern "C" int hoisdf_batchnorm_relu_bwd(const float* dy, const float* x, const float* y, const float* gamma,
                                         const float* mean, const float* invstd, int relu, float* dx, float* dresidual,
                                         float* dgamma, float* dbeta, long R, int C, void* workspace, long workspace_bytes,
                                         void* stream) {
  HOISDF_REQUIRE(dy && x && gamma && mean && invstd && dx && workspace && (!relu || y), HOISDF_ERR_INVALID,
                 "batchnorm_relu_bwd: null pointer");
  HOISDF_REQUIRE(R > 1 && bn_ok(C), HOISDF_ERR_INVALID, "batchnorm_relu_bwd: R=%ld C=%d", R, C);
  HOISDF_REQUIRE(workspace_bytes >= hoisdf_batchnorm_workspace(R, C), HOISDF_ERR_WORKSPACE, "batchnorm_relu_bwd: workspace too small");
  hipStream_t st = as_stream(stream);
  int rpb;
  const int nblk = bn_blocks(R, C, rpb);
  float* part = static_cast<float*>(workspace);
  float* sums = part + (size_t)nblk * 3 * C;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk, bn_geom(C).ncol), dim3(BN_T), 0, st, dy, x, y, mean, invstd, R, C, rpb, relu, part);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, part, nblk, C, sums, dgamma, dbeta);
  long blocks = (R * (C / 4) + BN_T - 1) / BN_T;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(BN_T), 0, st, dy, x, y, mean, invstd, gamma, sums, dx,
                     dresidual, R, C, relu);
  return check_launch("batchnorm_relu_bwd");
}
