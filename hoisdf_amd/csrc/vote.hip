// K12: vote aggregation (common/nets/loss.py:31-56): for every encoder depth l and sample b the
// 20 joints are the softmax-over-POINTS weighted sum of the per-point votes
//   joints[l][b][j] = sum_p softmax_p(cls[l][b][:, j])[p] * (pts[b][p] + off[l][b][p][j]).
// HBM-bound (reads 80 floats per (l,b,p) once in the forward): one workgroup per (l, b); thread
// t owns joint t % J and walks the points with a stride that is a multiple of J, so every
// wave-level access to cls is one contiguous run.  Max / sum statistics are saved for backward.
#include "common.h"

namespace hoisdf {

__global__ __launch_bounds__(256) void vote_fwd_kernel(const float* __restrict__ off, const float* __restrict__ cls,
                                                       const float* __restrict__ pts, float* __restrict__ joints,
                                                       float* __restrict__ stats, int B, int P, int J) {
  __shared__ float red[4][256];
  __shared__ float smax[64];
  const int lb = blockIdx.x, b = lb % B;
  const float* c = cls + (size_t)lb * P * J;
  const float* o = off + (size_t)lb * P * J * 3;
  const float* pp = pts + (size_t)b * P * 3;
  const int tid = threadIdx.x;
  const int per = 256 / J;
  const int j = tid % J, pl = tid / J;
  const bool active = pl < per;
  float m = -INFINITY;
  if (active)
    for (int p = pl; p < P; p += per) m = fmaxf(m, c[(size_t)p * J + j]);
  red[0][tid] = m;
  __syncthreads();
  if (tid < J) {
    float mm = -INFINITY;
    for (int k = 0; k < per; ++k) mm = fmaxf(mm, red[0][k * J + tid]);
    smax[tid] = mm;
  }
  __syncthreads();
  float s = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
  if (active) {
    const float M = smax[j];
    for (int p = pl; p < P; p += per) {
      const float e = expf(c[(size_t)p * J + j] - M);
      const float* oo = o + ((size_t)p * J + j) * 3;
      s += e;
      a0 += e * (pp[p * 3 + 0] + oo[0]);
      a1 += e * (pp[p * 3 + 1] + oo[1]);
      a2 += e * (pp[p * 3 + 2] + oo[2]);
    }
  }
  red[0][tid] = s; red[1][tid] = a0; red[2][tid] = a1; red[3][tid] = a2;
  __syncthreads();
  if (tid < J) {
    float S = 0.f, A0 = 0.f, A1 = 0.f, A2 = 0.f;
    for (int k = 0; k < per; ++k) {
      S += red[0][k * J + tid]; A0 += red[1][k * J + tid]; A1 += red[2][k * J + tid]; A2 += red[3][k * J + tid];
    }
    float* jo = joints + ((size_t)lb * J + tid) * 3;
    jo[0] = A0 / S; jo[1] = A1 / S; jo[2] = A2 / S;
    stats[((size_t)lb * J + tid) * 2 + 0] = smax[tid];
    stats[((size_t)lb * J + tid) * 2 + 1] = S;
  }
}

// doff[p][j][d] = w dJ[j][d];  dcls[p][j] = w * sum_d dJ[j][d] (vote[p][j][d] - joints[j][d])
__global__ __launch_bounds__(256) void vote_bwd_kernel(const float* __restrict__ off, const float* __restrict__ cls,
                                                       const float* __restrict__ pts, const float* __restrict__ joints,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ djoints, float* __restrict__ doff,
                                                       float* __restrict__ dcls, int B, int P, int J) {
  const int lb = blockIdx.y, b = lb % B;
  const int tid = threadIdx.x;
  const int per = 256 / J;
  const int j = tid % J, pl = tid / J;
  if (pl >= per) return;
  const int p = blockIdx.x * per + pl;
  if (p >= P) return;
  const size_t e = ((size_t)lb * P + p) * J + j;
  const float M = stats[((size_t)lb * J + j) * 2 + 0], S = stats[((size_t)lb * J + j) * 2 + 1];
  const float w = expf(cls[e] - M) / S;
  const float* jo = joints + ((size_t)lb * J + j) * 3;
  const float* dj = djoints + ((size_t)lb * J + j) * 3;
  const float* pp = pts + ((size_t)b * P + p) * 3;
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float vote = pp[d] + off[e * 3 + d];
    doff[e * 3 + d] = w * dj[d];
    acc += dj[d] * (vote - jo[d]);
  }
  dcls[e] = w * acc;
}

}  // namespace hoisdf

using namespace hoisdf;

extern "C" int hoisdf_vote_fwd(const float* off, const float* cls, const float* pts, float* joints, float* stats,
                               int L, int B, int P, int J, void* stream) {
  HOISDF_REQUIRE(off && cls && pts && joints && stats, HOISDF_ERR_INVALID, "vote_fwd: null pointer");
  HOISDF_REQUIRE(L > 0 && B > 0 && P > 0 && J > 0 && J <= 64, HOISDF_ERR_INVALID, "vote_fwd: bad sizes");
  hipLaunchKernelGGL(vote_fwd_kernel, dim3(L * B), dim3(256), 0, as_stream(stream), off, cls, pts, joints, stats, B,
                     P, J);
  return check_launch("vote_fwd");
}

extern "C" int hoisdf_vote_bwd(const float* off, const float* cls, const float* pts, const float* joints,
                               const float* stats, const float* djoints, float* doff, float* dcls, int L, int B,
                               int P, int J, void* stream) {
  HOISDF_REQUIRE(off && cls && pts && joints && stats && djoints && doff && dcls, HOISDF_ERR_INVALID,
                 "vote_bwd: null pointer");
  HOISDF_REQUIRE(L > 0 && B > 0 && P > 0 && J > 0 && J <= 64, HOISDF_ERR_INVALID, "vote_bwd: bad sizes");
  const int per = 256 / J;
  hipLaunchKernelGGL(vote_bwd_kernel, dim3(cdiv(P, per), L * B), dim3(256), 0, as_stream(stream), off, cls, pts,
                     joints, stats, djoints, doff, dcls, B, P, J);
  return check_launch("vote_bwd");
}
