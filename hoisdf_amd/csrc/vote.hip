// K12: vote aggregation (common/nets/loss.py:31-56): for every encoder depth l and sample b the
// 20 joints are the softmax-over-POINTS weighted sum of the per-point votes
//   joints[l][b][j] = sum_p softmax_p(cls[l][b][:, j])[p] * (pts[b][p] + off[l][b][p][j]).
// HBM-bound (reads 80 floats per (l,b,p) once in the forward): one workgroup per (l, b); thread
// t owns joint t % J and walks the points with a stride that is a multiple of J, so every
// wave-level access to cls is one contiguous run.  Max / sum statistics are saved for backward.
#include <stdlib.h>

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

// ---- K12 with the JointvoteLoss reductions fused (common/nets/loss.py:31-56) -----------------
// Besides the joints, one pass over (l, b) accumulates
//   l3d_sum[l][b]  = sum_{p,j,d} smooth_l1(1000*vote - gt) * near[b][p][j]
//   bce_sum[l][b]  = sum_{p,j}   bce_with_logits(cls, near)
//   near_sum[b]    = sum_{p,j}   near            (written by the l == 0 workgroups)
// with near = ||pts - gt/1000|| < radius; the scalar losses are means of these sums.
__device__ __forceinline__ float smooth_l1(float x) { const float a = fabsf(x); return a < 1.f ? 0.5f * x * x : a - 0.5f; }
__device__ __forceinline__ float smooth_l1_grad(float x) { return x <= -1.f ? -1.f : (x >= 1.f ? 1.f : x); }

__global__ __launch_bounds__(256) void vote_loss_fwd_kernel(const float* __restrict__ off, const float* __restrict__ cls,
                                                            const float* __restrict__ pts, const float* __restrict__ gt,
                                                            float radius, float* __restrict__ joints,
                                                            float* __restrict__ stats, float* __restrict__ l3d_sum,
                                                            float* __restrict__ bce_sum, float* __restrict__ near_sum,
                                                            int B, int P, int J) {
  __shared__ float red[7][256];
  __shared__ float smax[64];
  const int lb = blockIdx.x, b = lb % B, l = lb / B;
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
  float s = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, l3 = 0.f, bce = 0.f, nr = 0.f;
  if (active) {
    const float M = smax[j];
    const float g0 = gt[((size_t)b * J + j) * 3 + 0], g1 = gt[((size_t)b * J + j) * 3 + 1],
                g2 = gt[((size_t)b * J + j) * 3 + 2];
    for (int p = pl; p < P; p += per) {
      const float cv = c[(size_t)p * J + j];
      const float e = expf(cv - M);
      const float* oo = o + ((size_t)p * J + j) * 3;
      const float px = pp[p * 3 + 0], py = pp[p * 3 + 1], pz = pp[p * 3 + 2];
      const float v0 = px + oo[0], v1 = py + oo[1], v2 = pz + oo[2];
      s += e; a0 += e * v0; a1 += e * v1; a2 += e * v2;
      const float dx = px - g0 / 1000.f, dy = py - g1 / 1000.f, dz = pz - g2 / 1000.f;
      const float near = sqrtf(dx * dx + dy * dy + dz * dz) < radius ? 1.f : 0.f;
      l3 += near * (smooth_l1(v0 * 1000.f - g0) + smooth_l1(v1 * 1000.f - g1) + smooth_l1(v2 * 1000.f - g2));
      bce += fmaxf(cv, 0.f) - cv * near + log1pf(expf(-fabsf(cv)));
      nr += near;
    }
  }
  red[0][tid] = s; red[1][tid] = a0; red[2][tid] = a1; red[3][tid] = a2;
  red[4][tid] = l3; red[5][tid] = bce; red[6][tid] = nr;
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
  if (tid == 64) {                       // a different wave sums the loss partials
    float L3 = 0.f, BC = 0.f, NR = 0.f;
    for (int k = 0; k < per * J; ++k) { L3 += red[4][k]; BC += red[5][k]; NR += red[6][k]; }
    l3d_sum[lb] = L3;
    bce_sum[lb] = BC;
    if (l == 0) near_sum[b] = NR;
  }
}

// Round 6: the same reductions with the POINTS of a (depth, sample) cut into segments - the kernel above is one block per (l, b):
// 24 blocks for configs[4] (B = 4, 6144 points: 483 us), 96 for configs[3].  A block reduces its segment against the segment's own
// maxima; the finishing kernel merges the segments in order (S = sum_seg s_seg exp(m_seg - M), ...): order-fixed, reproducible.
// part[lb][seg][5][J] = (m, s, a0, a1, a2) per joint, lpart[lb][seg][3] = (l3d, bce, near).
__global__ __launch_bounds__(256) void vote_loss_part_kernel(const float* __restrict__ off, const float* __restrict__ cls,
                                                             const float* __restrict__ pts, const float* __restrict__ gt, float radius,
                                                             float* __restrict__ part, float* __restrict__ lpart, int B, int P, int J, int chunk) {
  __shared__ float red[7][256];
  __shared__ float smax[64];
  const int lb = blockIdx.y, b = lb % B, seg = blockIdx.x, nseg = gridDim.x;
  const int p0 = seg * chunk, p1 = min(P, p0 + chunk);
  const float* c = cls + (size_t)lb * P * J;
  const float* o = off + (size_t)lb * P * J * 3;
  const float* pp = pts + (size_t)b * P * 3;
  const int tid = threadIdx.x;
  const int per = 256 / J;
  const int j = tid % J, pl = tid / J;
  const bool active = pl < per;
  float m = -INFINITY;
  if (active)
    for (int p = p0 + pl; p < p1; p += per) m = fmaxf(m, c[(size_t)p * J + j]);
  red[0][tid] = m;
  __syncthreads();
  if (tid < J) {
    float mm = -INFINITY;
    for (int k = 0; k < per; ++k) mm = fmaxf(mm, red[0][k * J + tid]);
    smax[tid] = mm;
  }
  __syncthreads();
  float s = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, l3 = 0.f, bce = 0.f, nr = 0.f;
  if (active) {
    const float M = smax[j];
    const float g0 = gt[((size_t)b * J + j) * 3 + 0], g1 = gt[((size_t)b * J + j) * 3 + 1],
                g2 = gt[((size_t)b * J + j) * 3 + 2];
    for (int p = p0 + pl; p < p1; p += per) {
      const float cv = c[(size_t)p * J + j];
      const float e = expf(cv - M);
      const float* oo = o + ((size_t)p * J + j) * 3;
      const float px = pp[p * 3 + 0], py = pp[p * 3 + 1], pz = pp[p * 3 + 2];
      const float v0 = px + oo[0], v1 = py + oo[1], v2 = pz + oo[2];
      s += e; a0 += e * v0; a1 += e * v1; a2 += e * v2;
      const float dx = px - g0 / 1000.f, dy = py - g1 / 1000.f, dz = pz - g2 / 1000.f;
      const float near = sqrtf(dx * dx + dy * dy + dz * dz) < radius ? 1.f : 0.f;
      l3 += near * (smooth_l1(v0 * 1000.f - g0) + smooth_l1(v1 * 1000.f - g1) + smooth_l1(v2 * 1000.f - g2));
      bce += fmaxf(cv, 0.f) - cv * near + log1pf(expf(-fabsf(cv)));
      nr += near;
    }
  }
  red[0][tid] = s; red[1][tid] = a0; red[2][tid] = a1; red[3][tid] = a2;
  red[4][tid] = l3; red[5][tid] = bce; red[6][tid] = nr;
  __syncthreads();
  float* out = part + ((size_t)lb * nseg + seg) * 5 * J;
  if (tid < J) {
    float S = 0.f, A0 = 0.f, A1 = 0.f, A2 = 0.f;
    for (int k = 0; k < per; ++k) {
      S += red[0][k * J + tid]; A0 += red[1][k * J + tid]; A1 += red[2][k * J + tid]; A2 += red[3][k * J + tid];
    }
    out[tid] = smax[tid]; out[J + tid] = S; out[2 * J + tid] = A0; out[3 * J + tid] = A1; out[4 * J + tid] = A2;
  }
  if (tid == 64) {
    float L3 = 0.f, BC = 0.f, NR = 0.f;
    for (int k = 0; k < per * J; ++k) { L3 += red[4][k]; BC += red[5][k]; NR += red[6][k]; }
    float* lo = lpart + ((size_t)lb * nseg + seg) * 3;
    lo[0] = L3; lo[1] = BC; lo[2] = NR;
  }
}
__global__ __launch_bounds__(64) void vote_loss_merge_kernel(const float* __restrict__ part, const float* __restrict__ lpart, int nseg,
                                                             float* __restrict__ joints, float* __restrict__ stats, float* __restrict__ l3d_sum,
                                                             float* __restrict__ bce_sum, float* __restrict__ near_sum, int B, int J) {
  const int lb = blockIdx.x, b = lb % B, l = lb / B, tid = threadIdx.x;
  const float* in = part + (size_t)lb * nseg * 5 * J;
  if (tid < J) {
    float M = -INFINITY;
    for (int g = 0; g < nseg; ++g) M = fmaxf(M, in[(size_t)g * 5 * J + tid]);
    float S = 0.f, A0 = 0.f, A1 = 0.f, A2 = 0.f;
    for (int g = 0; g < nseg; ++g) {
      const float* q = in + (size_t)g * 5 * J;
      const float w = expf(q[tid] - M);              // (an empty segment: m = -inf, s = 0 -> w = 0)
      S += q[J + tid] * w; A0 += q[2 * J + tid] * w; A1 += q[3 * J + tid] * w; A2 += q[4 * J + tid] * w;
    }
    float* jo = joints + ((size_t)lb * J + tid) * 3;
    jo[0] = A0 / S; jo[1] = A1 / S; jo[2] = A2 / S;
    stats[((size_t)lb * J + tid) * 2 + 0] = M;
    stats[((size_t)lb * J + tid) * 2 + 1] = S;
  }
  if (tid == 63) {
    float L3 = 0.f, BC = 0.f, NR = 0.f;
    for (int g = 0; g < nseg; ++g) { const float* q = lpart + ((size_t)lb * nseg + g) * 3; L3 += q[0]; BC += q[1]; NR += q[2]; }
    l3d_sum[lb] = L3;
    bce_sum[lb] = BC;
    if (l == 0) near_sum[b] = NR;
  }
}

// doff = w dJ + dl3d[l][b] * near * 1000 * sl1'(1000 vote - gt)
// dcls = w * sum_d dJ (vote - joints) + dbce[l][b] * (sigmoid(cls) - near)
__global__ __launch_bounds__(256) void vote_loss_bwd_kernel(const float* __restrict__ off, const float* __restrict__ cls,
                                                            const float* __restrict__ pts, const float* __restrict__ gt,
                                                            float radius, const float* __restrict__ joints,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ djoints,
                                                            const float* __restrict__ dl3d, const float* __restrict__ dbce,
                                                            float* __restrict__ doff, float* __restrict__ dcls, int B,
                                                            int P, int J) {
  const int lb = blockIdx.y, b = lb % B;
  const int tid = threadIdx.x;
  const int per = 256 / J;
  const int j = tid % J, pl = tid / J;
  if (pl >= per) return;
  const int p = blockIdx.x * per + pl;
  if (p >= P) return;
  const size_t e = ((size_t)lb * P + p) * J + j;
  const float M = stats[((size_t)lb * J + j) * 2 + 0], S = stats[((size_t)lb * J + j) * 2 + 1];
  const float cv = cls[e];
  const float w = expf(cv - M) / S;
  const float* jo = joints + ((size_t)lb * J + j) * 3;
  const float* dj = djoints ? djoints + ((size_t)lb * J + j) * 3 : nullptr;
  const float* pp = pts + ((size_t)b * P + p) * 3;
  const float* g = gt + ((size_t)b * J + j) * 3;
  const float c3 = dl3d ? dl3d[lb] : 0.f, cb = dbce ? dbce[lb] : 0.f;
  float d2 = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) { const float t = pp[d] - g[d] / 1000.f; d2 += t * t; }
  const float near = sqrtf(d2) < radius ? 1.f : 0.f;
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float vote = pp[d] + off[e * 3 + d];
    const float djd = dj ? dj[d] : 0.f;
    doff[e * 3 + d] = w * djd + c3 * near * 1000.f * smooth_l1_grad(vote * 1000.f - g[d]);
    acc += djd * (vote - jo[d]);
  }
  dcls[e] = w * acc + cb * (1.f / (1.f + expf(-cv)) - near);
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

extern "C" int hoisdf_vote_loss_fwd(const float* off, const float* cls, const float* pts, const float* joint_gt_mm,
                                    float radius, float* joints, float* stats, float* l3d_sum, float* bce_sum,
                                    float* near_sum, int L, int B, int P, int J, void* stream) {
  HOISDF_REQUIRE(off && cls && pts && joint_gt_mm && joints && stats && l3d_sum && bce_sum && near_sum,
                 HOISDF_ERR_INVALID, "vote_loss_fwd: null pointer");
  HOISDF_REQUIRE(L > 0 && B > 0 && P > 0 && J > 0 && J <= 64, HOISDF_ERR_INVALID, "vote_loss_fwd: bad sizes");
  // few (depth, sample) pairs and many points: segments of the points in separate blocks + an ordered merge (HOISDF_VOTE_SPLIT=0: never)
  static int split_on = -1;
  if (split_on < 0) { const char* e = getenv("HOISDF_VOTE_SPLIT"); split_on = (e && atoi(e) == 0) ? 0 : 1; }
  int nseg = split_on ? min(cdiv(P, 512), max(1, 512 / (L * B))) : 1;
  float* scratch = nullptr;
  if (nseg > 1) scratch = reinterpret_cast<float*>(mag_scratch(as_stream(stream), (long)L * B * nseg * (5 * J + 3)));
  if (nseg > 1 && scratch) {
    const int chunk = cdiv(P, nseg);
    float* lpart = scratch + (size_t)L * B * nseg * 5 * J;
    hipLaunchKernelGGL(vote_loss_part_kernel, dim3(nseg, L * B), dim3(256), 0, as_stream(stream), off, cls, pts, joint_gt_mm, radius, scratch, lpart,
                       B, P, J, chunk);
    hipLaunchKernelGGL(vote_loss_merge_kernel, dim3(L * B), dim3(64), 0, as_stream(stream), scratch, lpart, nseg, joints, stats, l3d_sum, bce_sum,
                       near_sum, B, J);
    return check_launch("vote_loss_fwd (segments)");
  }
  hipLaunchKernelGGL(vote_loss_fwd_kernel, dim3(L * B), dim3(256), 0, as_stream(stream), off, cls, pts, joint_gt_mm,
                     radius, joints, stats, l3d_sum, bce_sum, near_sum, B, P, J);
  return check_launch("vote_loss_fwd");
}

extern "C" int hoisdf_vote_loss_bwd(const float* off, const float* cls, const float* pts, const float* joint_gt_mm,
                                    float radius, const float* joints, const float* stats, const float* djoints,
                                    const float* dl3d_sum, const float* dbce_sum, float* doff, float* dcls, int L,
                                    int B, int P, int J, void* stream) {
  HOISDF_REQUIRE(off && cls && pts && joint_gt_mm && joints && stats && doff && dcls, HOISDF_ERR_INVALID,
                 "vote_loss_bwd: null pointer");
  HOISDF_REQUIRE(L > 0 && B > 0 && P > 0 && J > 0 && J <= 64, HOISDF_ERR_INVALID, "vote_loss_bwd: bad sizes");
  const int per = 256 / J;
  hipLaunchKernelGGL(vote_loss_bwd_kernel, dim3(cdiv(P, per), L * B), dim3(256), 0, as_stream(stream), off, cls, pts,
                     joint_gt_mm, radius, joints, stats, djoints, dl3d_sum, dbce_sum, doff, dcls, B, P, J);
  return check_launch("vote_loss_bwd");
}
