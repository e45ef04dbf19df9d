// (f3) MANO head: 6D pose -> rotation (Gram-Schmidt) -> quaternion -> axis-angle -> MANO layer (shape / pose blend shapes, joint
// regression, forward kinematics, linear blend skinning) -> vertices / 21 joints, with the four ManoLoss squared-error sums
// fused, and the whole backward in one kernel.  Reference: common/nets/mano_head.py:12-278 (conversions, head),
// manopth/manopth/manolayer.py:111-276 (layer, the configuration main/model.py:735-742 builds), common/nets/loss.py:81-171.
//
// One workgroup per hand (4 * B hands a step: 3 decoder layers of predictions + the ground truth): everything of a hand - 2334
// vertex coordinates, 16 rotations, the kinematic chain - lives in LDS; the blend-shape tables are read through a transposed image
// ([145][2334]: 10 shape + 135 pose directions, built once by hoisdf_mano_prepare) so every table read is coalesced, forward and
// backward.  The torch chain this replaces was ~1 400 ATen launches a step (profiles/r03_bench_kernel_stats.csv).
//
// Backward: the predicted rotation goes R6 = GramSchmidt(x) -> log (quaternion, axis-angle) -> exp (the layer's Rodrigues).
// With a zero hand mean (flat_hand_mean=True, the reference's configuration; the host refuses anything else) exp(log(.)) is the
// identity ON SO(3), and Gram-Schmidt only ever moves R6 along SO(3): the Jacobian of the round trip restricted to the tangent
// space is the identity, so the gradient arriving at the layer's rotation is applied to R6 directly - the exact derivative of the
// same function, without differentiating atan2 / the four-branch quaternion (whose fp32 derivative is noise near the branch
// edges).  The forward still walks the reference's full conversion chain, so forward values carry the reference's rounding.
#include "common.h"

namespace hoisdf {
namespace {

constexpr int MV = 778, MVC = 2334, MJ = 16, MPM = 135, MB = 10, MJT = 21;
constexpr int MNT = 1024, MNW = MNT / 64;   // 16 waves a hand: the phases are chains of L2-latency loads, more waves = more of them in flight
constexpr int MDIRS = MB + MPM;                 // rows of the transposed direction image
constexpr long MWT_OFF = (long)MDIRS * MVC;     // ... followed by the skinning weights transposed [16][778]

__constant__ int c_tip[5] = {745, 317, 444, 556, 673};
__constant__ int c_order[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

struct ManoArgs {
  const float* pose; int ldpose;      // mode 0: [H][16][6] (ldpose = 96); mode 1: axis-angle coefficients [H][>= 48]
  const float* betas; int ldbetas;    // [H][>= 10]
  int mode, H;
  const float* dirs;                  // [145][2334]
  const float* v_template;            // [2334]
  const float* j_reg;                 // [16][778]
  const float* weights;               // [778][16]
  const float* hands_mean;            // [45]
  // ground truth of the fused losses (mode 0; null = no losses): hand h compares with gt hand h % gt_hands
  const float* gt_verts; const float* gt_joints; const float* gt_rot; const float* gt_shape; int ldgt_shape, gt_hands;
  float* verts; float* joints; float* rot; float* sums;       // [H][778][3], [H][21][3], [H][16][9], [H][4]
  // backward only
  const float* g_sums; const float* g_verts; const float* g_joints; const float* g_rot;   // each may be null
  float* d_pose; float* d_betas;                                                         // [H][16][6], [H][10]
};

struct ManoLds {
  float vs[MVC];        // shaped vertices; backward: gradient of the posed / shaped vertices
  float vp[MVC];        // posed vertices
  float raw[MVC];       // skinned vertices before centring; backward: their gradient
  float R[MJ * 9];      // rotations the layer uses
  float R6[MJ * 9];     // mode 0: Gram-Schmidt rotations (columns b1 b2 b3); mode 1: Rodrigues of the mean-free coefficients
  float J[MJ * 3];
  float G[MJ * 12];     // world transforms, rows of [R | t]
  float A[MJ * 12];     // skinning transforms
  float pm[MPM + 1];
  float beta[MB + 2];
  float cat[MJT * 3];   // 16 joint positions + 5 finger tips, before the reordering
  float red[MNW * 4];
};

__device__ __forceinline__ void rodrigues(const float t[3], float* R) {     // mano_head.py:12-52 / rodrigues_layer.py:43-54
  const float e0 = t[0] + 1e-8f, e1 = t[1] + 1e-8f, e2 = t[2] + 1e-8f;
  const float ang = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  const float half = 0.5f * ang, sn = sinf(half);
  float w = cosf(half), x = sn * (t[0] / ang), y = sn * (t[1] / ang), z = sn * (t[2] / ang);
  const float n = sqrtf(w * w + x * x + y * y + z * z);
  w /= n; x /= n; y /= n; z /= n;
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * x * y - 2 * w * z;         R[2] = 2 * w * y + 2 * x * z;
  R[3] = 2 * w * z + 2 * x * y;         R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * y * z - 2 * w * x;
  R[6] = 2 * x * z - 2 * w * y;         R[7] = 2 * w * x + 2 * y * z;         R[8] = w * w - x * x - y * y + z * z;
}

// Gram-Schmidt of the 6D representation (mano_head.py:185-194); F.normalize's 1e-12 clamp on the norms
__device__ __forceinline__ void gram_schmidt(const float* x, float b1[3], float b2[3], float b3[3], float& n1, float& n2, float& d) {
  n1 = fmaxf(sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), 1e-12f);
  b1[0] = x[0] / n1; b1[1] = x[1] / n1; b1[2] = x[2] / n1;
  d = b1[0] * x[3] + b1[1] * x[4] + b1[2] * x[5];
  const float u0 = x[3] - d * b1[0], u1 = x[4] - d * b1[1], u2 = x[5] - d * b1[2];
  n2 = fmaxf(sqrtf(u0 * u0 + u1 * u1 + u2 * u2), 1e-12f);
  b2[0] = u0 / n2; b2[1] = u1 / n2; b2[2] = u2 / n2;
  b3[0] = b1[1] * b2[2] - b1[2] * b2[1]; b3[1] = b1[2] * b2[0] - b1[0] * b2[2]; b3[2] = b1[0] * b2[1] - b1[1] * b2[0];
}

// rotation (rows of T = R^T are b1 b2 b3) -> quaternion -> axis-angle (mano_head.py:54-182), NaN components -> 0
__device__ __forceinline__ void rotation_to_axis_angle(const float b1[3], const float b2[3], const float b3[3], float aa[3]) {
  const float t00 = b1[0], t11 = b2[1], t22 = b3[2];
  const float t01 = b1[1], t02 = b1[2], t10 = b2[0], t12 = b2[2], t20 = b3[0], t21 = b3[1];
  float q0, q1, q2, q3, tr;
  if (t22 < 1e-6f) {
    if (t00 > t11) { tr = 1 + t00 - t11 - t22; q0 = t12 - t21; q1 = tr; q2 = t01 + t10; q3 = t20 + t02; }
    else           { tr = 1 - t00 + t11 - t22; q0 = t20 - t02; q1 = t01 + t10; q2 = tr; q3 = t12 + t21; }
  } else {
    if (t00 < -t11) { tr = 1 - t00 - t11 + t22; q0 = t01 - t10; q1 = t20 + t02; q2 = t12 + t21; q3 = tr; }
    else            { tr = 1 + t00 + t11 + t22; q0 = tr; q1 = t12 - t21; q2 = t20 - t02; q3 = t01 - t10; }
  }
  const float sc = 0.5f / sqrtf(tr);            // q / sqrt(t) * 0.5
  q0 *= sc; q1 *= sc; q2 *= sc; q3 *= sc;
  const float s2 = q1 * q1 + q2 * q2 + q3 * q3, s = sqrtf(s2);
  const float two_theta = 2.f * (q0 < 0.f ? atan2f(-s, -q0) : atan2f(s, q0));
  const float k = s2 > 0.f ? two_theta / s : 2.f;
  aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
#pragma unroll
  for (int i = 0; i < 3; ++i) if (aa[i] != aa[i]) aa[i] = 0.f;
}

// the forward of one hand into LDS (all threads of the workgroup); leaves s.raw (uncentred vertices), s.cat, and everything before them
__device__ void mano_forward(const ManoArgs& a, int h, ManoLds& s) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < MJ) {
    const int j = tid;
    float full[3];
    if (a.mode == 0) {
      const float* x = a.pose + (size_t)h * a.ldpose + j * 6;
      const float xv[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
      float b1[3], b2[3], b3[3], n1, n2, d;
      gram_schmidt(xv, b1, b2, b3, n1, n2, d);
#pragma unroll
      for (int r = 0; r < 3; ++r) { s.R6[j * 9 + r * 3 + 0] = b1[r]; s.R6[j * 9 + r * 3 + 1] = b2[r]; s.R6[j * 9 + r * 3 + 2] = b3[r]; }
      rotation_to_axis_angle(b1, b2, b3, full);
#pragma unroll
      for (int i = 0; i < 3; ++i) if (j > 0) full[i] += a.hands_mean[(j - 1) * 3 + i];
    } else {
      // the head hands the layer "coefficients minus the mean" and the layer adds the mean back (mano_head.py:258-262)
      const float* p = a.pose + (size_t)h * a.ldpose + j * 3;
      float c[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float m = j > 0 ? a.hands_mean[(j - 1) * 3 + i] : 0.f;
        c[i] = j > 0 ? p[i] - m : p[i];
        full[i] = j > 0 ? m + c[i] : c[i];
      }
      rodrigues(c, &s.R6[j * 9]);
    }
    rodrigues(full, &s.R[j * 9]);
  }
  if (tid >= 64 && tid < 64 + MB) s.beta[tid - 64] = a.betas[(size_t)h * a.ldbetas + (tid - 64)];
  __syncthreads();
  if (tid < MPM) s.pm[tid] = s.R[9 + tid] - ((tid % 9) % 4 == 0 ? 1.f : 0.f);
  for (int vc = tid; vc < MVC; vc += MNT) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < MB; ++k) acc += a.dirs[k * MVC + vc] * s.beta[k];
    s.vs[vc] = acc + a.v_template[vc];
  }
  __syncthreads();
  for (int o = wave; o < MJ * 3; o += MNT / 64) {
    const int j = o / 3, c = o - j * 3;
    float acc = 0.f;
    for (int v = lane; v < MV; v += 64) acc += a.j_reg[j * MV + v] * s.vs[v * 3 + c];
    acc = wave_sum(acc);
    if (lane == 0) s.J[o] = acc;
  }
  for (int vc = tid; vc < MVC; vc += MNT) {
    float acc = 0.f;
#pragma unroll 9
    for (int k = 0; k < MPM; ++k) acc += a.dirs[(MB + k) * MVC + vc] * s.pm[k];
    s.vp[vc] = s.vs[vc] + acc;
  }
  __syncthreads();
  // forward kinematics: the wrist, then one thread per finger walks its three joints
  if (tid < 5) {
    float Gp[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) { Gp[r * 4 + 0] = s.R[r * 3 + 0]; Gp[r * 4 + 1] = s.R[r * 3 + 1]; Gp[r * 4 + 2] = s.R[r * 3 + 2]; Gp[r * 4 + 3] = s.J[r]; }
    if (tid == 0) {
#pragma unroll
      for (int e = 0; e < 12; ++e) s.G[e] = Gp[e];
    }
    int par = 0;
    for (int lvl = 0; lvl < 3; ++lvl) {
      const int j = 1 + 3 * tid + lvl;
      const float* Rj = &s.R[j * 9];
      const float t0 = s.J[j * 3 + 0] - s.J[par * 3 + 0], t1 = s.J[j * 3 + 1] - s.J[par * 3 + 1], t2 = s.J[j * 3 + 2] - s.J[par * 3 + 2];
      float Gn[12];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Gn[r * 4 + c] = Gp[r * 4 + 0] * Rj[c] + Gp[r * 4 + 1] * Rj[3 + c] + Gp[r * 4 + 2] * Rj[6 + c];
        Gn[r * 4 + 3] = Gp[r * 4 + 0] * t0 + Gp[r * 4 + 1] * t1 + Gp[r * 4 + 2] * t2 + Gp[r * 4 + 3];
      }
#pragma unroll
      for (int e = 0; e < 12; ++e) { s.G[j * 12 + e] = Gn[e]; Gp[e] = Gn[e]; }
      par = j;
    }
  }
  __syncthreads();
  if (tid < MJ * 3) {                                       // skinning transform: remove the rest-pose joint location
    const int j = tid / 3, r = tid - j * 3;
    const float* g = &s.G[j * 12 + r * 4];
    s.A[j * 12 + r * 4 + 0] = g[0]; s.A[j * 12 + r * 4 + 1] = g[1]; s.A[j * 12 + r * 4 + 2] = g[2];
    s.A[j * 12 + r * 4 + 3] = g[3] - (g[0] * s.J[j * 3] + g[1] * s.J[j * 3 + 1] + g[2] * s.J[j * 3 + 2]);
    s.cat[j * 3 + r] = g[3];
  }
  __syncthreads();
  for (int v = tid; v < MV; v += MNT) {
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    const float4* wv = reinterpret_cast<const float4*>(a.weights + (size_t)v * MJ);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w4 = wv[q];
      const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] += w[i] * s.A[(q * 4 + i) * 12 + e];
    }
    const float p0 = s.vp[v * 3], p1 = s.vp[v * 3 + 1], p2 = s.vp[v * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) s.raw[v * 3 + r] = T[r * 4] * p0 + T[r * 4 + 1] * p1 + T[r * 4 + 2] * p2 + T[r * 4 + 3];
  }
  __syncthreads();
  if (tid < 15) s.cat[MJ * 3 + tid] = s.raw[c_tip[tid / 3] * 3 + tid % 3];
  __syncthreads();
}

// the layer returns millimetres and the head divides by 1000 again (manolayer.py:270-276, mano_head.py:246-247)
__device__ __forceinline__ float to_metres(float x) { return (x * 1000.f) / 1000.f; }

__global__ __launch_bounds__(MNT) void mano_head_fwd_kernel(ManoArgs a) {
  __shared__ ManoLds s;
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  mano_forward(a, h, s);
  const float c0 = s.cat[0], c1 = s.cat[1], c2 = s.cat[2];               // centre = joint 0 of the reordered list = the wrist
  const float cen[3] = {c0, c1, c2};
  const bool lossy = a.mode == 0 && a.gt_verts != nullptr;
  const int hg = lossy ? h % a.gt_hands : 0;
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  for (int vc = tid; vc < MVC; vc += MNT) {
    const float o = to_metres(s.raw[vc] - cen[vc % 3]);
    a.verts[(size_t)h * MVC + vc] = o;
    if (lossy) { const float e = o - a.gt_verts[(size_t)hg * MVC + vc]; part[0] += e * e; }
  }
  if (tid < MJT * 3) {
    const int i = tid / 3, c = tid - i * 3;
    const float o = to_metres(s.cat[c_order[i] * 3 + c] - cen[c]);
    a.joints[(size_t)h * MJT * 3 + tid] = o;
    if (lossy) { const float e = o - a.gt_joints[(size_t)hg * MJT * 3 + tid]; part[1] += e * e; }
  }
  if (tid < MJ * 9) {
    a.rot[(size_t)h * MJ * 9 + tid] = s.R6[tid];
    if (lossy) { const float e = s.R6[tid] - a.gt_rot[(size_t)hg * MJ * 9 + tid]; part[2] += e * e; }
  }
  if (lossy) {
    if (tid < MB) { const float e = s.beta[tid] - a.gt_shape[(size_t)hg * a.ldgt_shape + tid]; part[3] += e * e; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { part[i] = wave_sum(part[i]); if (lane == 0) s.red[wave * 4 + i] = part[i]; }
    __syncthreads();
    if (tid < 4) {
      float t = 0.f;
      for (int wv = 0; wv < MNW; ++wv) t += s.red[wv * 4 + tid];      // wave order: fixed
      a.sums[(size_t)h * 4 + tid] = t;
    }
  }
}

__global__ __launch_bounds__(MNT) void mano_head_bwd_kernel(ManoArgs a) {
  __shared__ ManoLds s;
  __shared__ float gcat[MJT * 3];      // gradient of the 21 uncentred positions (cat order)
  __shared__ float dA[MJ * 12], dG[MJ * 12], dJ[MJ * 3], dR[MJ * 9], dpm[MPM + 1], root[5 * 16], gsum_s[4];
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  mano_forward(a, h, s);
  const float cen[3] = {s.cat[0], s.cat[1], s.cat[2]};
  const bool lossy = a.gt_verts != nullptr && a.g_sums != nullptr;
  const int hg = lossy ? h % a.gt_hands : 0;
  if (tid < 4) gsum_s[tid] = lossy ? a.g_sums[(size_t)h * 4 + tid] : 0.f;
  __syncthreads();
  // gradients of the centred outputs; their sum flows (negated) into the centre = cat[0]
  float cs[3] = {0.f, 0.f, 0.f};
  if (tid < MJT * 3) {
    const int i = tid / 3, c = tid - i * 3;
    float g = a.g_joints ? a.g_joints[(size_t)h * MJT * 3 + tid] : 0.f;
    if (lossy) g += 2.f * gsum_s[1] * (to_metres(s.cat[c_order[i] * 3 + c] - cen[c]) - a.gt_joints[(size_t)hg * MJT * 3 + tid]);
    gcat[c_order[i] * 3 + c] = g;
    cs[c] += g;
  }
  for (int vc = tid; vc < MVC; vc += MNT) {
    float g = a.g_verts ? a.g_verts[(size_t)h * MVC + vc] : 0.f;
    if (lossy) g += 2.f * gsum_s[0] * (to_metres(s.raw[vc] - cen[vc % 3]) - a.gt_verts[(size_t)hg * MVC + vc]);
    s.raw[vc] = g;                      // (a thread only ever touches its own vc here)
    // MVC and MNT: vc % 3 of one thread's elements differ; sort the centre sums by coordinate
    cs[0] += (vc % 3 == 0) ? g : 0.f; cs[1] += (vc % 3 == 1) ? g : 0.f; cs[2] += (vc % 3 == 2) ? g : 0.f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { cs[c] = wave_sum(cs[c]); if (lane == 0) s.red[wave * 4 + c] = cs[c]; }
  __syncthreads();
  if (tid < 3) {
    float t = 0.f;
    for (int wv = 0; wv < MNW; ++wv) t += s.red[wv * 4 + tid];
    gcat[tid] -= t;
  }
  if (tid >= 64 && tid < 64 + 15) { const int k = tid - 64; s.raw[c_tip[k / 3] * 3 + k % 3] += gcat[MJ * 3 + k]; }   // finger tips are vertices
  __syncthreads();
  // skinning backward, vertex side: d posed vertex = T_R^T g
  for (int v = tid; v < MV; v += MNT) {
    float T[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) T[e] = 0.f;
    const float4* wv = reinterpret_cast<const float4*>(a.weights + (size_t)v * MJ);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w4 = wv[q];
      const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) T[r * 3 + c] += w[i] * s.A[(q * 4 + i) * 12 + r * 4 + c];
    }
    const float g0 = s.raw[v * 3], g1 = s.raw[v * 3 + 1], g2 = s.raw[v * 3 + 2];
#pragma unroll
    for (int c = 0; c < 3; ++c) s.vs[v * 3 + c] = T[c] * g0 + T[3 + c] * g1 + T[6 + c] * g2;
  }
  // ... transform side: dA[j][r][c] = sum_v w[v][j] g[v][r] [vp_v ; 1][c] - a wave per output, lanes across the vertices
  // (transposed weights: coalesced), 12 outputs a wave
  for (int o = wave; o < MJ * 12; o += MNW) {
    const int j = o / 12, e = o - j * 12, r = e >> 2, c = e & 3;
    const float* wt = a.dirs + MWT_OFF + (size_t)j * MV;
    float acc = 0.f;
    if (c < 3) { for (int v = lane; v < MV; v += 64) acc += wt[v] * (s.raw[v * 3 + r] * s.vp[v * 3 + c]); }
    else       { for (int v = lane; v < MV; v += 64) acc += wt[v] * s.raw[v * 3 + r]; }
    acc = wave_sum(acc);
    if (lane == 0) dA[o] = acc;
  }
  __syncthreads();
  // A = [G_R | G_t - G_R J]  ->  dG, dJ ; joint positions are the translations
  if (tid < MJ) {
    const int j = tid;
    const float* G = &s.G[j * 12];
    const float at[3] = {dA[j * 12 + 3], dA[j * 12 + 7], dA[j * 12 + 11]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dG[j * 12 + r * 4 + c] = dA[j * 12 + r * 4 + c] - at[r] * s.J[j * 3 + c];
      dG[j * 12 + r * 4 + 3] = at[r] + gcat[j * 3 + r];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dJ[j * 3 + c] = -(G[c] * at[0] + G[4 + c] * at[1] + G[8 + c] * at[2]);
  }
  __syncthreads();
  // kinematic chain backward: a finger thread walks tip-side joint -> knuckle, what reaches the wrist is summed by thread 0
  if (tid < 5) {
    float gR[9], gt[3];                 // gradient of the current joint's world rotation / translation
    float carryR[9], carryT[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) carryR[e] = 0.f;
    carryT[0] = carryT[1] = carryT[2] = 0.f;
    for (int lvl = 2; lvl >= 0; --lvl) {
      const int j = 1 + 3 * tid + lvl, par = lvl == 0 ? 0 : j - 1;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gR[r * 3 + c] = dG[j * 12 + r * 4 + c] + carryR[r * 3 + c];
        gt[r] = dG[j * 12 + r * 4 + 3] + carryT[r];
      }
      const float* Gp = &s.G[par * 12];
      const float* Rj = &s.R[j * 9];
      const float t[3] = {s.J[j * 3] - s.J[par * 3], s.J[j * 3 + 1] - s.J[par * 3 + 1], s.J[j * 3 + 2] - s.J[par * 3 + 2]};
      // local rotation: dR_j = Gp_R^T gR ; local translation: dt = Gp_R^T gt
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) dR[j * 9 + r * 3 + c] = Gp[r] * gR[c] + Gp[4 + r] * gR[3 + c] + Gp[8 + r] * gR[6 + c];
      float dt[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) dt[r] = Gp[r] * gt[0] + Gp[4 + r] * gt[1] + Gp[8 + r] * gt[2];
#pragma unroll
      for (int r = 0; r < 3; ++r) { dJ[j * 3 + r] += dt[r]; }
      // parent: dGp_R = gR R_j^T + gt (x) t ; dGp_t = gt ; dJ_par -= dt
      float nR[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          nR[r * 3 + c] = gR[r * 3] * Rj[c * 3] + gR[r * 3 + 1] * Rj[c * 3 + 1] + gR[r * 3 + 2] * Rj[c * 3 + 2] + gt[r] * t[c];
      if (lvl > 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) carryR[e] = nR[e];
#pragma unroll
        for (int r = 0; r < 3; ++r) { carryT[r] = gt[r]; dJ[par * 3 + r] -= dt[r]; }
      } else {
#pragma unroll
        for (int e = 0; e < 9; ++e) root[tid * 16 + e] = nR[e];
#pragma unroll
        for (int r = 0; r < 3; ++r) { root[tid * 16 + 9 + r] = gt[r]; root[tid * 16 + 12 + r] = dt[r]; }
      }
    }
  }
  __syncthreads();
  if (tid < 9) {
    float acc = dG[(tid / 3) * 4 + tid % 3];
    for (int f = 0; f < 5; ++f) acc += root[f * 16 + tid];
    dR[tid] = acc;                                         // wrist: local = world
  } else if (tid >= 16 && tid < 19) {
    const int r = tid - 16;
    float gt0 = dG[r * 4 + 3], dts = 0.f;
    for (int f = 0; f < 5; ++f) { gt0 += root[f * 16 + 9 + r]; dts += root[f * 16 + 12 + r]; }
    dJ[r] += gt0 - dts;                                    // t_0 = J_0 ; the knuckles' t = J_j - J_0
  }
  // pose blend shapes: d pose_map[k] = sum_vc dvp[vc] dirs[10 + k][vc]
  for (int k = wave; k < MPM; k += MNT / 64) {
    float acc = 0.f;
    const float* dk = a.dirs + (size_t)(MB + k) * MVC;
    for (int vc = lane; vc < MVC; vc += 64) acc += s.vs[vc] * dk[vc];
    acc = wave_sum(acc);
    if (lane == 0) dpm[k] = acc;
  }
  __syncthreads();
  // shaped vertices: dvs = dvp + Jreg^T dJ ; d betas
  for (int v = tid; v < MV; v += MNT) {
    float g0 = s.vs[v * 3], g1 = s.vs[v * 3 + 1], g2 = s.vs[v * 3 + 2];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const float w = a.j_reg[j * MV + v];
      g0 += w * dJ[j * 3]; g1 += w * dJ[j * 3 + 1]; g2 += w * dJ[j * 3 + 2];
    }
    s.vs[v * 3] = g0; s.vs[v * 3 + 1] = g1; s.vs[v * 3 + 2] = g2;
  }
  __syncthreads();
  for (int k = wave; k < MB; k += MNT / 64) {
    float acc = 0.f;
    const float* dk = a.dirs + (size_t)k * MVC;
    for (int vc = lane; vc < MVC; vc += 64) acc += s.vs[vc] * dk[vc];
    acc = wave_sum(acc);
    if (lane == 0) {
      if (lossy) acc += 2.f * gsum_s[3] * (s.beta[k] - a.gt_shape[(size_t)hg * a.ldgt_shape + k]);
      a.d_betas[(size_t)h * MB + k] = acc;
    }
  }
  // rotations: layer gradient + pose-map gradient + explicit / loss gradient on R6, through Gram-Schmidt
  if (tid < MJ) {
    const int j = tid;
    float g[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      float v = dR[j * 9 + e] + (j > 0 ? dpm[(j - 1) * 9 + e] : 0.f);
      if (a.g_rot) v += a.g_rot[(size_t)h * MJ * 9 + j * 9 + e];
      if (lossy) v += 2.f * gsum_s[2] * (s.R6[j * 9 + e] - a.gt_rot[(size_t)hg * MJ * 9 + j * 9 + e]);
      g[e] = v;
    }
    const float* x = a.pose + (size_t)h * a.ldpose + j * 6;
    const float xv[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
    float b1[3], b2[3], b3[3], n1, n2, d;
    gram_schmidt(xv, b1, b2, b3, n1, n2, d);
    float g1[3] = {g[0], g[3], g[6]}, g2[3] = {g[1], g[4], g[7]};
    const float g3[3] = {g[2], g[5], g[8]};
    // b3 = b1 x b2
    g1[0] += b2[1] * g3[2] - b2[2] * g3[1]; g1[1] += b2[2] * g3[0] - b2[0] * g3[2]; g1[2] += b2[0] * g3[1] - b2[1] * g3[0];
    g2[0] += g3[1] * b1[2] - g3[2] * b1[1]; g2[1] += g3[2] * b1[0] - g3[0] * b1[2]; g2[2] += g3[0] * b1[1] - g3[1] * b1[0];
    // b2 = u / |u|
    const float g2b = g2[0] * b2[0] + g2[1] * b2[1] + g2[2] * b2[2];
    const float gu[3] = {(g2[0] - g2b * b2[0]) / n2, (g2[1] - g2b * b2[1]) / n2, (g2[2] - g2b * b2[2]) / n2};
    // u = a2 - (b1 . a2) b1
    const float gub = gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2];
    float* o = a.d_pose + ((size_t)h * MJ + j) * 6;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      o[3 + i] = gu[i] - gub * b1[i];
      g1[i] += -gub * xv[3 + i] - d * gu[i];
    }
    const float g1b = g1[0] * b1[0] + g1[1] * b1[1] + g1[2] * b1[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = (g1[i] - g1b * b1[i]) / n1;
  }
}

__global__ void mano_transpose_dirs_kernel(const float* __restrict__ shapedirs, const float* __restrict__ posedirs,
                                           const float* __restrict__ weights, float* __restrict__ image) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < MDIRS * MVC) {
    const int k = i / MVC, vc = i - k * MVC;
    image[i] = k < MB ? shapedirs[vc * MB + k] : posedirs[vc * MPM + (k - MB)];
  } else if (i < MDIRS * MVC + MJ * MV) {
    const int t = i - MDIRS * MVC, j = t / MV, v = t - j * MV;
    image[i] = weights[v * MJ + j];
  }
}

}  // namespace
}  // namespace hoisdf

using namespace hoisdf;

extern "C" long hoisdf_mano_dirs_image_floats(void) { return (long)MDIRS * MVC + (long)MJ * MV; }

extern "C" int hoisdf_mano_prepare(const float* shapedirs, const float* posedirs, const float* weights, float* image, void* stream) {
  HOISDF_REQUIRE(shapedirs && posedirs && weights && image, HOISDF_ERR_INVALID, "mano_prepare: null pointer");
  hipLaunchKernelGGL(mano_transpose_dirs_kernel, dim3((unsigned)cdiv(hoisdf_mano_dirs_image_floats(), 256)), dim3(256), 0, as_stream(stream),
                     shapedirs, posedirs, weights, image);
  return check_launch("mano_prepare");
}

extern "C" int hoisdf_mano_head_fwd(const float* pose, int ldpose, int mode, const float* betas, int ldbetas, int hands,
                                    const float* dirs_image, const float* v_template, const float* j_regressor, const float* weights,
                                    const float* hands_mean, const float* gt_verts, const float* gt_joints, const float* gt_rot,
                                    const float* gt_shape, int ldgt_shape, int gt_hands, float* verts, float* joints, float* rot,
                                    float* loss_sums, void* stream) {
  HOISDF_REQUIRE(hands >= 0 && (mode == 0 || mode == 1), HOISDF_ERR_INVALID, "mano_head_fwd: hands=%d mode=%d", hands, mode);
  if (hands == 0) return HOISDF_OK;
  HOISDF_REQUIRE(pose && betas && dirs_image && v_template && j_regressor && weights && hands_mean && verts && joints && rot,
                 HOISDF_ERR_INVALID, "mano_head_fwd: null pointer");
  HOISDF_REQUIRE(ldpose >= (mode == 0 ? 96 : 48) && ldbetas >= 10, HOISDF_ERR_INVALID, "mano_head_fwd: ldpose=%d ldbetas=%d", ldpose, ldbetas);
  HOISDF_REQUIRE((reinterpret_cast<uintptr_t>(weights) & 15) == 0, HOISDF_ERR_INVALID, "mano_head_fwd: skinning weights must be 16-byte aligned");
  const bool lossy = gt_verts != nullptr;
  HOISDF_REQUIRE(!lossy || (mode == 0 && gt_joints && gt_rot && gt_shape && loss_sums && gt_hands > 0 && ldgt_shape >= 10),
                 HOISDF_ERR_INVALID, "mano_head_fwd: the fused losses need mode 0, all four ground-truth arrays and loss_sums");
  ManoArgs a{};
  a.pose = pose; a.ldpose = ldpose; a.betas = betas; a.ldbetas = ldbetas; a.mode = mode; a.H = hands;
  a.dirs = dirs_image; a.v_template = v_template; a.j_reg = j_regressor; a.weights = weights; a.hands_mean = hands_mean;
  a.gt_verts = gt_verts; a.gt_joints = gt_joints; a.gt_rot = gt_rot; a.gt_shape = gt_shape; a.ldgt_shape = ldgt_shape; a.gt_hands = gt_hands;
  a.verts = verts; a.joints = joints; a.rot = rot; a.sums = loss_sums;
  hipLaunchKernelGGL(mano_head_fwd_kernel, dim3((unsigned)hands), dim3(MNT), 0, as_stream(stream), a);
  return check_launch("mano_head_fwd");
}

extern "C" int hoisdf_mano_head_bwd(const float* pose6d, const float* betas, int hands, const float* dirs_image, const float* v_template,
                                    const float* j_regressor, const float* weights, const float* hands_mean, const float* gt_verts,
                                    const float* gt_joints, const float* gt_rot, const float* gt_shape, int ldgt_shape, int gt_hands,
                                    const float* g_loss_sums, const float* g_verts, const float* g_joints, const float* g_rot,
                                    float* d_pose6d, float* d_betas, void* stream) {
  HOISDF_REQUIRE(hands >= 0, HOISDF_ERR_INVALID, "mano_head_bwd: hands=%d", hands);
  if (hands == 0) return HOISDF_OK;
  HOISDF_REQUIRE(pose6d && betas && dirs_image && v_template && j_regressor && weights && hands_mean && d_pose6d && d_betas,
                 HOISDF_ERR_INVALID, "mano_head_bwd: null pointer");
  HOISDF_REQUIRE((reinterpret_cast<uintptr_t>(weights) & 15) == 0, HOISDF_ERR_INVALID, "mano_head_bwd: skinning weights must be 16-byte aligned");
  HOISDF_REQUIRE(!g_loss_sums || (gt_verts && gt_joints && gt_rot && gt_shape && gt_hands > 0 && ldgt_shape >= 10), HOISDF_ERR_INVALID,
                 "mano_head_bwd: a loss gradient needs the ground-truth arrays of the forward call");
  ManoArgs a{};
  a.pose = pose6d; a.ldpose = 96; a.betas = betas; a.ldbetas = 10; a.mode = 0; a.H = hands;
  a.dirs = dirs_image; a.v_template = v_template; a.j_reg = j_regressor; a.weights = weights; a.hands_mean = hands_mean;
  a.gt_verts = g_loss_sums ? gt_verts : nullptr; a.gt_joints = gt_joints; a.gt_rot = gt_rot; a.gt_shape = gt_shape;
  a.ldgt_shape = ldgt_shape; a.gt_hands = gt_hands > 0 ? gt_hands : 1;
  a.g_sums = g_loss_sums; a.g_verts = g_verts; a.g_joints = g_joints; a.g_rot = g_rot;
  a.d_pose = d_pose6d; a.d_betas = d_betas;
  hipLaunchKernelGGL(mano_head_bwd_kernel, dim3((unsigned)hands), dim3(MNT), 0, as_stream(stream), a);
  return check_launch("mano_head_bwd");
}
