// (f4) encoder-side auxiliary image losses in one pass (reference main/model.py:128-143 render_gaussian_heatmap,
// :404-422: MSE(reduction none) of the joint heat-map channel, BCE(reduction none) of the hand / object
// segmentation channels).  The reference materialises a (B, 21, 128, 128) Gaussian stack and ~10 elementwise
// kernels; here one thread owns a pixel: 21 exps, three losses, and the same thread layout for the backward.
#include "common.h"

namespace hoisdf {

struct AuxArgs {
  const float* dec;            // decoder_out (B, 3, H, W), arbitrary strides (NCHW or channels_last)
  long sb, sc, sh, sw;
  const float* joints;         // (B, J, 2) pixel coordinates (x, y) in heat-map space
  const float* hand_seg;       // (B, H, W)
  const float* obj_seg;
  int B, J, H, W;
  float inv_sigma;
};

__device__ __forceinline__ float bce(float p, float t) {
  // F.binary_cross_entropy: log terms clamped at -100
  return -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(log1pf(-p), -100.f));
}

__global__ __launch_bounds__(256) void aux_losses_fwd_kernel(AuxArgs a, float* __restrict__ heatmap,
                                                             float* __restrict__ l_hm, float* __restrict__ l_obj,
                                                             float* __restrict__ l_hand) {
  extern __shared__ float js[];                      // this sample's joints
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < a.J * 2; i += 256) js[i] = a.joints[(size_t)b * a.J * 2 + i];
  __syncthreads();
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.H * a.W) return;
  const int y = pix / a.W, x = pix - y * a.W;
  float hm = 0.f;
  for (int j = 0; j < a.J; ++j) {
    const float dx = ((float)x - js[2 * j]) * a.inv_sigma, dy = ((float)y - js[2 * j + 1]) * a.inv_sigma;
    hm += expf(-(dx * dx) * 0.5f - (dy * dy) * 0.5f);
  }
  hm *= 255.f;
  const float* d = a.dec + b * a.sb + y * a.sh + x * a.sw;
  const size_t o = (size_t)b * a.H * a.W + pix;
  const float e = d[0] - hm;
  heatmap[o] = hm;
  l_hm[o] = e * e;
  l_hand[o] = bce(d[a.sc], a.hand_seg[o]);
  l_obj[o] = bce(d[2 * a.sc], a.obj_seg[o]);
}

// d dec[:, 0] = 2 (dec0 - hm) g_hm ; d p = g (p - t) / max((1 - p) p, 1e-12)   (ATen binary_cross_entropy_backward)
__global__ __launch_bounds__(256) void aux_losses_bwd_kernel(AuxArgs a, const float* __restrict__ heatmap,
                                                             const float* __restrict__ g_hm,
                                                             const float* __restrict__ g_obj,
                                                             const float* __restrict__ g_hand, float* __restrict__ ddec) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.H * a.W) return;
  const int y = pix / a.W, x = pix - y * a.W;
  const long off = b * a.sb + y * a.sh + x * a.sw;
  const float* d = a.dec + off;
  float* g = ddec + off;
  const size_t o = (size_t)b * a.H * a.W + pix;
  g[0] = g_hm ? 2.f * (d[0] - heatmap[o]) * g_hm[o] : 0.f;
  const float ph = d[a.sc], po = d[2 * a.sc];
  g[a.sc] = g_hand ? g_hand[o] * (ph - a.hand_seg[o]) / fmaxf((1.f - ph) * ph, 1e-12f) : 0.f;
  g[2 * a.sc] = g_obj ? g_obj[o] * (po - a.obj_seg[o]) / fmaxf((1.f - po) * po, 1e-12f) : 0.f;
}

static int check_aux(const AuxArgs& a, const char* who) {
  HOISDF_REQUIRE(a.dec && a.joints && a.hand_seg && a.obj_seg, HOISDF_ERR_INVALID, "%s: null pointer", who);
  HOISDF_REQUIRE(a.B > 0 && a.J > 0 && a.J <= 1024 && a.H > 0 && a.W > 0 && a.inv_sigma > 0.f, HOISDF_ERR_INVALID,
                 "%s: bad sizes B=%d J=%d H=%d W=%d", who, a.B, a.J, a.H, a.W);
  return 0;
}

}  // namespace hoisdf

using namespace hoisdf;

extern "C" int hoisdf_aux_image_losses_fwd(const float* dec, long sb, long sc, long sh, long sw, const float* joints,
                                           const float* hand_seg, const float* obj_seg, int B, int J, int H, int W,
                                           float sigma, float* heatmap, float* loss_heatmap, float* loss_obj_seg,
                                           float* loss_hand_seg, void* stream) {
  AuxArgs a{dec, sb, sc, sh, sw, joints, hand_seg, obj_seg, B, J, H, W, sigma > 0.f ? 1.f / sigma : 0.f};
  if (int rc = check_aux(a, "aux_image_losses_fwd")) return rc;
  HOISDF_REQUIRE(heatmap && loss_heatmap && loss_obj_seg && loss_hand_seg, HOISDF_ERR_INVALID,
                 "aux_image_losses_fwd: null output");
  hipLaunchKernelGGL(aux_losses_fwd_kernel, dim3(cdiv((long)H * W, 256), B), dim3(256), sizeof(float) * 2 * J,
                     as_stream(stream), a, heatmap, loss_heatmap, loss_obj_seg, loss_hand_seg);
  return check_launch("aux_image_losses_fwd");
}

extern "C" int hoisdf_aux_image_losses_bwd(const float* dec, long sb, long sc, long sh, long sw, const float* hand_seg,
                                           const float* obj_seg, const float* heatmap, const float* g_heatmap,
                                           const float* g_obj_seg, const float* g_hand_seg, int B, int H, int W,
                                           float* ddec, void* stream) {
  AuxArgs a{dec, sb, sc, sh, sw, heatmap /* non-null placeholder for the joints check */, hand_seg, obj_seg, B, 1, H, W, 1.f};
  if (int rc = check_aux(a, "aux_image_losses_bwd")) return rc;
  HOISDF_REQUIRE(heatmap && ddec, HOISDF_ERR_INVALID, "aux_image_losses_bwd: null pointer");
  hipLaunchKernelGGL(aux_losses_bwd_kernel, dim3(cdiv((long)H * W, 256), B), dim3(256), 0, as_stream(stream), a, heatmap,
                     g_heatmap, g_obj_seg, g_hand_seg, ddec);
  return check_launch("aux_image_losses_bwd");
}
