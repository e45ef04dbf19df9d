// hoisdf_sdf_query_fwd: the whole gradient-free SDF query (K1 gather -> K2 input MLP -> K3 posenc -> K4 decoder -> head)
// behind ONE C-ABI call (reference main/model.py:181-244 sdf_forward and :285-354, the body of
// sdf_infer).  What is fused is the DATAFLOW: every intermediate lives in a caller workspace laid out so that no
// concatenation or copy kernel runs between the stages -
//   * the gathered 992 / 3968-wide feature rows can be handed in (shared by the callers that query the same camera
//     points: the reference gathers the same pixels three times, main/model.py:445/486/499) or out;
//   * linear_sdfin's second layer, the positional encoding and xyz are written straight into the decoder-input row
//     x0 = [feat256 | pe30 | xyz3 | 0 0 0], which sits at column 224 of a 516-wide row [h1 (223) | 0 | x0 (292)] (row stride 544); decoder
//     layer 1 writes its 223 outputs (+ one zero column: a padded weight row) at column 0, so the skip-concatenation
//     [h1 | x0] of common/nets/sdf_net.py:104-106 is the row itself - layer 2 contracts all 516 columns with a
//     column-padded weight matrix (zeros under the pad columns);
//   * the weight-norm fold of the four decoder layers is done once by the caller (cached across calls in eval mode).
// The contractions run on the fp32-emulating bf16x3 kernel (gemm_emu.hip; default) or on gemm_f32_kernel (hoisdf_set_gemm_emu(0)): at ~1.7 kFLOP per byte of activations these layers are
// MFMA-bound, and a monolithic kernel that keeps a point tile's 512-wide activations on-chip is limited to 64-row
// tiles by the 160 KB LDS (64 x 512 x 4 B = 128 KB + weight slab), i.e. 30 FLOP per streamed weight byte and two waves
// per SIMD - measured/estimated below the 105-120 TF the tiled GEMM reaches on these shapes.
#include "common.h"

using namespace hoisdf;

namespace {
constexpr int HID0 = 512, LAT = 256, PF = 33, H1 = 223, X0 = LAT + PF, CAT_K = 516, X0_COL = 224;
// row stride of the concatenated row in the workspaces: 544 floats = 17 x 128 bytes (the 516-float stride of round 5 left every row
// and every 64-byte slab segment of the K = 292 / 516 contractions at a 16-byte alignment: -7 ... -10 % on those two layers,
// tools/mb_ld_emu.py); the weights keep their [512][516] layout (include/hoisdf.h), columns 516 .. 543 are never read
constexpr int CAT_LD = 544;
inline long align64(long v) { return (v + 63) / 64 * 64; }
constexpr long EMU_MIN_ROWS_Q = 2048;

// scratch of the emulated form (hoisdf_set_gemm_emu): the bf16x3 slab image of the layer's weight, built right before the layer
inline long emu_scratch_bytes(int C) {
  const long a = hoisdf_linear_emu_image_bytes(HID0, C), b = hoisdf_linear_emu_image_bytes(HID0, CAT_K);
  return ((a > b ? a : b) + 255) / 256 * 256;
}

// one layer: fp32 emulated on the bf16 MFMA pipe (default, large point sets) or the exact-f32 MFMA GEMM.  K_pad >= K: columns K .. K_pad - 1 of x are zero padding (they meet zero weights in the image).
// `given`: the caller's cached image of this weight (hoisdf_sdf_weights.emu_img), NULL = built into emu_img right here
// x_mag / y_mag (f16x2 form): magnitude words (common.h) of x (null: measured) and for y (zero on entry); *y_has says whether the
// emulated kernel ran and filled them
int layer(const float* x, int ldx, const float* W, int ldw, const float* b, float* y, int ldy, long M, int N, int K,
          int K_pad, float drop_p, uint64_t seed, void* emu_img, const void* given, void* stream, const uint32_t* x_mag = nullptr,
          uint32_t* y_mag = nullptr, bool* y_has = nullptr) {
  if (y_has) *y_has = false;
  if (emu_img && M >= EMU_MIN_ROWS_Q && hoisdf_linear_emu_supported(x, ldx, K_pad) && (K_pad + 15) / 16 == (K + 15) / 16) {
    if (!given) { if (int rc = hoisdf_linear_emu_prepare(W, ldw, N, K, 0, emu_img, stream)) return rc; }
    if (y_has) *y_has = y_mag != nullptr;
    return linear_fwd_emu_mag(x, ldx, given ? given : emu_img, b, y, ldy, M, N, K_pad, 1, drop_p, seed, nullptr, x_mag, y_mag, stream);
  }
  return hoisdf_linear_fwd(x, ldx, W, ldw, b, y, ldy, M, N, K, 1, drop_p, seed, nullptr, stream);
}
inline long qmag_bytes(long n_rows) { return align64(5L * n_rows) * 4; }       // row magnitudes (common.h) of feat, ha, cat, h0, h2 behind the image scratch
}  // namespace

extern "C" long hoisdf_sdf_query_workspace(long n_rows, int C, int need_feat) {
  if (n_rows <= 0 || C <= 0) return 0;
  long fl = align64(n_rows * HID0) * 2 + align64(n_rows * CAT_LD);
  if (need_feat) fl += align64(n_rows * (long)C);
  return fl * (long)sizeof(float) + emu_scratch_bytes(C) + qmag_bytes(n_rows);
}

extern "C" int hoisdf_sdf_query_fwd(const hoisdf_pyramid* pyr, const float* points, const int32_t* sample_idx, long n_rows,
                                    int rows_per_sample, const float* center, const float* cam_intr, float scale,
                                    int img_h, int img_w, const float* feat_in, float* feat_out,
                                    const hoisdf_sdf_weights* w, float clamp, float drop_p, uint64_t seed, float* sdf,
                                    float* sdf_raw, float* pe, float* cam_out, void* workspace, long workspace_bytes,
                                    void* stream) {
  HOISDF_REQUIRE(w && points && sdf && sdf_raw, HOISDF_ERR_INVALID, "sdf_query_fwd: null pointer");
  HOISDF_REQUIRE(n_rows >= 0 && n_rows < (1L << 31), HOISDF_ERR_INVALID, "sdf_query_fwd: n_rows=%ld", n_rows);
  HOISDF_REQUIRE(drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID, "sdf_query_fwd: drop_p=%f", drop_p);
  if (n_rows == 0) return HOISDF_OK;
  const int C = w->C;
  HOISDF_REQUIRE(C > 0 && w->dec_ld0 >= X0, HOISDF_ERR_INVALID, "sdf_query_fwd: bad weight descriptor");
  HOISDF_REQUIRE(feat_in != nullptr || pyr != nullptr, HOISDF_ERR_INVALID, "sdf_query_fwd: neither a pyramid nor gathered rows");
  const int own_feat = (feat_in == nullptr && feat_out == nullptr);
  const long need = hoisdf_sdf_query_workspace(n_rows, C, own_feat);
  HOISDF_REQUIRE(workspace && workspace_bytes >= need, HOISDF_ERR_INVALID,
                 "sdf_query_fwd: workspace of %ld bytes, need %ld", workspace_bytes, need);
  float* ws = static_cast<float*>(workspace);
  float* ha = ws;
  float* hb = ha + align64(n_rows * HID0);
  float* cat = hb + align64(n_rows * HID0);
  float* feat_ws = cat + align64(n_rows * CAT_LD);
  void* img = gemm_emu_mode() ? static_cast<void*>(static_cast<char*>(workspace) + (need - emu_scratch_bytes(C) - qmag_bytes(n_rows)))
                                                      : nullptr;
  // row magnitudes (f16x2 form) handed from each kernel to the contraction that reads its output
  uint32_t* qm = (img && emu_form_h2()) ? reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) + (need - qmag_bytes(n_rows))) : nullptr;
  auto mg = [&](int i) -> uint32_t* { return qm ? qm + (long)i * n_rows : nullptr; };
  if (qm && hipMemsetAsync(qm, 0, (size_t)qmag_bytes(n_rows), as_stream(stream)) != hipSuccess) { set_error("sdf_query_fwd: memset failed"); return HOISDF_ERR_LAUNCH; }
  const uint32_t* m_feat = nullptr; bool has = false;
  int rc;
  // K1 (unless the caller shares its gathered rows)
  const float* feat = feat_in;
  if (!feat) {
    float* f = feat_out ? feat_out : feat_ws;
    rc = project_gather_fwd_mag(pyr, points, sample_idx, n_rows, rows_per_sample, center, cam_intr, scale, img_h, img_w,
                                f, C, cam_out, nullptr, mg(0), stream);
    if (rc) return rc;
    feat = f; m_feat = mg(0);
  } else if (cam_out) {
    HOISDF_REQUIRE(false, HOISDF_ERR_INVALID, "sdf_query_fwd: cam_out needs the gather to run here (feat_in given)");
  }
  float* x0 = cat + X0_COL;
  // K2: linear_sdfin (main/model.py:63-69): C -> 512 -> 256, ReLU after both; the second layer lands in x0[:, 0:256]
  rc = layer(feat, C, w->sdfin_w0, C, w->sdfin_b0, ha, HID0, n_rows, HID0, C, C, 0.f, 0, img, w->emu_img[0], stream, m_feat, mg(1), &has);
  if (rc) return rc;
  bool has_cat = false;
  rc = layer(ha, HID0, w->sdfin_w1, HID0, w->sdfin_b1, x0, CAT_LD, n_rows, LAT, HID0, HID0, 0.f, 0, img, w->emu_img[1], stream, has ? mg(1) : nullptr, mg(2), &has_cat);
  if (rc) return rc;
  // K3: posenc + xyz into x0[:, 256:289], pad columns 289..291 zeroed (common/utils/sdf_utils.py:96-141)
  rc = posenc_fwd_mag(points, n_rows, cat, CAT_LD, X0_COL + LAT, pe, has_cat ? mg(2) : nullptr, stream);
  if (rc) return rc;
  // K4: decoder (common/nets/sdf_net.py:87-122); dropout(p) after every hidden ReLU when the module is in train() mode
  // (the reference's detached training-time queries run with it on), stream ids seed + layer
  // (cat's words: x0's columns are complete here; layer 1 adds its columns' before layer 2 reads the whole row)
  rc = layer(x0, CAT_LD, w->dec_w0, w->dec_ld0, w->dec_b0, ha, HID0, n_rows, HID0, X0, X0 + 3, drop_p, seed, img, w->emu_img[2], stream, has_cat ? mg(2) : nullptr, mg(3), &has);   // the three pad columns of x0 are zero
  if (rc) return rc;
  bool has1 = false;
  rc = layer(ha, HID0, w->dec_w1, HID0, w->dec_b1, cat, CAT_LD, n_rows, H1 + 1, HID0, HID0, drop_p, seed + 1, img, w->emu_img[3], stream, has ? mg(3) : nullptr, has_cat ? mg(2) : nullptr, &has1);
  if (rc) return rc;
  rc = layer(cat, CAT_LD, w->dec_w2, CAT_K, w->dec_b2, ha, HID0, n_rows, HID0, CAT_K, CAT_K, drop_p, seed + 2, img, w->emu_img[4], stream, has_cat && has1 ? mg(2) : nullptr, mg(4), &has);
  if (rc) return rc;
  rc = layer(ha, HID0, w->dec_w3, HID0, w->dec_b3, hb, HID0, n_rows, HID0, HID0, HID0, drop_p, seed + 3, img, w->emu_img[5], stream, has ? mg(4) : nullptr, nullptr, nullptr);
  if (rc) return rc;
  return hoisdf_sdf_head_fwd(hb, HID0, w->dec_w4, w->dec_b4, sdf_raw, sdf, n_rows, HID0, clamp, stream);
}

// ---- sdf_infer (main/model.py:246-355) as two calls: count the lattice survivors (one device -> host read of B integers, which
// sizes everything else; hoisdf_sdf_infer_count_begin only QUEUES it - the counts depend on the camera inputs alone, so a host
// can request them ahead of the image encoder and wait for them when it gets here), then lattice fill -> SDF query -> top-k by |sdf| -> gather of the selected points / values / encodings ----
namespace hoisdf {
namespace {
inline long up256(long b) { return (b + 255) & ~255L; }
}  // namespace
}  // namespace hoisdf

extern "C" int hoisdf_sdf_infer_count_begin(const float* center, const float* cam_intr, const float* bbox, float scale, int bins_n, int B,
                                            int32_t* counts_device, int32_t* counts_host, void* stream) {
  HOISDF_REQUIRE(center && cam_intr && bbox && counts_device && counts_host && B > 0 && bins_n > 0, HOISDF_ERR_INVALID,
                 "sdf_infer_count_begin: bad arguments");
  if (int rc = hoisdf_lattice_count(center, cam_intr, bbox, scale, bins_n, B, counts_device, stream)) return rc;
  if (hipMemcpyAsync(counts_host, counts_device, sizeof(int32_t) * B, hipMemcpyDeviceToHost, as_stream(stream)) != hipSuccess) {
    set_error("sdf_infer_count_begin: queueing the read-back failed: %s", hipGetErrorString(hipGetLastError()));
    return HOISDF_ERR_LAUNCH;
  }
  return HOISDF_OK;
}

extern "C" int hoisdf_sdf_infer_count(const float* center, const float* cam_intr, const float* bbox, float scale, int bins_n, int B,
                                      int32_t* counts_device, int32_t* counts_host, long* n_rows, void* stream) {
  HOISDF_REQUIRE(n_rows, HOISDF_ERR_INVALID, "sdf_infer_count: bad arguments");
  if (int rc = hoisdf_sdf_infer_count_begin(center, cam_intr, bbox, scale, bins_n, B, counts_device, counts_host, stream)) return rc;
  if (hipStreamSynchronize(as_stream(stream)) != hipSuccess) {
    set_error("sdf_infer_count: reading the counts back failed: %s", hipGetErrorString(hipGetLastError()));
    return HOISDF_ERR_LAUNCH;
  }
  long n = 0;
  for (int b = 0; b < B; ++b) n += counts_host[b];
  *n_rows = n;
  return HOISDF_OK;
}

namespace hoisdf {
// offsets[b] = sum of counts[0 .. b): one wave, 64 samples per round, the carry in a register
__global__ __launch_bounds__(64) void exclusive_scan_counts_kernel(const int32_t* __restrict__ counts, int B, int32_t* __restrict__ offsets) {
  const int lane = threadIdx.x;
  int carry = 0;
  for (int base = 0; base < B; base += 64) {
    const int b = base + lane;
    const int c = b < B ? counts[b] : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (b < B) offsets[b] = carry + incl - c;
    carry += __shfl(incl, 63, 64);
  }
}
}  // namespace hoisdf

extern "C" long hoisdf_sdf_infer_workspace(long n_rows, int B, int C) {
  if (n_rows < 0 || B <= 0 || C <= 0) return -1;
  // offsets [B] | points [n][3] | sample_idx [n] | lattice_idx [n] | sdf [n] | raw [n] | pe [n][30] | sel [B][k <= n] | query workspace
  return up256(4L * B) + up256(12 * n_rows) + 2 * up256(4 * n_rows) + 2 * up256(4 * n_rows) + up256(120 * n_rows) + up256(4 * n_rows) +
         up256(hoisdf_sdf_query_workspace(n_rows, C, 1)) + 256;
}

extern "C" int hoisdf_sdf_infer(const hoisdf_pyramid* pyr, const float* center, const float* cam_intr, const float* bbox, float scale,
                                int bins_n, int B, const int32_t* counts_device, const int32_t* counts_host, int num_points, int img_h,
                                int img_w, const hoisdf_sdf_weights* w, float clamp, float drop_p, uint64_t seed, float* points_out,
                                float* sdf_out, float* pe_out, void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(pyr && center && cam_intr && bbox && counts_device && counts_host && w && points_out && sdf_out && workspace && B > 0 &&
                     num_points > 0,
                 HOISDF_ERR_INVALID, "sdf_infer: bad arguments");
  long n = 0;
  for (int b = 0; b < B; ++b) {
    // the reference indexes the first num_points of the sorted survivors and fails on a short sample (main/model.py:348)
    HOISDF_REQUIRE(counts_host[b] >= num_points, HOISDF_ERR_INVALID,
                   "sdf_infer: sample %d has only %d lattice points inside its bbox, fewer than num_points=%d", b, counts_host[b], num_points);
    n += counts_host[b];
  }
  const long need = hoisdf_sdf_infer_workspace(n, B, w->C);
  HOISDF_REQUIRE(workspace_bytes >= need, HOISDF_ERR_WORKSPACE, "sdf_infer: workspace of %ld bytes, need %ld", workspace_bytes, need);
  char* p = static_cast<char*>(workspace);
  auto take = [&](long bytes) { char* r = p; p += up256(bytes); return r; };
  int32_t* offsets = reinterpret_cast<int32_t*>(take(4L * B));
  float* pts = reinterpret_cast<float*>(take(12 * n));
  int32_t* sidx = reinterpret_cast<int32_t*>(take(4 * n));
  int32_t* lidx = reinterpret_cast<int32_t*>(take(4 * n));
  float* sdf = reinterpret_cast<float*>(take(4 * n));
  float* raw = reinterpret_cast<float*>(take(4 * n));
  float* pe = reinterpret_cast<float*>(take(120 * n));
  int32_t* sel = reinterpret_cast<int32_t*>(take(4L * B * num_points));
  const long qbytes = hoisdf_sdf_query_workspace(n, w->C, 1);
  void* qws = take(0);
  hipStream_t st = as_stream(stream);
  // exclusive prefix of the counts on the device (no host staging, no synchronisation: this call only enqueues work)
  hipLaunchKernelGGL(exclusive_scan_counts_kernel, dim3(1), dim3(64), 0, st, counts_device, B, offsets);
  if (int rc0 = check_launch("sdf_infer offsets")) return rc0;
  int rc = hoisdf_lattice_fill(center, cam_intr, bbox, scale, bins_n, B, offsets, pts, sidx, lidx, stream);
  if (rc) return rc;
  rc = hoisdf_sdf_query_fwd(pyr, pts, sidx, n, 1, center, cam_intr, scale, img_h, img_w, nullptr, nullptr, w, clamp, drop_p, seed, sdf, raw, pe,
                            nullptr, qws, qbytes, stream);
  if (rc) return rc;
  rc = hoisdf_select_smallest_abs(raw, offsets, counts_device, B, num_points, sel, stream);
  if (rc) return rc;
  const long ns = (long)B * num_points;
  rc = hoisdf_gather_rows(pts, 3, sel, ns, 3, points_out, 3, stream);
  if (rc) return rc;
  rc = hoisdf_gather_rows(sdf, 1, sel, ns, 1, sdf_out, 1, stream);
  if (rc) return rc;
  if (pe_out) rc = hoisdf_gather_rows(pe, 30, sel, ns, 30, pe_out, 30, stream);
  return rc;
}

// ============================================================================================================================
// The TRAINING-time SDF query (main/model.py:181-244 sdf_forward with gradients: the two SDF-loss queries of a step) as one
// call per direction: the dataflow of hoisdf_sdf_query_fwd with the ReLU / dropout sign bitmaps and the activations the backward
// needs kept in a caller-provided `saved` buffer, and hoisdf_sdf_query_bwd = head -> decoder layers 3..0 (the skip
// concatenation's gradient is the layer-2 input gradient itself: layer 0's input gradient is accumulated into its x0 columns)
// -> linear_sdfin -> scatter-add into the pyramid gradient.  Weights: the folded hoisdf_sdf_weights of the forward-only entry;
// the gradients come back in the same (padded) layouts.
// ============================================================================================================================
#include "chain.h"

namespace {
struct TrainSaved { float *feat, *ha, *cat, *h0, *h2, *h3, *raw; uint32_t *ba, *bf, *b0, *b1, *b2, *b3; uint32_t* mag; };
// magnitude words (common.h) of the forward's contraction operands, kept for the grad-weights: feat, ha, cat (x0 is a column slice of
// it: both contractions read the one array, filled in the order the columns are written), h0, h2
enum { TM_FEAT = 0, TM_HA = 1, TM_CAT = 2, TM_H0 = 3, TM_H2 = 4, TM_N = 5 };
inline bool train_mags(const Ctx& c, long n) { return c.emu && emu_form_h2() && n >= EMU_MIN_ROWS; }
inline long bits_words(int N) { return (N + 31) / 32; }
void train_carve(long n, int C, Bump& b, TrainSaved& s) {
  s.feat = b.floats(n * C); s.ha = b.floats(n * HID0); s.cat = b.floats(n * CAT_LD); s.h0 = b.floats(n * HID0); s.h2 = b.floats(n * HID0);
  s.h3 = b.floats(n * HID0); s.raw = b.floats(n);
  s.ba = static_cast<uint32_t*>(b.take(n * bits_words(HID0) * 4)); s.bf = static_cast<uint32_t*>(b.take(n * bits_words(LAT) * 4));
  s.b0 = static_cast<uint32_t*>(b.take(n * bits_words(HID0) * 4)); s.b1 = static_cast<uint32_t*>(b.take(n * bits_words(H1 + 1) * 4));
  s.b2 = static_cast<uint32_t*>(b.take(n * bits_words(HID0) * 4)); s.b3 = static_cast<uint32_t*>(b.take(n * bits_words(HID0) * 4));
  s.mag = static_cast<uint32_t*>(b.take((long)TM_N * n * 4));
}
}  // namespace

extern "C" long hoisdf_sdf_query_train_saved_bytes(long n_rows, int C) {
  if (n_rows <= 0 || C <= 0) return 0;
  Bump b(nullptr, 0); TrainSaved s;
  train_carve(n_rows, C, b, s);
  return b.off + 256;
}

static int sdf_train_forward(const hoisdf_pyramid* pyr, const float* points, const int32_t* sample_idx, long n, int rps, const float* center,
                             const float* cam_intr, float scale, int img_h, int img_w, const hoisdf_sdf_weights* w, float clamp, float drop_p,
                             uint64_t seed, float* sdf, float* pe, float* cam_out, Bump& saved, Bump& ws, bool dry, void* stream) {
  Ctx c{as_stream(stream), stream, &ws, dry, gemm_emu_mode()};
  TrainSaved s;
  const int C = w ? w->C : 1;
  train_carve(n, C, saved, s);
  const bool mags = train_mags(c, n) && !dry && s.mag;
  if (!dry) saved_mags().put(s.mag, mags);                       // (what the backward of this block may rely on: chain.h)
  auto mg = [&](int i) -> uint32_t* { return mags ? s.mag + (long)i * n : nullptr; };
  if (!dry) {
    if (mags) c.rc = hipMemsetAsync(s.mag, 0, (size_t)TM_N * n * 4, c.st) == hipSuccess ? HOISDF_OK : HOISDF_ERR_LAUNCH;
    if (c.ok()) c.rc = project_gather_fwd_mag(pyr, points, sample_idx, n, rps, center, cam_intr, scale, img_h, img_w, s.feat, C, cam_out, nullptr, mg(TM_FEAT), stream);
    if (c.ok()) c.rc = hipMemsetAsync(s.cat, 0, sizeof(float) * n * CAT_LD, c.st) == hipSuccess ? HOISDF_OK : HOISDF_ERR_LAUNCH;   // pad columns
  }
  float* x0 = dry ? nullptr : s.cat + X0_COL;
  // linear_sdfin: C -> 512 -> 256, ReLU after both (main/model.py:63-69)
  lin_fwd(c, s.feat, C, w->sdfin_w0, C, w->emu_img[0], w->sdfin_b0, s.ha, HID0, n, HID0, C, 1, 0.f, 0, s.ba, 0, mg(TM_FEAT), mg(TM_HA));
  lin_fwd(c, s.ha, HID0, w->sdfin_w1, HID0, w->emu_img[1], w->sdfin_b1, x0, CAT_LD, n, LAT, HID0, 1, 0.f, 0, s.bf, 0, mg(TM_HA), mg(TM_CAT));
  if (!dry && c.ok()) c.rc = posenc_fwd_mag(points, n, s.cat, CAT_LD, X0_COL + LAT, pe, mg(TM_CAT), stream);
  // decoder (common/nets/sdf_net.py:87-122); layer i draws dropout stream seed + i
  // (the words of cat: x0's columns are complete here; layer 1 adds its own columns' before layer 2 reads the whole row)
  lin_fwd(c, x0, CAT_LD, w->dec_w0, w->dec_ld0, w->emu_img[2], w->dec_b0, s.h0, HID0, n, HID0, X0, 1, drop_p, seed, s.b0, X0 + 3, mg(TM_CAT), mg(TM_H0));   // as hoisdf_sdf_query_fwd: the three pad columns of x0 are zero
  lin_fwd(c, s.h0, HID0, w->dec_w1, HID0, w->emu_img[3], w->dec_b1, s.cat, CAT_LD, n, H1 + 1, HID0, 1, drop_p, seed + 1, s.b1, 0, mg(TM_H0), mg(TM_CAT));
  lin_fwd(c, s.cat, CAT_LD, w->dec_w2, CAT_K, w->emu_img[4], w->dec_b2, s.h2, HID0, n, HID0, CAT_K, 1, drop_p, seed + 2, s.b2, 0, mg(TM_CAT), mg(TM_H2));
  lin_fwd(c, s.h2, HID0, w->dec_w3, HID0, w->emu_img[5], w->dec_b3, s.h3, HID0, n, HID0, HID0, 1, drop_p, seed + 3, s.b3, 0, mg(TM_H2), nullptr);
  if (!dry && c.ok()) c.rc = hoisdf_sdf_head_fwd(s.h3, HID0, w->dec_w4, w->dec_b4, s.raw, sdf, n, HID0, clamp, stream);
  return c.rc;
}

static int sdf_train_backward(const hoisdf_pyramid_grad* dpyr, const float* points, const int32_t* sample_idx, long n, int rps, const float* center,
                              const float* cam_intr, float scale, int img_h, int img_w, const hoisdf_sdf_weights* w, float clamp, float drop_p,
                              const float* d_sdf, const hoisdf_sdf_weight_grads* G, Bump& saved, Bump& ws, bool dry, void* stream) {
  Ctx c{as_stream(stream), stream, &ws, dry, gemm_emu_mode()};
  TrainSaved s;
  const int C = w ? w->C : 1;
  train_carve(n, C, saved, s);
  float* dh3 = ws.floats(n * HID0); float* dh2 = ws.floats(n * HID0); float* dcat = ws.floats(n * CAT_LD); float* dh0 = ws.floats(n * HID0);
  float* dha = ws.floats(n * HID0); float* dfeat = ws.floats(n * (long)C);
  if (!dry && (!dh3 || !dh2 || !dcat || !dh0 || !dha || !dfeat)) { set_error("sdf_query_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
  // magnitude words of the gradients that feed contractions: dh3, dh2, dcat, dh0, dha (dx0 is accumulated into: measured by its consumers)
  enum { BM_DH3 = 0, BM_DH2 = 1, BM_DCAT = 2, BM_DH0 = 3, BM_DHA = 4, BM_N = 5 };
  uint32_t* bmag = static_cast<uint32_t*>(ws.take((long)BM_N * n * 4));
  const bool mags = train_mags(c, n) && !dry && bmag && s.mag;
  const bool fmags = mags && saved_mags().get(s.mag) != 0;       // (0: the forward of this block ran without them)
  auto mg = [&](int i) -> uint32_t* { return mags ? bmag + (long)i * n : nullptr; };
  auto fm = [&](int i) -> const uint32_t* { return fmags ? s.mag + (long)i * n : nullptr; };
  if (mags) c.rc = hipMemsetAsync(bmag, 0, (size_t)BM_N * n * 4, c.st) == hipSuccess ? HOISDF_OK : HOISDF_ERR_LAUNCH;
  if (!dry && c.ok()) c.rc = sdf_head_bwd_mag(d_sdf, s.raw, s.h3, HID0, w->dec_w4, dh3, HID0, G->d_dec_w4, G->d_dec_b4, n, HID0, clamp, mg(BM_DH3), stream);
  lin_bwd_input(c, dh3, HID0, s.b3, drop_p, w->dec_w3, HID0, w->emu_img_t[5], dh2, HID0, n, HID0, HID0, 0, mg(BM_DH3), mg(BM_DH2));
  lin_bwd_weight(c, dh3, HID0, s.b3, drop_p, s.h2, HID0, G->d_dec_w3, G->d_dec_b3, n, HID0, HID0, 0, mg(BM_DH3), fm(TM_H2));
  lin_bwd_input(c, dh2, HID0, s.b2, drop_p, w->dec_w2, CAT_K, w->emu_img_t[4], dcat, CAT_LD, n, HID0, CAT_K, 0, mg(BM_DH2), mg(BM_DCAT));
  lin_bwd_weight(c, dh2, HID0, s.b2, drop_p, s.cat, CAT_LD, G->d_dec_w2, G->d_dec_b2, n, HID0, CAT_K, 0, mg(BM_DH2), fm(TM_CAT));
  lin_bwd_input(c, dcat, CAT_LD, s.b1, drop_p, w->dec_w1, HID0, w->emu_img_t[3], dh0, HID0, n, H1 + 1, HID0, 0, mg(BM_DCAT), mg(BM_DH0));
  lin_bwd_weight(c, dcat, CAT_LD, s.b1, drop_p, s.h0, HID0, G->d_dec_w1, G->d_dec_b1, n, H1 + 1, HID0, 0, mg(BM_DCAT), fm(TM_H0));
  // layer 0 reads x0 = cat[:, 224:513]: its input gradient joins the skip connection's (accumulate), its weight gradient is [512][292]
  // (contracted over the padded row so that it takes the bf16 pipe like the forward; the three pad columns come out zero)
  float* dx0 = dry ? nullptr : dcat + X0_COL;
  const float* x0 = dry ? nullptr : s.cat + X0_COL;
  lin_bwd_input(c, dh0, HID0, s.b0, drop_p, w->dec_w0, w->dec_ld0, w->emu_img_t[2], dx0, CAT_LD, n, HID0, X0, 1, mg(BM_DH0), nullptr);
  lin_bwd_weight(c, dh0, HID0, s.b0, drop_p, x0, CAT_LD, G->d_dec_w0, G->d_dec_b0, n, HID0, X0, X0 + 3, mg(BM_DH0), fm(TM_CAT));
  // linear_sdfin: its output sits in x0[:, 0:256] (positional encoding / xyz columns carry no gradient)
  lin_bwd_input(c, dx0, CAT_LD, s.bf, 0.f, w->sdfin_w1, HID0, w->emu_img_t[1], dha, HID0, n, LAT, HID0, 0, nullptr, mg(BM_DHA));
  lin_bwd_weight(c, dx0, CAT_LD, s.bf, 0.f, s.ha, HID0, G->d_sdfin_w1, G->d_sdfin_b1, n, LAT, HID0, 0, nullptr, fm(TM_HA));
  lin_bwd_input(c, dha, HID0, s.ba, 0.f, w->sdfin_w0, C, w->emu_img_t[0], dfeat, C, n, HID0, C, 0, mg(BM_DHA), nullptr);
  lin_bwd_weight(c, dha, HID0, s.ba, 0.f, s.feat, C, G->d_sdfin_w0, G->d_sdfin_b0, n, HID0, C, 0, mg(BM_DHA), fm(TM_FEAT));
  if (!dry && c.ok() && dpyr)
    c.rc = hoisdf_project_gather_bwd(dpyr, points, sample_idx, n, rps, center, cam_intr, scale, img_h, img_w, dfeat, C, stream);
  if (c.ok() && !dry && ws.overflow) { set_error("sdf_query_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
  return c.rc;
}

extern "C" long hoisdf_sdf_query_train_workspace_bytes(long n_rows, int C, int backward_pass) {
  if (n_rows <= 0 || C <= 0) return 0;
  hoisdf_sdf_weights w{}; w.C = C; w.dec_ld0 = X0;
  hoisdf_sdf_weight_grads G{};
  Bump saved(nullptr, 0), ws(nullptr, 0);
  if (backward_pass) (void)sdf_train_backward(nullptr, nullptr, nullptr, n_rows, 1, nullptr, nullptr, 1.f, 1, 1, &w, 0.f, 0.f, nullptr, &G, saved, ws, true, nullptr);
  else (void)sdf_train_forward(nullptr, nullptr, nullptr, n_rows, 1, nullptr, nullptr, 1.f, 1, 1, &w, 0.f, 0.f, 0, nullptr, nullptr, nullptr, saved, ws, true, nullptr);
  return ws.off + 256;
}

extern "C" int hoisdf_sdf_query_train_fwd(const hoisdf_pyramid* pyr, const float* points, const int32_t* sample_idx, long n_rows, int rows_per_sample,
                                          const float* center, const float* cam_intr, float scale, int img_h, int img_w,
                                          const hoisdf_sdf_weights* w, float clamp, float drop_p, uint64_t seed, float* sdf, float* pe,
                                          float* cam_out, void* saved, long saved_bytes, void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(pyr && points && w && sdf && saved && workspace && center && cam_intr, HOISDF_ERR_INVALID, "sdf_query_train_fwd: null pointer");
  HOISDF_REQUIRE(n_rows > 0 && n_rows < (1L << 31) && w->C > 0 && w->C % 4 == 0 && w->dec_ld0 >= X0, HOISDF_ERR_INVALID, "sdf_query_train_fwd: bad sizes");
  HOISDF_REQUIRE(drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID, "sdf_query_train_fwd: drop_p=%f", drop_p);
  HOISDF_REQUIRE(saved_bytes >= hoisdf_sdf_query_train_saved_bytes(n_rows, w->C), HOISDF_ERR_WORKSPACE, "sdf_query_train_fwd: saved buffer of %ld bytes, need %ld",
                 saved_bytes, hoisdf_sdf_query_train_saved_bytes(n_rows, w->C));
  Bump sv(saved, saved_bytes), ws(workspace, workspace_bytes);
  int rc = sdf_train_forward(pyr, points, sample_idx, n_rows, rows_per_sample, center, cam_intr, scale, img_h, img_w, w, clamp, drop_p, seed, sdf, pe, cam_out,
                             sv, ws, false, stream);
  if (rc == HOISDF_OK && (sv.overflow || ws.overflow)) rc = HOISDF_ERR_WORKSPACE;
  if (rc == HOISDF_ERR_WORKSPACE) set_error("sdf_query_train_fwd: workspace (%ld bytes) too small", workspace_bytes);
  return rc;
}

extern "C" int hoisdf_sdf_query_bwd(const hoisdf_pyramid_grad* dpyr, const float* points, const int32_t* sample_idx, long n_rows, int rows_per_sample,
                                    const float* center, const float* cam_intr, float scale, int img_h, int img_w, const hoisdf_sdf_weights* w,
                                    float clamp, float drop_p, const void* saved, long saved_bytes, const float* d_sdf,
                                    const hoisdf_sdf_weight_grads* grads, void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(points && w && saved && d_sdf && grads && workspace && center && cam_intr, HOISDF_ERR_INVALID, "sdf_query_bwd: null pointer");
  HOISDF_REQUIRE(grads->d_sdfin_w0 && grads->d_sdfin_b0 && grads->d_sdfin_w1 && grads->d_sdfin_b1 && grads->d_dec_w0 && grads->d_dec_b0 && grads->d_dec_w1 &&
                     grads->d_dec_b1 && grads->d_dec_w2 && grads->d_dec_b2 && grads->d_dec_w3 && grads->d_dec_b3 && grads->d_dec_w4 && grads->d_dec_b4,
                 HOISDF_ERR_INVALID, "sdf_query_bwd: every weight gradient buffer is required (zero-filled)");
  HOISDF_REQUIRE(n_rows > 0 && n_rows < (1L << 31) && saved_bytes >= hoisdf_sdf_query_train_saved_bytes(n_rows, w->C), HOISDF_ERR_INVALID,
                 "sdf_query_bwd: bad sizes / saved buffer");
  Bump sv(const_cast<void*>(saved), saved_bytes), ws(workspace, workspace_bytes);
  int rc = sdf_train_backward(dpyr, points, sample_idx, n_rows, rows_per_sample, center, cam_intr, scale, img_h, img_w, w, clamp, drop_p, d_sdf, grads, sv, ws,
                              false, stream);
  if (rc == HOISDF_ERR_WORKSPACE) set_error("sdf_query_bwd: workspace (%ld bytes) too small", workspace_bytes);
  return rc;
}
