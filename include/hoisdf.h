/* libhoisdf_hip - C ABI of the MI355X (gfx950) HOISDF hot path.
 *
 * The reference (amathislab/HOISDF) has no FFI layer: its boundary for this path is the
 * Python surface Model.forward / sdf_forward / sdf_infer / get_input_transformer /
 * SDFDecoder.forward / Transformer.forward.  This header is the
 * C-ABI a binding for that surface calls into; every entry cites the reference lines it
 * replaces.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - all tensors are caller-allocated DEVICE buffers of float32 unless stated otherwise;
 *     the library never allocates, frees or retains caller memory;
 *   - every entry takes a HIP stream (hipStream_t passed as void*), is asynchronous, does
 *     no hidden synchronisation and keeps no mutable global state;
 *   - return value: 0 = ok, negative = hoisdf_status; the message of the last failure on
 *     the calling thread is available from hoisdf_last_error();
 *   - "rows" are point/token rows; `ld*` are leading dimensions in floats;
 *   - dropout: p in [0,1); the keep mask is a pure function of (seed, element index), the
 *     backward entry regenerates it from the same (p, seed).
 */
#ifndef HOISDF_H_
#define HOISDF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOISDF_VERSION_MAJOR 0
#define HOISDF_VERSION_MINOR 1

typedef enum hoisdf_status {
  HOISDF_OK = 0,
  HOISDF_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, unsupported dim) */
  HOISDF_ERR_LAUNCH = -2,    /* HIP launch / runtime error */
  HOISDF_ERR_TOO_FEW = -3,   /* sdf_infer: fewer bbox survivors than requested points */
  HOISDF_ERR_WORKSPACE = -4  /* workspace too small */
} hoisdf_status;

#define HOISDF_MAX_LEVELS 8

/* Feature pyramid in NHWC (channels-last) layout: level l is [B][H[l]][W[l]][C[l]].
 * Concatenation order = level order (reference: cfg.mutliscale_layers, main/config.py:99). */
typedef struct hoisdf_pyramid {
  int n_levels;
  int B;
  const float* data[HOISDF_MAX_LEVELS];
  int C[HOISDF_MAX_LEVELS];
  int H[HOISDF_MAX_LEVELS];
  int W[HOISDF_MAX_LEVELS];
} hoisdf_pyramid;

typedef struct hoisdf_pyramid_grad {
  int n_levels;
  int B;
  float* data[HOISDF_MAX_LEVELS];
  int C[HOISDF_MAX_LEVELS];
  int H[HOISDF_MAX_LEVELS];
  int W[HOISDF_MAX_LEVELS];
} hoisdf_pyramid_grad;

const char* hoisdf_version(void);
const char* hoisdf_last_error(void);
/* Deterministic mode (also: environment HOISDF_DETERMINISTIC=1, read at first use).  Every entry that otherwise
 * accumulates with float atomics switches to an order-fixed form: two identical call sequences give bit-identical
 * results.  Contract: one stream at a time (the ordered block reductions share a library-owned scratch); grad-weight is
 * order-fixed only when the caller passes the workspace (hoisdf_linear_bwd_weight_workspace). */
void hoisdf_set_deterministic(int on);
/* the contractions inside composite entries as fp32 emulated on the bf16 MFMA pipe (hoisdf_linear_fwd_emu); default ON
 * (environment HOISDF_GEMM=f32 or hoisdf_set_gemm_emu(0): the exact-f32 MFMA kernel) */
void hoisdf_set_gemm_emu(int on);
int hoisdf_get_gemm_emu(void);
int hoisdf_get_deterministic(void);

/* ---- K1: pinhole projection + 5-level bilinear gather --------------------------------
 * reference: main/model.py:148-175 (get_input_transformer), :190-214 (sdf_forward),
 * :286-328 (sdf_infer); F.grid_sample(bilinear, border, align_corners=True).
 * points [n_rows][3] in the scaled SDF frame; row r belongs to sample
 * sample_idx[r] (int32) or r / rows_per_sample when sample_idx is NULL.
 * cam = p/scale + center[b]; uv = (K[b] cam)_xy / (K[b] cam)_z; grid = (uv - n)/n with
 * n = ((img_w-1)/2, (img_h-1)/2).  feat [n_rows][ldf] receives the concatenated levels;
 * cam_out [n_rows][3] and uv_out [n_rows][2] are optional (may be NULL). */
int hoisdf_project_gather_fwd(const hoisdf_pyramid* pyr, const float* points,
                              const int32_t* sample_idx, long n_rows, int rows_per_sample,
                              const float* center, const float* cam_intr, float scale,
                              int img_h, int img_w, float* feat, int ldf, float* cam_out,
                              float* uv_out, void* stream);
/* Scatter-add of dfeat into the (pre-zeroed or accumulating) pyramid gradient. */
int hoisdf_project_gather_bwd(const hoisdf_pyramid_grad* dpyr, const float* points,
                              const int32_t* sample_idx, long n_rows, int rows_per_sample,
                              const float* center, const float* cam_intr, float scale,
                              int img_h, int img_w, const float* dfeat, int ldf, void* stream);

/* ---- K2/K7/K9/K11: fused linear layers (exact fp32 on the f32 MFMA pipe) --------------
 * reference: nn.Linear inside common/nets/layer.py:168-201 (MLP), common/nets/sdf_net.py
 * :87-113 (SDFDecoder hidden layers), nn.MultiheadAttention in/out projections and FFN
 * (common/nets/transformer.py:269-302).
 * y[M][N] = dropout(act(x[M][K] . W[N][K]^T + bias)), act: 0 none, 1 relu.
 * The dropout mask of element (m,n) is a hash of (seed, m, n).
 * relu_bits (optional, [M][ceil(N/32)] uint32): bit n%32 of word [m][n/32] is set iff
 * y[m][n] > 0, i.e. the element survived ReLU and dropout - all the backward needs. */
int hoisdf_linear_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias,
                      float* y, int ldy, long M, int N, int K, int act, float drop_p,
                      uint64_t seed, uint32_t* relu_bits, void* stream);
/* Backward contractions.  If relu_bits (from the forward) is given, the ReLU + dropout backward
 * is fused into the load of dy: dy_eff = dy * bit / (1 - p); NULL means dy is used as is.
 * dx[M][K] = dy_eff[M][N] . W[N][K] */
int hoisdf_linear_bwd_input(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p,
                            const float* W, int ldw, float* dx, int lddx, long M, int N, int K,
                            int accumulate /* 1: dx += (the gradient another consumer of x already left there) */,
                            void* stream);
/* dW[N][K] = dy_eff[M][N]^T . x[M][K] ; db[N] = column sums of dy_eff (db may be NULL).
 * Split-K over M.  With a workspace of hoisdf_linear_bwd_weight_workspace(M,N,K) floats the
 * partial tiles are written there and summed by a second kernel: dW (dense, lddw == K) and db
 * are fully overwritten.  Without it (workspace == NULL) the kernel accumulates with float
 * atomics into dW / db, which the caller must have zero-filled (or hold gradients to add to). */
long hoisdf_linear_bwd_weight_workspace(long M, int N, int K);
int hoisdf_linear_bwd_weight(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p,
                             const float* x, int ldx, float* dW, int lddw, float* db, long M, int N,
                             int K, float* workspace, long workspace_floats, void* stream);
/* ---- fp32 linear layers emulated on the bf16 MFMA pipe ("bf16x3"; cfg.gemm_emu) ------------------------------------------
 * reference: the same call sites as hoisdf_linear_fwd / hoisdf_linear_bwd_input (common/nets/layer.py:168-201,
 * common/nets/transformer.py:286-302, main/model.py:56-90).  Same contracts and argument meaning, fp32-equivalent results:
 * every f32 operand is split EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits, no scaling - bf16 has the f32
 * exponent range) and each product is accumulated in f32 from six bf16 MFMA products (the three dropped cross terms are
 * <= 2^-24 of the product); error against fp64 = that of the exact-f32 entries (tests/test_gpu_emu.py), not bit-identical
 * to them (another summation order).  The weight operand is handed over as a pre-split "slab image"
 * (hoisdf_linear_emu_image_bytes(rows, K) bytes, 16-byte aligned, caller-owned):
 *   hoisdf_linear_emu_prepare(W, ldw, N, K, 0, image)  -> image for hoisdf_linear_fwd_emu       (rows = N, contraction K)
 *   hoisdf_linear_emu_prepare(W, ldw, N, K, 1, image)  -> image for hoisdf_linear_bwd_input_emu (rows = K, contraction N)
 * to be rebuilt whenever W changes (once per optimizer step).  The activation operand must be 16-byte aligned with a leading
 * dimension and a contraction length that are multiples of 4 (hoisdf_linear_emu_supported); other shapes stay on the f32
 * entries. */
long hoisdf_linear_emu_image_bytes(int rows, int K);
int hoisdf_linear_emu_prepare(const float* W, int ldw, int N, int K, int transpose, void* image, void* stream);
/* All images of a model in ONE launch (after an optimizer step: ~170 weights x two orientations in the reference's hot path,
 * a few microseconds each when built one by one).  `d_items` is a DEVICE table of n items, first_block ascending from 0
 * (item i occupies hoisdf_linear_emu_prepare_blocks(N, K, transpose) blocks), total_blocks = the sum. */
typedef struct hoisdf_emu_prep_item {
  const float* W;          /* [N][ldw] */
  void* image;             /* hoisdf_linear_emu_image_bytes(transpose ? K : N, transpose ? N : K) bytes, 16-byte aligned */
  long first_block;
  int ldw, N, K, transpose;
} hoisdf_emu_prep_item;
long hoisdf_linear_emu_prepare_blocks(int N, int K, int transpose);
int hoisdf_linear_emu_prepare_batch(const hoisdf_emu_prep_item* d_items, int n, long total_blocks, void* stream);
int hoisdf_linear_emu_supported(const float* a, long lda, int contraction);
/* ---- the two arithmetic forms of the emulated entries (process-wide; environment HOISDF_EMU_FORM, read at first use) -----------
 * "b3"  bf16x3: the exact three-piece split described above, six products per product, any operand range.
 * "h2"  f16x2 (default since round 5): every operand is scaled by a power of two s that puts its largest magnitude in
 *       [2^13, 2^14) and split into TWO f16 pieces, x s = hi + lo + r with |r| <= max(2^-22 |x s|, 2^-25): 22 significant bits
 *       (on average 2^-24 relative, the rounding of an f32 operation) for every element within 2^-16 of the largest one sharing its
 *       scale, an ABSOLUTE error of 2^-38 of that largest one below (the low piece is an f16 subnormal there); each product is
 *       accumulated in f32 from THREE f16 MFMA products (lo hi + hi lo + hi hi; the dropped lo lo term is <= 2^-22 of the product) and
 *       the result is scaled back.  Half the matrix-pipe work of b3.  Error against fp64 on the model's operands: that of b3 and of
 *       the exact-f32 entries (tests/test_gpu_emu.py, test_gpu_bench_geometry.py).
 *       WHO SHARES A SCALE (round 6): a weight matrix has one scale (kept in its image).  The row operand of a forward / grad-input
 *       contraction has ONE SCALE PER ROW (token): a row's rounding depends on that row alone, so the samples of a batch do not
 *       influence each other's results (sample 0 of a batch is bit-identical whatever the other samples are) and a row far below
 *       the matrix maximum keeps its 22 bits.  The grad-weight contracts over the rows: each of its row slices takes the largest
 *       magnitude of its own rows.  The attention operands (Q, K, V, dO) have one scale per (sample, head).
 *       The scales need the magnitudes at launch time.  ROW MAGNITUDES of a matrix of M rows: hoisdf_mag_words(M) = M u32 words, word
 *       r = the IEEE bits of max |x[r][:]| (or of an upper bound), ZERO-FILLED before the producer(s) run: the *_mag entries below
 *       fold their output's into y_mag / dx_mag with an unsigned atomic max per row and column tile (not with accumulate = 1), so do
 *       the LayerNorm, gather, positional-encoding, SDF-head and attention entries of the composite calls.  Without words (x_mag =
 *       NULL, and in the plain entries) the library measures the operand itself: one more read of it (hoisdf_mag_measure does the
 *       same once for several consumers).  HEAD MAGNITUDES of an attention operand matrix (rows = B samples x L tokens, groups of 64
 *       columns = heads): hoisdf_head_mag_words(M, groups, L) words, word[group * B + sample]; zero-filled, then folded by
 *       hoisdf_linear_fwd_emu_heads / the composite entries, or made by hoisdf_head_mag_measure.
 *       A word SMALLER than the true maximum (a stale array) cannot make Inf / NaN: the splitting kernels run with f16 saturation
 *       on (MODE.FP16_OVFL) - up to 4x too small is harmless (head room of the [2^13, 2^14) target), beyond that the row's largest
 *       elements clip at 65504 / scale.
 * hoisdf_linear_emu_pieces() = 2 (h2) or 3 (b3).  The image format follows the form: images are built and consumed in one process. */
int hoisdf_linear_emu_pieces(void);
long hoisdf_mag_words(long rows);
/* the row magnitudes of a matrix whose producer left none: every word is written (no clearing needed); one read of x */
int hoisdf_mag_measure(const float* x, long ldx, long M, int K, uint32_t* words, void* stream);
long hoisdf_head_mag_words(long M, int groups, int L);
/* the head magnitudes of x[M][>= groups * 64] (samples of L consecutive rows): clears `words`, then one read of x */
int hoisdf_head_mag_measure(const float* x, long ldx, long M, int groups, int L, uint32_t* words, void* stream);
/* hoisdf_linear_fwd_emu_mag that also folds the head magnitudes of y (N % 64 == 0; zero-filled `y_heads`, samples of L rows) - what an
 * attention in-projection leaves for hoisdf_attention_fwd_emu_mag; y_mag may be NULL */
int hoisdf_linear_fwd_emu_heads(const float* x, int ldx, const void* w_image, const float* bias, float* y, int ldy, long M, int N,
                                int K, const uint32_t* x_mag, uint32_t* y_mag, uint32_t* y_heads, int L, void* stream);
/* the same for a grad-input (no sign bitmap, no accumulation): what an out-projection's backward leaves for the dO operand of
 * hoisdf_attention_bwd_emu_mag (K % 64 == 0: the heads are the columns of dx) */
int hoisdf_linear_bwd_input_emu_heads(const float* dy, int lddy, const void* wt_image, float* dx, int lddx, long M, int N, int K,
                                      const uint32_t* dy_mag, uint32_t* dx_mag, uint32_t* dx_heads, int L, void* stream);
int hoisdf_linear_fwd_emu_mag(const float* x, int ldx, const void* w_image, const float* bias, float* y, int ldy, long M, int N,
                              int K, int act, float drop_p, uint64_t seed, uint32_t* relu_bits, const uint32_t* x_mag,
                              uint32_t* y_mag, void* stream);
int hoisdf_linear_bwd_input_emu_mag(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const void* wt_image,
                                    float* dx, int lddx, long M, int N, int K, int accumulate, const uint32_t* dy_mag,
                                    uint32_t* dx_mag, void* stream);
/* grad-weight in the f16x2 form (h2 processes, 256-wide tiles; otherwise identical to hoisdf_linear_bwd_weight_emu, whose
 * bf16x3 arithmetic needs no magnitudes and stays available in either process form): dy_mag / x_mag as above, NULL = measured. */
int hoisdf_linear_bwd_weight_emu_mag(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x, int ldx,
                                     float* dW, int lddw, float* db, long M, int N, int K, float* workspace, long workspace_floats,
                                     const uint32_t* dy_mag, const uint32_t* x_mag, void* stream);
int hoisdf_linear_fwd_emu(const float* x, int ldx, const void* w_image, const float* bias, float* y, int ldy, long M, int N,
                          int K, int act, float drop_p, uint64_t seed, uint32_t* relu_bits, void* stream);
int hoisdf_linear_bwd_input_emu(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const void* wt_image,
                                float* dx, int lddx, long M, int N, int K, int accumulate, void* stream);
/* The same three contractions for SMALL row counts (M <= hoisdf_linear_emu_small_max_rows() = 2047: the 17-query decoder stack,
 * common/nets/transformer.py:366-395, and the regression heads - ~100 calls of 544 x 256 x 256 per training step that the tiled
 * kernels cut into a handful of tiles): one wave per 32 x 32 output tile, operands read from the f32 matrices themselves (no
 * weight image, no workspace), same arithmetic (exact three-way bf16 split, six products, f32 accumulation), same epilogues and
 * sign-map convention as the entries above.  Operands 16-byte aligned, leading dims / N / K multiples of 4
 * (hoisdf_linear_emu_small_supported).  The grad-weight form OVERWRITES dW (any lddw >= K) and db, order-fixed (no atomics). */
int hoisdf_linear_emu_small_max_rows(void);
int hoisdf_linear_emu_small_supported(const float* a, long lda, const float* W, long ldw, long M, int N, int K);
int hoisdf_linear_fwd_emu_small(const float* x, int ldx, const float* W, int ldw, const float* bias, float* y, int ldy, long M,
                                int N, int K, int act, float drop_p, uint64_t seed, uint32_t* relu_bits, void* stream);
int hoisdf_linear_bwd_input_emu_small(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* W, int ldw,
                                      float* dx, int lddx, long M, int N, int K, int accumulate, void* stream);
int hoisdf_linear_bwd_weight_emu_small(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x, int ldx,
                                       float* dW, int lddw, float* db, long M, int N, int K, void* stream);
/* dW[N][K] = dy_eff[M][N]^T . x[M][K], db[N] = column sums of dy_eff (db may be NULL), same emulation: both activation operands
 * are split into bf16 triples inside the kernel (a register transpose per 4-column x 8-row patch, no transposed copy through
 * HBM).  dW (dense: lddw == K) and db are fully OVERWRITTEN; the rows are split over the workgroups into partial tiles in
 * `workspace` (hoisdf_linear_bwd_weight_emu_workspace(M, N, K) floats, 16-byte aligned) and summed in slice order: no atomics,
 * run-to-run identical.  N, K, lddy, ldx multiples of 4, 16-byte aligned operands. */
long hoisdf_linear_bwd_weight_emu_workspace(long M, int N, int K);
int hoisdf_linear_bwd_weight_emu(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x, int ldx,
                                 float* dW, int lddw, float* db, long M, int N, int K, float* workspace, long workspace_floats,
                                 void* stream);
/* dpre = dy * (y > 0) * 1/(1-p): backward of relu followed by dropout, given the
 * post-dropout output y (an element is kept-and-positive iff y > 0).  In place allowed. */
int hoisdf_relu_dropout_bwd(const float* y, int ldy, const float* dy, int lddy, float* dpre,
                            int ldd, long M, int N, float drop_p, void* stream);

/* ---- K3/K4: SDF decoder pieces ---------------------------------------------------------
 * reference: common/utils/sdf_utils.py:96-141 (Embedder), main/model.py:218-228 (decoder
 * input [feat | posenc | xyz]), common/nets/sdf_net.py:57-62 (weight norm).
 * Writes columns [col0, col0+30) = posenc(points) and [col0+30, col0+33) = points of the
 * decoder-input rows x0 (ld = ldx0, trailing pad columns up to ldx0 zeroed), and the
 * stand-alone posenc rows pe [n_rows][30] (may be NULL). */
int hoisdf_posenc_fwd(const float* points, long n_rows, float* x0, int ldx0, int col0,
                      float* pe, void* stream);
/* W[r][:] = g[r] * v[r][:] / ||v[r][:]||, written with leading dimension ldw (>= in);
 * pad columns are zeroed.  norms[r] (optional) receives ||v[r]||. */
int hoisdf_weightnorm_fwd(const float* v, const float* g, float* W, int ldw, float* norms,
                          int out, int in, void* stream);
/* dv, dg from dW (same ldw): dg[r] = <dW[r], v[r]>/||v[r]||,
 * dv[r] = g[r]/||v[r]|| * (dW[r] - v[r] <dW[r], v[r]>/||v[r]||^2). */
int hoisdf_weightnorm_bwd(const float* v, const float* g, const float* dW, int ldw, float* dv,
                          float* dg, int out, int in, void* stream);
/* Final 512 -> 1 layer + tanh + clamp (common/nets/sdf_net.py:115-122, main/model.py:241).
 * sdf_raw = tanh(h . w + b) (unclamped, what sdf_infer sorts on); sdf = clamp(sdf_raw). */
int hoisdf_sdf_head_fwd(const float* h, int ldh, const float* w, const float* b, float* sdf_raw,
                        float* sdf, long n_rows, int K, float clamp, void* stream);
int hoisdf_sdf_head_bwd(const float* dsdf, const float* sdf_raw, const float* h, int ldh,
                        const float* w, float* dh, int lddh, float* dw, float* db, long n_rows,
                        int K, float clamp, void* stream);

/* ---- K1-K4 behind one call: the gradient-free SDF query ----------------------------------------------------
 * reference: Model.sdf_forward (main/model.py:181-244) and the body of Model.sdf_infer (:285-354) - project + gather,
 * linear_sdfin, positional encoding, SDFDecoder, tanh + clamp - for the call sites whose results are only ever used
 * detached (:483-484,:517-518,:540,:558) and for inference.  SURVEY.md section 8(b) `hoisdf_sdf_query_fwd`.
 * Weights: plain device pointers; the four decoder matrices are EFFECTIVE weights (weight-norm folded,
 * hoisdf_weightnorm_fwd), two of them laid out for the in-place skip-concatenation:
 *   dec_w1 [224][512]  rows 0..222 = layer 1, row 223 = 0 (and dec_b1[223] = 0): the extra output is the zero pad column;
 *   dec_w2 [512][516]  columns 0..222 = W2[:, 0:223] (h1), 223 = 0, 224..512 = W2[:, 223:512] (x0), 513..515 = 0.
 * feat_in  (optional) [n_rows][C]: rows already gathered for these camera points (skips K1; cam_out must be NULL);
 * feat_out (optional) [n_rows][C]: receives the gathered rows for other consumers of the same points.
 * drop_p / seed: dropout after the decoder's hidden ReLUs (the module's train() mode; 0 in eval), layer i draws
 * stream seed + i of the counter hash.
 * Outputs: sdf (clamped), sdf_raw (tanh, unclamped: what sdf_infer ranks on), pe [n_rows][30] (optional),
 * cam_out [n_rows][3] (optional).  workspace: hoisdf_sdf_query_workspace(n_rows, C, need_feat) bytes, need_feat = 1
 * when neither feat_in nor feat_out is given. */
typedef struct hoisdf_sdf_weights {
  int C;                                  /* pyramid channels (992 / 3968) */
  const float *sdfin_w0, *sdfin_b0;       /* [512][C], [512] */
  const float *sdfin_w1, *sdfin_b1;       /* [256][512], [256] */
  const float *dec_w0, *dec_b0;           /* [512][dec_ld0 >= 289], [512] */
  int dec_ld0;
  const float *dec_w1, *dec_b1;           /* [224][512], [224] */
  const float *dec_w2, *dec_b2;           /* [512][516], [512] */
  const float *dec_w3, *dec_b3;           /* [512][512], [512] */
  const float *dec_w4, *dec_b4;           /* [512], [1] */
  /* optional (NULL = built per call in the workspace): bf16x3 slab images (hoisdf_linear_emu_prepare) of the six matrices
   * sdfin_w0, sdfin_w1, dec_w0 .. dec_w3 in this order - emu_img for the forward GEMMs, emu_img_t (transpose = 1) for the
   * grad-input GEMMs of hoisdf_sdf_query_bwd.  The caller rebuilds them whenever it re-folds the weights (one
   * hoisdf_linear_emu_prepare_batch launch); a training step otherwise builds 48 images inside these entries. */
  const void* emu_img[6];
  const void* emu_img_t[6];
} hoisdf_sdf_weights;
long hoisdf_sdf_query_workspace(long n_rows, int C, int need_feat);
int hoisdf_sdf_query_fwd(const hoisdf_pyramid* pyr, const float* points, const int32_t* sample_idx, long n_rows,
                         int rows_per_sample, const float* center, const float* cam_intr, float scale, int img_h,
                         int img_w, const float* feat_in, float* feat_out, const hoisdf_sdf_weights* w, float clamp,
                         float drop_p, uint64_t seed, float* sdf, float* sdf_raw, float* pe, float* cam_out,
                         void* workspace, long workspace_bytes, void* stream);

/* ---- K5/K6: dense-grid candidate generation + selection (sdf_infer) -------------------
 * reference: main/model.py:257-302 (sheared lattice + strict bbox filter, on CPU there) and
 * :345-352 (sort by |sdf|, keep the first num_points).
 * Pass 1 (count) + pass 2 (fill) stream compaction of the bins_n^3 lattice per sample:
 *   counts [B] int32 survivors per sample;
 *   with offsets [B] (exclusive prefix of counts) the fill pass writes, for each survivor in
 *   ascending lattice order, points [n][3] (scaled frame), sample_idx [n] and lattice_idx [n].*/
int hoisdf_lattice_count(const float* center, const float* cam_intr, const float* bbox, float scale,
                         int bins_n, int B, int32_t* counts, void* stream);
int hoisdf_lattice_fill(const float* center, const float* cam_intr, const float* bbox, float scale,
                        int bins_n, int B, const int32_t* offsets, float* points,
                        int32_t* sample_idx, int32_t* lattice_idx, void* stream);
/* Per sample b, select the k rows with the smallest |sdf_raw| among rows
 * [offsets[b], offsets[b]+counts[b]) (ties: lower row first) and emit them in ascending
 * (|sdf|, row) order: sel [B][k] int32 global row indices.  Exact (radix select + rank). */
int hoisdf_select_smallest_abs(const float* sdf_raw, const int32_t* offsets, const int32_t* counts,
                               int B, int k, int32_t* sel, void* stream);
/* out[r][0:width] = src[sel[r]][0:width] */
int hoisdf_gather_rows(const float* src, int lds, const int32_t* sel, long n_sel, int width,
                       float* out, int ldo, void* stream);
/* ---- the training-time SDF query, one call per direction (SURVEY.md section 8(b) `hoisdf_sdf_query_bwd`) ------------------
 * reference: Model.sdf_forward with gradients (main/model.py:181-244; the two SDF-loss queries of a training step).
 * hoisdf_sdf_query_train_fwd = the dataflow of hoisdf_sdf_query_fwd keeping, in `saved` (hoisdf_sdf_query_train_saved_bytes),
 *   what the backward needs: gathered rows, activations, the six ReLU / dropout sign bitmaps, the tanh output.
 * hoisdf_sdf_query_bwd: d_sdf [n_rows] (gradient of the CLAMPED sdf) -> weight gradients in the layouts of hoisdf_sdf_weights
 *   (d_dec_w0 [512][292]: the 289 real columns + the three zero pad columns of the decoder-input row; d_dec_w1 [224][512] and d_dec_b1 [224] with the pad row; d_dec_w2 [512][516] with the pad
 *   columns; all zero on entry), and the pyramid gradient scatter-ADDED into dpyr (NULL: skipped).  The weight-norm fold's own
 *   backward (hoisdf_weightnorm_bwd on the effective-weight gradients) is the caller's. */
typedef struct hoisdf_sdf_weight_grads {
  float *d_sdfin_w0, *d_sdfin_b0, *d_sdfin_w1, *d_sdfin_b1;
  float *d_dec_w0, *d_dec_b0, *d_dec_w1, *d_dec_b1, *d_dec_w2, *d_dec_b2, *d_dec_w3, *d_dec_b3, *d_dec_w4, *d_dec_b4;
} hoisdf_sdf_weight_grads;
long hoisdf_sdf_query_train_saved_bytes(long n_rows, int C);
long hoisdf_sdf_query_train_workspace_bytes(long n_rows, int C, int backward_pass);
int hoisdf_sdf_query_train_fwd(const hoisdf_pyramid* pyr, const float* points, const int32_t* sample_idx, long n_rows,
                               int rows_per_sample, const float* center, const float* cam_intr, float scale, int img_h, int img_w,
                               const hoisdf_sdf_weights* w, float clamp, float drop_p, uint64_t seed, float* sdf, float* pe,
                               float* cam_out, void* saved, long saved_bytes, void* workspace, long workspace_bytes, void* stream);
int hoisdf_sdf_query_bwd(const hoisdf_pyramid_grad* dpyr, const float* points, const int32_t* sample_idx, long n_rows,
                         int rows_per_sample, const float* center, const float* cam_intr, float scale, int img_h, int img_w,
                         const hoisdf_sdf_weights* w, float clamp, float drop_p, const void* saved, long saved_bytes,
                         const float* d_sdf, const hoisdf_sdf_weight_grads* grads, void* workspace, long workspace_bytes,
                         void* stream);

/* ---- sdf_infer in two calls (SURVEY.md section 8(b) `hoisdf_sdf_infer`) ----------------------------------------------
 * reference: Model.sdf_infer (main/model.py:246-355): dense sheared lattice inside the bbox -> SDF of every survivor ->
 * the num_points survivors of every sample with the smallest |sdf| (ascending) -> their points (scaled frame), clamped
 * SDF values and positional encodings.
 * hoisdf_sdf_infer_count_begin: queues the survivor count per sample into counts_device [B] and its copy into counts_host [B]
 *   (page-locked host memory that stays alive until the stream has executed the copy) and RETURNS WITHOUT WAITING.  The counts
 *   depend only on center / cam_intr / bbox (main/model.py:286-302), so a host can issue this ahead of the image encoder,
 *   record an event behind it and wait for that event when it reaches sdf_infer - no pipeline drain.
 * hoisdf_sdf_infer_count: the same followed by hipStreamSynchronize(stream) - the ONE blocking device -> host read of the
 *   path, for hosts that do not overlap it; *n_rows = the total.  The caller sizes the workspace with
 *   hoisdf_sdf_infer_workspace(n_rows, B, C) and calls
 * hoisdf_sdf_infer: exclusive scan of the counts (device) -> lattice fill -> hoisdf_sdf_query_fwd ->
 *   hoisdf_select_smallest_abs -> hoisdf_gather_rows.  It only enqueues work (no host <-> device copy, no synchronisation);
 *   counts_host is read on the host for validation and sizing.  A sample with fewer than num_points survivors is refused
 *   (the reference raises at main/model.py:348).
 *   points_out [B][num_points][3], sdf_out [B][num_points], pe_out [B][num_points][30] (optional). */
int hoisdf_sdf_infer_count_begin(const float* center, const float* cam_intr, const float* bbox, float scale, int bins_n,
                                 int B, int32_t* counts_device, int32_t* counts_host, void* stream);
int hoisdf_sdf_infer_count(const float* center, const float* cam_intr, const float* bbox, float scale, int bins_n, int B,
                           int32_t* counts_device, int32_t* counts_host, long* n_rows, void* stream);
long hoisdf_sdf_infer_workspace(long n_rows, int B, int C);
int hoisdf_sdf_infer(const hoisdf_pyramid* pyr, const float* center, const float* cam_intr, const float* bbox, float scale,
                     int bins_n, int B, const int32_t* counts_device, const int32_t* counts_host, int num_points, int img_h,
                     int img_w, const hoisdf_sdf_weights* w, float clamp, float drop_p, uint64_t seed, float* points_out,
                     float* sdf_out, float* pe_out, void* workspace, long workspace_bytes, void* stream);

/* ---- (f2) dataset-side SDF sample selection on the device ------------------------------------------------
 * reference: data/dexycb.py:514-546 (np.random.choice without replacement of num_samp_hand / num_samp_obj rows of
 * the frame's sdf_processed array, and - training - of the rows with |sdf| < points_filter_dist).
 * rows: HBM-resident store of sdf_processed rows [.. ld >= 6 floats: x y z sdf_hand sdf_obj label].  A segment s is
 * the row range [seg_row0[s], seg_row0[s] + seg_len[s]) of the store; seg_col[s] = 3 / 4 keeps only rows with
 * |row[col]| < dist eligible, -1 = all rows.  Writes keys[seg_off[s] + i] = uniform [0,1) (hash of seed, s, i) for
 * eligible rows, 1e30 otherwise, and eligible[s] = number of eligible rows.  Then
 * hoisdf_select_smallest_abs(keys, seg_off, seg_len, n_seg, k, sel) yields k uniformly drawn rows per segment
 * (indices into keys; the caller checks eligible[s] >= k) and hoisdf_gather_rows fetches them. */
int hoisdf_sdf_sample_keys(const float* rows, int ld, const int64_t* seg_row0, const int32_t* seg_len,
                           const int32_t* seg_off, const int32_t* seg_col, int n_seg, int max_len, float dist,
                           uint64_t seed, float* keys, int32_t* eligible, void* stream);

/* ---- K8: sigma gate + token assembly ----------------------------------------------------
 * reference: main/model.py:123-126 (sdf_activation), :520-562 (token concat).
 * For sample b and point p:  tok[b][row0+p][:] = [cam[b][p]-center[b] (3) | pe (30) |
 * feat[b][p][0:F] * sigmoid(sdf/beta)/beta],  beta = max(*beta_ptr, 2e-3), F = D - 33.
 * tok is batch-first [B][S][D]. */
int hoisdf_token_build_fwd(const float* cam, const float* center, const float* pe,
                           const float* feat, int ldfeat, const float* sdf, const float* beta_ptr,
                           float* tok, int B, int P, int S, int row0, int D, void* stream);
/* dfeat = dtok[33:] * sigma; *dbeta += sum dtok[33:] * feat * dsigma/dbeta (atomic). */
int hoisdf_token_build_bwd(const float* dtok, const float* feat, int ldfeat, const float* sdf,
                           const float* beta_ptr, float* dfeat, int lddfeat, float* dbeta, int B,
                           int P, int S, int row0, int D, void* stream);

/* The same with the beta gradient ORDER-FIXED: every block parks its partial sum in partials [hoisdf_token_build_bwd_partials()]
 * and a one-wave launch adds them to *dbeta in block order - run-to-run identical, and without the cross-block float atomics whose
 * order noise reached 1e-2 of this heavily cancelling scalar (hoisdf_amd/ops.py and hoisdf_tokens_bwd use this form). */
int hoisdf_token_build_bwd_partials(void);
int hoisdf_token_build_bwd_ordered(const float* dtok, const float* feat, int ldfeat, const float* sdf, const float* beta_ptr,
                                   float* dfeat, int lddfeat, float* dbeta, float* partials, int B, int P, int S, int row0,
                                   int D, void* stream);

/* ---- K9/K10: attention -------------------------------------------------------------------
 * reference: nn.MultiheadAttention as used by common/nets/transformer.py:286-302,366-395.
 * q rows [B][Lq][.. ldq], k/v rows [B][Lk][.. ldk/ldv], head h occupies columns
 * [h*64, h*64+64) of each.  Only the first kv_len keys are attended (memory_mask of
 * common/utils/misc.py:34-47 keeps keys < num_samp_hand).  Streaming-softmax (never
 * materialises Lq x Lk), exact fp32 on the f32 MFMA pipe; q is scaled by 1/sqrt(64).
 * o [B][Lq][ldo]; lse [B][H][Lq] = log2-domain log-sum-exp of the scaled scores (opaque,
 * only to be handed back to hoisdf_attention_bwd).
 * Dropout on the probabilities: mask of (b,h,i,j) is a hash of (seed, (b*H+h)*Lq+i, j). */
int hoisdf_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                         float* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len,
                         float drop_p, uint64_t seed, void* stream);
/* delta [B][H][Lq] is scratch. dq/dk/dv have the same layouts/ld as q/k/v. Rows of dk/dv at
 * keys >= kv_len are written as zero. */
int hoisdf_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                         const float* o, int ldo, const float* dout, int lddo, const float* lse,
                         float* delta, float* dq, float* dk, float* dv, int B, int H, int Lq, int Lk,
                         int kv_len, float drop_p, uint64_t seed, void* stream);
/* Inference-only variant with f16 MFMA operands (BASELINE.json configs[4], "fp16 MFMA attention"): Q, K, V and the
 * probabilities are each split into f16 hi + lo parts (3 MFMA products per contraction, ~21 significant bits), both
 * products accumulate in f32 (v_mfma_f32_32x32x16_f16), softmax state in f32.  Same layouts and kv_len semantics as
 * hoisdf_attention_fwd; no dropout, no lse (not differentiable).  workspace: hoisdf_attention_f16_workspace(B, H, Lk)
 * bytes of device memory (f16 hi/lo copies of K and V^T), 16-byte aligned.  Error vs float64 attention < 1e-4 of
 * max|o| (tests); 1.8x the f32 kernel at 8192 keys.  The f32 entry point stays the parity configuration. */
long hoisdf_attention_f16_workspace(int B, int H, int Lk);
int hoisdf_attention_fwd_f16(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                             float* o, int ldo, int B, int H, int Lq, int Lk, int kv_len, void* workspace,
                             long workspace_bytes, void* stream);

/* The 16-bit-operand evaluation attention on the pipelined forward (round 5; what cfg.attention_f16_eval selects by default):
 * Q, K, V and the probabilities as bf16 hi + lo pairs (two planes per operand: 16 significant bits, the f32 exponent range - none of
 * the f16 scheme's power-of-two scaling, no overflow at trained sigma gates), three v_mfma_f32_32x32x16_bf16
 * products per product, f32 softmax state and accumulation.  Same layouts and kv_len semantics as hoisdf_attention_fwd; no dropout,
 * no lse.  reference: nn.MultiheadAttention forward inside the encoder layers (common/nets/transformer.py:269), BASELINE.json
 * configs[4].  workspace: hoisdf_attention_bf16x2_workspace(B, H, Lq, Lk) bytes, 16-byte aligned. */
long hoisdf_attention_bf16x2_workspace(int B, int H, int Lq, int Lk);
int hoisdf_attention_fwd_bf16x2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                int B, int H, int Lq, int Lk, int kv_len, void* workspace, long workspace_bytes, void* stream);

/* ---- fp32 attention emulated on the bf16 MFMA pipe ("bf16x3"; cfg.attention_emu, default) -------------------------------
 * reference: nn.MultiheadAttention inside the encoder layers (common/nets/transformer.py:269,286-302), forward + autograd
 * backward.  Same contracts, argument meaning, LSE convention (log2 domain) and dropout mask as hoisdf_attention_fwd / _bwd;
 * every contraction (QK^T, PV, dO V^T, dO^T P, Q^T dS, dS K) takes both f32 operands as exact bf16 triples and six bf16 MFMA
 * products per product with f32 accumulation (the arithmetic of hoisdf_linear_fwd_emu); softmax, dropout and the dS algebra stay
 * f32.  fp32-equivalent results (tests/test_gpu_emu.py), not bit-identical to the f32 entries.
 *   forward : workspace = hoisdf_attention_emu_workspace(B, H, Lq, Lk, keep ? 2 : 0) bytes, 16-byte aligned; keep = 1 leaves every
 *             Q / K / V plane the backward needs in it (pass it on as fwd_workspace).
 *   backward: ONE fused pass (dK, dV, dQ; 5 GEMM-equivalents), dQ through per-key-block partials summed in order - no atomics,
 *             run-to-run identical.  workspace = hoisdf_attention_bwd_emu_workspace(B, H, Lq, Lk, fwd_workspace != NULL) bytes.
 *             dq, dk, dv are fully overwritten (leading dimensions ldq, ldk, ldv of q, k, v). */
long hoisdf_attention_emu_workspace(int B, int H, int Lq, int Lk, int mode);
int hoisdf_attention_fwd_emu(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                             float* lse, int B, int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed, void* workspace,
                             long workspace_bytes, int keep, void* stream);
/* the forward in the f16x2 form (round 5; see "the two arithmetic forms" at hoisdf_linear_fwd_emu_mag): q_mag / k_mag / v_mag = the head
 * magnitudes of q, k and v (each pointer positioned at ITS operand's first head: for q, k, v that are column slices 0 / E / 2E of one
 * [q | k | v] matrix with words w = hoisdf_head_mag_words(B L, 3 H, L): w, w + H B, w + 2 H B) - one power-of-two scale per (sample,
 * head) and operand (round 6; round 5 had one per matrix: a sample's rounding depended on its batch companions); two f16 planes per
 * operand, three v_mfma_f32_32x32x16_f16 products per product, P carried as 2^6 P.  Same contracts, LSE convention and dropout mask
 * as hoisdf_attention_fwd_emu; drop_p < 0.75.  Accuracy: the operand pieces keep 22 bits, so a score s (log2 domain) carries an
 * absolute error of ~2^-22 |s| where the bf16x3 form has f32 rounding only - indistinguishable at the |s| <~ 100 of trained attention,
 * 4x the bf16x3 form's output error at |s| ~ 2000.  The planes it keeps (keep = 1) are f16x2 planes - only a backward of the same form
 * reads them.  o_mag (optional, zero-filled, B Lq words) receives the row magnitudes of o. */
int hoisdf_attention_fwd_emu_mag(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                 float* lse, int B, int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed, void* workspace,
                                 long workspace_bytes, int keep, const uint32_t* q_mag, const uint32_t* k_mag, const uint32_t* v_mag,
                                 uint32_t* o_mag, void* stream);
long hoisdf_attention_bwd_emu_workspace(int B, int H, int Lq, int Lk, int kept);
int hoisdf_attention_bwd_emu(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                             const float* dout, int lddo, const float* lse, float* delta, float* dq, float* dk, float* dv, int B,
                             int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed, const void* fwd_workspace,
                             void* workspace, long workspace_bytes, void* stream);
/* the backward in the f16x2 form (emu_attn_bwd4h_kernel, round 5): Q, K, V, dO and P as two scaled f16 pieces (three products per
 * product), dS as three (its magnitude follows P: five products in dQ and dK) - 76 instead of 120 MFMAs per query tile; one pass, no
 * atomics, run-to-run identical like hoisdf_attention_bwd_emu.  q_mag / k_mag / v_mag: as hoisdf_attention_fwd_emu_mag; do_mag: the
 * head magnitudes of dout (all required; hoisdf_head_mag_measure, or hoisdf_linear_bwd_input_emu_heads).  fwd_workspace: the planes a
 * forward OF THIS FORM kept, or NULL.  g_mag (optional, zero-filled, B L words; needs Lq == Lk): the row magnitudes of a [dq | dk | dv] matrix. */
int hoisdf_attention_bwd_emu_mag(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                                 const float* dout, int lddo, const float* lse, float* delta, float* dq, float* dk, float* dv, int B,
                                 int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed, const void* fwd_workspace,
                                 void* workspace, long workspace_bytes, const uint32_t* q_mag, const uint32_t* k_mag,
                                 const uint32_t* v_mag, const uint32_t* do_mag, uint32_t* g_mag, void* stream);
/* Small masked attention (17 MANO queries, tgt_mask of common/utils/misc.py:11-31):
 * mask [Lq][Lk] uint8, 1 = masked; Lq, Lk <= 64. probs [B][H][Lq][Lk] saved for backward. */
int hoisdf_attention_small_fwd(const float* q, int ldq, const float* k, int ldk, const float* v,
                               int ldv, const uint8_t* mask, float* o, int ldo, float* probs, int B,
                               int H, int Lq, int Lk, float drop_p, uint64_t seed, void* stream);
int hoisdf_attention_small_bwd(const float* q, int ldq, const float* k, int ldk, const float* v,
                               int ldv, const float* probs, const float* dout, int lddo, float* dq,
                               float* dk, float* dv, int B, int H, int Lq, int Lk, float drop_p,
                               uint64_t seed, void* stream);

/* ---- residual + dropout + LayerNorm (common/nets/transformer.py:290-301) -----------------
 * y = LN(x + dropout(r)) * gamma + beta over rows of width D (D <= 1024, multiple of 4);
 * r may be NULL (plain LN, e.g. encoder.inter_norm / decoder.norm).  mean/rstd [M] saved. */
int hoisdf_add_layernorm_fwd(const float* x, const float* r, const float* gamma, const float* beta,
                             float* y, float* mean, float* rstd, long M, int D, float eps,
                             float drop_p, uint64_t seed, void* stream);
/* dx (gradient w.r.t. x), dr (w.r.t. r, may be NULL), dgamma/dbeta accumulated atomically. */
int hoisdf_add_layernorm_bwd(const float* dy, const float* x, const float* r, const float* gamma,
                             const float* mean, const float* rstd,
                             const float* dx_add /* optional [M][D]: added into dx (another consumer's gradient) */,
                             float* dx, float* dr, float* dgamma, float* dbeta, long M, int D, float drop_p,
                             uint64_t seed, void* stream);

/* y = x + dropout(r) over rows of width D (multiple of 4) - the residual of a PRE-norm layer (reference
 * common/nets/transformer.py:304-331 TransformerEncoderLayer.forward_pre, :397-437 TransformerDecoderLayer.forward_pre; main/config.py:122
 * cfg.pre_norm).  x == NULL: y = dropout(r), which is also the backward (dr = dropout-mask(dy) with the SAME seed; dx = dy).
 * The mask is the (seed, row, column) function of hoisdf_add_layernorm_fwd.  Buffers 16-byte aligned; y may alias r. */
int hoisdf_residual_dropout(const float* x, const float* r, float* y, long M, int D, float drop_p, uint64_t seed, void* stream);

/* Plain LayerNorm of the FIRST `take` rows of every group of `rows_per_group` input rows (the encoder stack's inter_norm:
 * main/model.py:587-593 only ever reads the hand / object rows of each layer's normalised output).  x [groups *
 * rows_per_group][D]; y, mean, rstd compact [groups * take].  Backward: dy compact; dx covers ALL input rows - rows that
 * were normalised get their LayerNorm gradient (+ dx_add), the others dx_add (or 0 when dx_add is NULL). */
int hoisdf_layernorm_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                              long groups, int rows_per_group, int take, int D, float eps, void* stream);
int hoisdf_layernorm_rows_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                              const float* dx_add, float* dx, float* dgamma, float* dbeta, long groups, int rows_per_group,
                              int take, int D, void* stream);

/* ---- one transformer encoder layer per call (SURVEY.md section 8(b) coarse entries) ---------------------------------
 * reference: common/nets/transformer.py:286-302 (TransformerEncoderLayer.forward_post: self-attention, out-projection,
 * residual + dropout + LayerNorm, FFN linear1 -> ReLU -> dropout -> linear2, residual + dropout + LayerNorm) and the
 * stack's inter_norm of each layer output (:117-131).  Host-side chains of the per-op entries above on the caller's stream.
 * x [B][S][E] (dense).  Rows < n_query of every sample are produced: x_out [B][n_query][E]; with weights.g3 the
 * inter_norm of rows < n_inter goes to y_out [B][n_inter][E].  Keys / values always come from all S rows.
 * Arithmetic: linear layers follow hoisdf_set_gemm_emu (default: fp32 emulated on the bf16 pipe from 2048 rows up; the
 * img_* fields may carry hoisdf_linear_emu_prepare images of the weights (img_t_*: transpose = 1) - a NULL image is built
 * in the workspace on every call); attention as desc.attention says.  Split-precision / f16 modes are not offered here.
 * saved: hoisdf_encoder_layer_saved_bytes(desc) bytes the backward re-reads (training = 1); workspace:
 * hoisdf_encoder_layer_workspace_bytes(desc, 0 | 1) bytes of scratch (sized for the case that every image is built).
 * Backward: g_x_out / g_y = gradients of the two outputs (either may be NULL, not both); dx [B][S][E] is overwritten; the
 * parameter gradients must be zero on entry and hold the gradient on return. */
typedef struct hoisdf_encoder_layer_desc {
  int B, S, E, F, H;            /* batch, tokens per sample, model width (multiple of 4 and of H), FFN width, heads */
  int n_query;                  /* <= 0 or >= S: all rows */
  int n_inter;                  /* <= 0 or >= n_query: all produced rows */
  float eps, drop_p;
  uint64_t seed[4];             /* dropout streams: attention, after out-projection, FFN hidden, after the FFN */
  int attention;                /* 0: exact-f32 kernels; 2: emulated-fp32 forward (hoisdf_attention_fwd_emu) */
  int attention_bwd_emulated;   /* with attention == 2: the order-fixed emulated backward instead of the f32 fused one */
  int training;                 /* 0: nothing is saved (saved may be NULL), hoisdf_encoder_layer_bwd cannot follow */
  const uint32_t* x_mag;        /* optional: row magnitudes of x (B S words; f16x2 form, see hoisdf_linear_fwd_emu_mag; NULL = the layer
                                 * measures x itself), e.g. hoisdf_encoder_layer_out_mag of the layer below; must stay valid until the backward */
} hoisdf_encoder_layer_desc;
/* where a training forward left the row magnitudes of x_out (B n_query words) inside `saved` (NULL when the layer's contractions do
 * not run in the f16x2 form): the x_mag of the next layer's descriptor */
const uint32_t* hoisdf_encoder_layer_out_mag(const hoisdf_encoder_layer_desc* d, const void* saved);
typedef struct hoisdf_encoder_layer_weights {
  const float *w_in, *b_in;     /* [3E][E], [3E]: packed q | k | v in-projection */
  const float *w_out, *b_out;   /* [E][E], [E] */
  const float *g1, *be1;        /* norm1 */
  const float *w1, *b1;         /* [F][E], [F] */
  const float *w2, *b2;         /* [E][F], [E] */
  const float *g2, *be2;        /* norm2 */
  const float *g3, *be3;        /* inter_norm of the stack; NULL: no y_out */
  const void *img_in, *img_in_q, *img_in_kv, *img_out, *img_1, *img_2;              /* optional (see above); _q = rows [0, E), _kv = rows [E, 3E) */
  const void *img_t_in, *img_t_in_q, *img_t_in_kv, *img_t_out, *img_t_1, *img_t_2;
} hoisdf_encoder_layer_weights;
typedef struct hoisdf_encoder_layer_grads {
  float *dw_in, *db_in, *dw_out, *db_out, *dg1, *dbe1, *dw1, *db1, *dw2, *db2, *dg2, *dbe2, *dg3, *dbe3;
} hoisdf_encoder_layer_grads;
long hoisdf_encoder_layer_saved_bytes(const hoisdf_encoder_layer_desc* desc);
long hoisdf_encoder_layer_workspace_bytes(const hoisdf_encoder_layer_desc* desc, int backward_pass);
int hoisdf_encoder_layer_fwd(const float* x, const hoisdf_encoder_layer_weights* weights, const hoisdf_encoder_layer_desc* desc,
                             float* x_out, float* y_out, void* saved, long saved_bytes, void* workspace, long workspace_bytes,
                             void* stream);
int hoisdf_encoder_layer_bwd(const float* x, const float* x_out, const hoisdf_encoder_layer_weights* weights,
                             const hoisdf_encoder_layer_desc* desc, const void* saved, long saved_bytes, const float* g_x_out,
                             const float* g_y, float* dx, const hoisdf_encoder_layer_grads* grads, void* workspace,
                             long workspace_bytes, void* stream);

/* ---- one transformer decoder layer per call ---------------------------------------------------------------------------
 * reference: common/nets/transformer.py:366-395 (TransformerDecoderLayer.forward_post) as the hand stack runs it, and the
 * decoder stack's norm of every layer output (:150-163).  tgt [B][Q][E] (Q <= 64 MANO queries), memory [B][S][E] (the encoder
 * output; only keys < kv_len are attended = the memory mask of common/utils/misc.py:34-47), query_pos [Q][E] (broadcast over
 * the batch, added to the queries / keys of the self-attention and the queries of the cross-attention), tgt_mask [Q][Q]
 * uint8 (1 = masked).  out [B][Q][E]; with weights.g4, y_out = the stack norm of out.  Buffers and arithmetic as for the
 * encoder layer (the 17-query attention kernels are exact f32; the memory's K / V projection follows hoisdf_set_gemm_emu).
 * Backward: d_tgt is overwritten; d_memory is overwritten or (accumulate_memory = 1: the memory feeds every decoder layer)
 * added to; d_query_pos [Q][E] (may be NULL) is ADDED to; parameter gradients zero on entry. */
typedef struct hoisdf_decoder_layer_desc {
  int B, Q, S, E, F, H, kv_len;
  float eps, drop_p;
  uint64_t seed[6];             /* dropout streams: self-attention, after its out-projection, cross-attention, after its
                                   out-projection, FFN hidden, after the FFN */
  int training;
} hoisdf_decoder_layer_desc;
typedef struct hoisdf_decoder_layer_weights {
  const float *sa_w_in, *sa_b_in, *sa_w_out, *sa_b_out;      /* self_attn: [3E][E], [3E], [E][E], [E] */
  const float *ca_w_in, *ca_b_in, *ca_w_out, *ca_b_out;      /* multihead_attn */
  const float *w1, *b1, *w2, *b2;                            /* [F][E], [F], [E][F], [E] */
  const float *g1, *be1, *g2, *be2, *g3, *be3;               /* norm1..3 */
  const float *g4, *be4;                                     /* the stack's norm; NULL: no y_out */
  const void *img_ca_kv, *img_t_ca_kv;                       /* optional bf16x3 images of ca_w_in rows [E, 3E) (and transposed) */
} hoisdf_decoder_layer_weights;
typedef struct hoisdf_decoder_layer_grads {
  float *dsa_w_in, *dsa_b_in, *dsa_w_out, *dsa_b_out, *dca_w_in, *dca_b_in, *dca_w_out, *dca_b_out, *dw1, *db1, *dw2, *db2,
      *dg1, *dbe1, *dg2, *dbe2, *dg3, *dbe3, *dg4, *dbe4;
} hoisdf_decoder_layer_grads;
long hoisdf_decoder_layer_saved_bytes(const hoisdf_decoder_layer_desc* desc);
long hoisdf_decoder_layer_workspace_bytes(const hoisdf_decoder_layer_desc* desc, int backward_pass);
int hoisdf_decoder_layer_fwd(const float* tgt, const float* memory, const float* query_pos, const uint8_t* tgt_mask,
                             const hoisdf_decoder_layer_weights* weights, const hoisdf_decoder_layer_desc* desc, float* out,
                             float* y_out, void* saved, long saved_bytes, void* workspace, long workspace_bytes, void* stream);
int hoisdf_decoder_layer_bwd(const float* tgt, const float* memory, const uint8_t* tgt_mask, const float* out,
                             const hoisdf_decoder_layer_weights* weights, const hoisdf_decoder_layer_desc* desc, const void* saved,
                             long saved_bytes, const float* g_out, const float* g_y, float* d_tgt, float* d_memory,
                             int accumulate_memory, float* d_query_pos, const hoisdf_decoder_layer_grads* grads, void* workspace,
                             long workspace_bytes, void* stream);

/* ---- K12: vote aggregation ----------------------------------------------------------------
 * reference: common/nets/loss.py:31-56.  off [L][B][P][J*3], cls [L][B][P][J] (batch-first
 * rows), pts [B][P][3].  joints[l][b][j] = sum_p softmax_p(cls)[p] * (pts[p] + off[p][j]).
 * stats [L][B][J][2] = (max, sum of exp) of each softmax column, saved for the backward. */
int hoisdf_vote_fwd(const float* off, const float* cls, const float* pts, float* joints, float* stats,
                    int L, int B, int P, int J, void* stream);
int hoisdf_vote_bwd(const float* off, const float* cls, const float* pts, const float* joints,
                    const float* stats, const float* djoints, float* doff, float* dcls, int L, int B,
                    int P, int J, void* stream);

/* K12 with the three JointvoteLoss reductions fused (common/nets/loss.py:31-56): besides joints / stats
 * (as hoisdf_vote_fwd) one pass writes, per (l, b):
 *   l3d_sum [L][B] = sum_{p,j,d} smooth_l1(1000*(pts+off) - gt_mm) * near,   near = ||pts - gt_mm/1000|| < radius
 *   bce_sum [L][B] = sum_{p,j} binary_cross_entropy_with_logits(cls, near);  near_sum [B] = sum_{p,j} near.
 * The reference's scalars are loss_joint_3d = mean_l(sum_b l3d_sum / sum_b near_sum) / 3,
 * loss_joint_cls = sum(bce_sum) / (L*B*P*J).  The backward takes d(loss)/d(l3d_sum), d/d(bce_sum) [L][B]
 * and d/d(joints) [L][B][J][3] (any of them may be NULL = zero). */
int hoisdf_vote_loss_fwd(const float* off, const float* cls, const float* pts, const float* joint_gt_mm,
                         float radius, float* joints, float* stats, float* l3d_sum, float* bce_sum,
                         float* near_sum, int L, int B, int P, int J, void* stream);
int hoisdf_vote_loss_bwd(const float* off, const float* cls, const float* pts, const float* joint_gt_mm,
                         float radius, const float* joints, const float* stats, const float* djoints,
                         const float* dl3d_sum, const float* dbce_sum, float* doff, float* dcls, int L,
                         int B, int P, int J, void* stream);

/* ---- K1 + K7 + K8 and K11 + K12 as single calls (SURVEY.md section 8(b) `hoisdf_token_build_*`, `hoisdf_heads_vote_*`) ----------
 * An MLP = Linear (+ ReLU) chain as common/nets/layer.py:168-201 builds it: layer i maps dims[i] -> dims[i + 1] with weight
 * w[i] [dims[i + 1]][dims[i]] (dense) and bias b[i]; ReLU after every layer but the last, and after the last one too when
 * act_last (linear_transformerin, main/model.py:58-62).  Gradient buffers are zero on entry and hold the gradient on return.
 *
 * hoisdf_tokens_fwd: the token rows of one point set (reference: Model.get_input_transformer, main/model.py:145-179, + the
 *   sigma gate and concatenation, :123-126,520-562): feat [B P][C] = the gathered pyramid rows (feat_in with their camera points
 *   cam_in [B P][3]; or feat_in = NULL: projected + gathered here from pyr / points [B][P][3] / cam_intr, cam_out [B P][3]
 *   receives the camera points) -> MLP -> fea [B P][D - 33] -> tok[b][row0 + p][:] = [cam - center | pe | fea * sigmoid(sdf /
 *   beta) / beta] (hoisdf_token_build_fwd).  fea_out (optional) receives the MLP output for other consumers (the detached
 *   cross-field tokens, :540,:558).  hoisdf_tokens_bwd: dtok [B][S][D] -> the MLP's weight gradients, *dbeta += ..., and the
 *   gradient of the gathered rows into dfeat [B P][C] (overwritten or, accumulate_dfeat = 1, added to) - or, when the rows were
 *   gathered inside and dpyr is given, scatter-added into the pyramid gradient.
 * hoisdf_heads_vote_fwd: enc [L][B][P][E] = the intermediate hand rows of all L encoder depths (main/model.py:587-593) ->
 *   linear_handvote / linear_handcls -> hoisdf_vote_loss_fwd: joints [L][B][J][3] and the three reductions of
 *   JointvoteLoss (see hoisdf_vote_loss_fwd).  hoisdf_heads_vote_bwd: gradients of joints / l3d_sum / bce_sum (any may be NULL)
 *   -> both MLPs' weight gradients and denc [L B P][E] (overwritten). */
#define HOISDF_MLP_MAX_LAYERS 4
typedef struct hoisdf_mlp {
  int n_layers, act_last;
  int dims[HOISDF_MLP_MAX_LAYERS + 1];
  const float* w[HOISDF_MLP_MAX_LAYERS];
  const float* b[HOISDF_MLP_MAX_LAYERS];
  /* optional (NULL = built per call in the workspace): the bf16x3 slab images of w[i] (hoisdf_linear_emu_prepare, transpose 0 for
   * the forward, transpose 1 for the grad-input GEMM) when the caller keeps them cached across calls - a training step otherwise
   * rebuilds ~30 of them inside these entries */
  const void* img[HOISDF_MLP_MAX_LAYERS];
  const void* img_t[HOISDF_MLP_MAX_LAYERS];
} hoisdf_mlp;
typedef struct hoisdf_mlp_grads {
  float* dw[HOISDF_MLP_MAX_LAYERS];
  float* db[HOISDF_MLP_MAX_LAYERS];
} hoisdf_mlp_grads;
long hoisdf_tokens_saved_bytes(const hoisdf_mlp* mlp, long n_rows, int gather_inside);
long hoisdf_tokens_workspace_bytes(const hoisdf_mlp* mlp, long n_rows, int backward_pass);
int hoisdf_tokens_fwd(const hoisdf_pyramid* pyr, const float* points, const float* center, const float* cam_intr, float scale,
                      int img_h, int img_w, const float* feat_in, const float* cam_in, const hoisdf_mlp* mlp, const float* pe,
                      const float* sdf, const float* beta_ptr, float* tok, float* fea_out, float* cam_out, int B, int P, int S,
                      int row0, int D, void* saved, long saved_bytes, void* workspace, long workspace_bytes, void* stream);
int hoisdf_tokens_bwd(const hoisdf_pyramid_grad* dpyr, const float* points, const float* center, const float* cam_intr,
                      float scale, int img_h, int img_w, const float* feat_in, const hoisdf_mlp* mlp, const float* sdf,
                      const float* beta_ptr, const float* dtok, const void* saved, long saved_bytes,
                      const hoisdf_mlp_grads* grads, float* dfeat, int accumulate_dfeat, float* dbeta, int B, int P, int S,
                      int row0, int D, void* workspace, long workspace_bytes, void* stream);
long hoisdf_heads_vote_saved_bytes(const hoisdf_mlp* vote, const hoisdf_mlp* cls, int L, int B, int P, int J);
long hoisdf_heads_vote_workspace_bytes(const hoisdf_mlp* vote, const hoisdf_mlp* cls, int L, int B, int P, int J,
                                       int backward_pass);
int hoisdf_heads_vote_fwd(const float* enc, const hoisdf_mlp* vote, const hoisdf_mlp* cls, const float* pts,
                          const float* joint_gt_mm, float radius, float* joints, float* l3d_sum, float* bce_sum,
                          float* near_sum, int L, int B, int P, int J, void* saved, long saved_bytes, void* workspace,
                          long workspace_bytes, void* stream);
int hoisdf_heads_vote_bwd(const float* enc, const hoisdf_mlp* vote, const hoisdf_mlp* cls, const float* pts,
                          const float* joint_gt_mm, float radius, const float* joints, const void* saved, long saved_bytes,
                          const float* djoints, const float* dl3d_sum, const float* dbce_sum,
                          const hoisdf_mlp_grads* vote_grads, const hoisdf_mlp_grads* cls_grads, float* denc, int L, int B,
                          int P, int J, void* workspace, long workspace_bytes, void* stream);

/* ---- a15: the scalar point losses ------------------------------------------------
 * reference: common/nets/loss.py:64-78 (SepSDFLoss = torch.nn.L1Loss(mean) of the clamped SDF predictions against the
 * ground truth clamped as main/model.py:393-400 does), main/model.py:35-36,656-662 (obj_rot / obj_trans:
 * torch.nn.SmoothL1Loss, beta 1, mean over (L, B, P, 3) against the (B, 3) target expanded over depth and points),
 * common/nets/loss.py:57-59 (loss_all_joint_3d: SmoothL1 of joints * 1000 against the ground truth expanded over depth).
 * Element i of pred [n] is compared with target[((i / (rep * C)) % Bt) * C + i % C] - a (Bt, C) target broadcast over
 * leading dimensions and over `rep` repeats between Bt and C.  kind 0: |pred * pred_scale - clamp(target, +-clamp)|
 * (clamp <= 0: the target is taken as is); kind 1: SmoothL1 with beta = 1 of the same difference.
 * fwd: loss[0] = out_scale * sum_i (out_scale = 1 / n gives the reference's mean); order-fixed (per-block partial sums in
 *   partials [hoisdf_point_loss_blocks(n)] added in block order): bit-reproducible.
 * bwd: dpred[i] = g[0] * out_scale * pred_scale * dl/dd (torch's subgradients: sign(0) = 0). */
int hoisdf_point_loss_blocks(long n);
int hoisdf_point_loss_fwd(const float* pred, const float* target, long n, long rep, int C, long Bt, int kind,
                          float clamp, float pred_scale, float out_scale, float* partials, float* loss, void* stream);
int hoisdf_point_loss_bwd(const float* pred, const float* target, long n, long rep, int C, long Bt, int kind,
                          float clamp, float pred_scale, float out_scale, const float* g, float* dpred, void* stream);

/* ---- (f3) MANO head ---------------------------------------------------------------------------------------------
 * reference: common/nets/mano_head.py:12-278 (6D -> rotation -> quaternion -> axis-angle, the head), manopth/manopth/
 * manolayer.py:111-276 (the layer as main/model.py:735-742 configures it: use_pca=False, flat_hand_mean=True,
 * center_idx=0, side right), common/nets/loss.py:81-171 (ManoLoss: four MSE terms).  One workgroup per hand.
 *
 * hoisdf_mano_prepare: the layer's blend-shape tables th_shapedirs [778][3][10] and th_posedirs [778][3][135] transposed
 *   into one image [145][2334], followed by th_weights [778][16] transposed (hoisdf_mano_dirs_image_floats() floats in
 *   all) - build it once per set of assets.
 * hoisdf_mano_head_fwd, for `hands` hands:
 *   mode 0 (predictions): pose = 6D rotations [hands][16][6] (row stride ldpose >= 96), betas [hands][>= 10];
 *   mode 1 (ground truth): pose = axis-angle MANO coefficients [hands][>= 48] as the dataset stores them (mano_param[:, :48]:
 *     the head subtracts hands_mean from [3:48] and the layer adds it back; rot = Rodrigues of the mean-free coefficients);
 *   v_template [778][3], j_regressor [16][778], weights [778][16] (16-byte aligned), hands_mean [45];
 *   outputs in metres, centred on the wrist: verts [hands][778][3], joints [hands][21][3] (the reference's 21-joint order),
 *   rot [hands][16][3][3] (mode 0: the Gram-Schmidt rotations = pred mano_pose).
 *   With gt_verts != NULL (mode 0) hand h is compared with ground-truth hand h % gt_hands and
 *   loss_sums [hands][4] = sum of squared errors of (verts, joints, rot, betas) - the reference's four ManoLoss terms are
 *   lambda_i * sum_h loss_sums[h][i] / (hands * {2334, 63, 144, 10}).
 * hoisdf_mano_head_bwd (mode 0 inputs again; nothing is saved between the calls): upstream gradients g_loss_sums
 *   [hands][4], g_verts, g_joints, g_rot (each may be NULL = zero) -> d_pose6d [hands][16][6], d_betas [hands][10].
 *   hands_mean must be zero (flat_hand_mean=True): the backward applies the layer's rotation gradient to the 6D rotation
 *   directly, which is exact only then (csrc/mano.hip header). */
long hoisdf_mano_dirs_image_floats(void);
int hoisdf_mano_prepare(const float* shapedirs, const float* posedirs, const float* weights, float* image, void* stream);
int hoisdf_mano_head_fwd(const float* pose, int ldpose, int mode, const float* betas, int ldbetas, int hands,
                         const float* dirs_image, const float* v_template, const float* j_regressor, const float* weights,
                         const float* hands_mean, const float* gt_verts, const float* gt_joints, const float* gt_rot,
                         const float* gt_shape, int ldgt_shape, int gt_hands, float* verts, float* joints, float* rot,
                         float* loss_sums, void* stream);
int hoisdf_mano_head_bwd(const float* pose6d, const float* betas, int hands, const float* dirs_image, const float* v_template,
                         const float* j_regressor, const float* weights, const float* hands_mean, const float* gt_verts,
                         const float* gt_joints, const float* gt_rot, const float* gt_shape, int ldgt_shape, int gt_hands,
                         const float* g_loss_sums, const float* g_verts, const float* g_joints, const float* g_rot,
                         float* d_pose6d, float* d_betas, void* stream);

/* ---- (f4) auxiliary image losses of the encoder outputs ---------------------------------------------------
 * reference: main/model.py:128-143 (render_gaussian_heatmap) and :404-422 (MSELoss / BCELoss with reduction none).
 * dec = decoder_out (B, 3, H, W) [heat-map, hand seg, object seg] with element strides (sb, sc, sh, sw) - NCHW or
 * channels_last; joints (B, J, 2) = (x, y) in heat-map pixels; segs (B, H, W) in [0, 1].
 * heatmap (B, H, W) = 255 * sum_j exp(-((x - jx)/sigma)^2/2 - ((y - jy)/sigma)^2/2) (kept for the backward);
 * loss_heatmap = (dec0 - heatmap)^2, loss_*_seg = binary cross entropy with the log terms clamped at -100.
 * bwd: ddec gets all three channels (same strides as dec); a NULL upstream gradient means zero. */
int hoisdf_aux_image_losses_fwd(const float* dec, long sb, long sc, long sh, long sw, const float* joints,
                                const float* hand_seg, const float* obj_seg, int B, int J, int H, int W, float sigma,
                                float* heatmap, float* loss_heatmap, float* loss_obj_seg, float* loss_hand_seg,
                                void* stream);
int hoisdf_aux_image_losses_bwd(const float* dec, long sb, long sc, long sh, long sw, const float* hand_seg,
                                const float* obj_seg, const float* heatmap, const float* g_heatmap,
                                const float* g_obj_seg, const float* g_hand_seg, int B, int H, int W, float* ddec,
                                void* stream);

/* ---- (f4) BatchNorm2d (+ residual add) (+ ReLU) of a channels_last encoder map -------------------------------------
 * reference: the BatchNorm2d -> ReLU pairs and the `out += identity; relu(out)` block tails of common/nets/resnet.py (the
 * torchvision ResNet blocks it wraps) and common/nets/layer.py:23-63 (Conv / ConvTranspose -> BatchNorm2d -> ReLU), i.e.
 * torch.nn.functional.batch_norm(training) [+ add] + relu and its autograd backward.  The convolutions stay MIOpen's.
 * A channels_last (N, C, H, W) map is the row-major matrix [M = N H W][C] (row stride ld >= C, ld % 4 == 0; C % 8 == 0,
 * C <= 2048).  hoisdf_bn_stats: batch mean and 1 / sqrt(biased variance + eps) per channel (shifted f32 sums per block, f64
 * combination in block order by the last block: bit-reproducible), running statistics updated as torch does (unbiased
 * variance, momentum; NULL = not tracked).  hoisdf_bn_apply_fwd: y [M][C] dense = relu?((x - mean) gamma invstd + beta
 * (+ residual)); second_is_variance = 1: the fourth argument holds variances (evaluation mode: the running statistics);
 * sign_bits [M][C / 8] bytes (bit j of byte (row, c / 8) = y[row][c + j] > 0; NULL = not wanted).  hoisdf_bn_bwd: dx [M][C]
 * dense, d_residual (NULL = no residual) = dy masked by the sign bits (NULL = no ReLU), dgamma / dbeta [C] overwritten.
 * workspace: hoisdf_bn_workspace_floats(M, C) floats. */
long hoisdf_bn_workspace_floats(long M, int C);
int hoisdf_bn_stats(const float* x, long ldx, long M, int C, float* mean, float* invstd, float* running_mean,
                    float* running_var, float momentum, float eps, float* workspace, long workspace_floats,
                    void* stream);
int hoisdf_bn_apply_fwd(const float* x, long ldx, const float* residual, long ldr, const float* mean,
                        const float* invstd_or_var, int second_is_variance, float eps, const float* gamma,
                        const float* beta, int relu, float* y, uint8_t* sign_bits, long M, int C, void* stream);
int hoisdf_bn_bwd(const float* dy, long lddy, const float* x, long ldx, const uint8_t* sign_bits, const float* mean,
                  const float* invstd, const float* gamma, float* dx, float* d_residual, float* dgamma, float* dbeta,
                  long M, int C, float* workspace, long workspace_floats, void* stream);

/* ---- optimizer step of the training loop ------------------------------------------------------------------
 * reference: torch.optim.AdamW(model.parameters(), lr=cfg.lr) (common/base.py:64-73; betas (0.9, 0.999), eps 1e-8,
 * weight_decay 1e-2 = torch defaults), stepped once per iteration (main/train.py:139).
 * chunks: device array; chunk i updates n <= 16384 consecutive elements of one parameter in place
 * (param, exp_avg, exp_avg_sq) from grad * grad_scale; step = 1-based count of this update (bias correction).
 * grad_scale = 1/world_size folds the gradient averaging of the all-reduce into the same pass.  Hyper-parameters are
 * doubles: 1 - beta2 formed from a float 0.999 is already 1.3e-5 off. */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;
} hoisdf_adamw_chunk;
int hoisdf_adamw_step(const hoisdf_adamw_chunk* chunks, int n_chunks, double lr, double beta1, double beta2,
                      double eps, double weight_decay, long step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HOISDF_H_ */
