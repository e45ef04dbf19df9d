// Shared device/host helpers for the HOISDF gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hoisdf.h"

namespace hoisdf {

// ---- error plumbing (thread-local message, int codes; never throws) -------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define HOISDF_REQUIRE(cond, code, ...)                 \
  do {                                                  \
    if (!(cond)) {                                      \
      ::hoisdf::set_error(__VA_ARGS__);                 \
      return (code);                                    \
    }                                                   \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- counter-based RNG for dropout ----------------------------------------------------
// keep(row, col) is a pure function of (seed, row, col) so the backward pass regenerates the mask:
//     x = mix(seed) + row * G1 + col * G2          (two Weyl sequences, odd 32-bit constants)
//     x ^= x >> 15;  x *= 0x2C1B3C6D;  x ^= x >> 12   (one multiply-xorshift round)
//     keep = x >= p * 2^32                          (integer compare, no float conversion)
// On gfx950 a 32-bit integer multiply is a quarter-rate instruction.  Wherever row or col is "tile base + compile-time
// offset" - the forward holds a query row per lane and walks keys in registers, the backward holds a key per lane and
// walks queries - the Weyl products fold into adds, leaving ONE multiply per element (the previous two-round finaliser
// over (col * C) ^ rowkey cost three).  Mask statistics (drop rate, row / column marginals, lag-1/32/64 and diagonal
// correlations over 4096 x 2048 elements) are indistinguishable from the two-round hash.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// row index modulo 2^32 (every caller's row space - B*H*Lq, GEMM / LayerNorm rows - is below 2^31): a 32-bit argument
// lets "row = base + constant" fold into one add per element
__device__ __forceinline__ uint32_t drop_rowkey(uint64_t seed, uint32_t row) {
  return mix32((uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0xC2B2AE3DU)) + row * 0x85EBCA77U;
}
// Round 4: ONE hash serves the two columns 2 c, 2 c + 1 of a row - its high 16 bits decide the odd column, its low 16 bits the even
// one (threshold at 16-bit granularity: p = 0.1 drops 6553 / 65536 = 0.09999).  The emulated attention forward, where the
// hash was the largest single VALU item of a key tile (16 scores per lane = 8 column pairs), computes 8 hashes instead of 16;
// everywhere else the function is evaluated per element as before (one extra shift).  Every kernel goes through these two
// helpers, so forward and backward masks stay the same function of (seed, row, col).
__device__ __forceinline__ uint32_t drop_hash(uint32_t rowkey, uint32_t colpair) {
  uint32_t x = rowkey + colpair * 0x9E3779B9U;
  x ^= x >> 15; x *= 0x2C1B3C6DU; x ^= x >> 12;
  return x;
}
__device__ __forceinline__ bool drop_keep(uint32_t rowkey, uint32_t col, uint32_t thresh) {
  const uint32_t h = drop_hash(rowkey, col >> 1);
  return ((col & 1u) ? h : (h << 16)) >= (thresh & 0xffff0000u);
}
// multiplier applied to a kept element; 0 for a dropped one.
__device__ __forceinline__ float drop_scale(uint32_t rowkey, uint32_t col, uint32_t thresh, float inv_keep) {
  return drop_keep(rowkey, col, thresh) ? inv_keep : 0.f;
}
static inline uint32_t drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

// ---- deterministic mode (HOISDF_DETERMINISTIC=1 / hoisdf_set_deterministic) -------------------------------
// Every kernel that accumulates with float atomics has an order-fixed alternative, selected per launch:
//   split-k grad-weight -> partial tiles + ordered reduce (caller workspace); small split-k forward / grad-input -> off;
//   attention dQ -> the two-kernel form; 17-query attention dK/dV -> one wave per (sample, head) in query order;
//   gather backward -> pixel-tile owners walking the points in order (gather.hip); per-block column sums of
//   LayerNorm / SDF head / sigma-gate backward -> parked per block and summed in block order by the last block
//   to arrive (block_column_sum below, library-owned scratch: deterministic mode is single-stream).
// ---- in-projection -> attention planes (round 4): the emulated forward GEMM of an attention in-projection can write its output
// tile straight into the bf16x3 planes the emulated attention kernels read (attention_emu.hip) instead of an f32 [M][N] matrix
// that a conversion pass would re-read: row planes [b H + head][Lp][64] and, for the value part, the transposed planes
// [b H + head][64][Lp].  The GEMM's columns are columns col0 .. col0 + N - 1 of the [q | k | v] projection (each part E wide,
// heads of 64), its rows are (b, s) = (row / L, row % L).  Requires L % 128 == 0 and N % 64 == 0 (whole wave tiles).
struct QkvPlanes {
  void* r[3][3];     // [part q / k / v][piece] row planes, null: that part's rows are not wanted
  void* vt[3];       // transposed value planes, null: not wanted
  int on, L, Lp, H, E, col0;
  float qscale;      // the query part is scaled before the split (softmax scale in the log2 domain)
};
int linear_fwd_emu_qkv(const float* x, int ldx, const void* w_image, const float* bias, long M, int N, int K, const QkvPlanes& pl,
                       void* stream, const uint32_t* x_mag = nullptr);
// attention_emu.hip: where the planes of a forward workspace live (hoisdf_attention_emu_workspace(.., keep ? 2 : 0) bytes), and the
// forward / backward over planes that are already there
void attention_emu_plane_targets(void* workspace, int B, int H, int Lq, int Lk, int keep, QkvPlanes& q_part, QkvPlanes& kv_part);
int attention_fwd_emu_planes(float* o, int ldo, float* lse, int B, int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed,
                             void* workspace, int keep, void* stream, uint32_t* o_mag = nullptr);
// hoisdf_attention_fwd_emu / _bwd_emu with row magnitudes (below) for o / for [dq | dk | dv] together, and - the f16x2 form - the head
// magnitudes (below) of q, k, v (and dout): all of them or none
int attention_fwd_emu_mag(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, float* lse, int B,
                          int H, int Lq, int Lk, int kv_len, float drop_p, uint64_t seed, void* workspace, long workspace_bytes, int keep,
                          uint32_t* o_mag, void* stream, const uint32_t* q_hm = nullptr, const uint32_t* k_hm = nullptr,
                          const uint32_t* v_hm = nullptr);
int attention_bwd_emu_mag(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                          const float* dout, int lddo, const float* lse, float* delta, float* dq, float* dk, float* dv, int B, int H,
                          int Lq, int Lk, int kv_len, float drop_p, uint64_t seed, const void* fwd_workspace, void* workspace,
                          long workspace_bytes, uint32_t* g_mag, void* stream, const uint32_t* q_hm = nullptr,
                          const uint32_t* k_hm = nullptr, const uint32_t* v_hm = nullptr, const uint32_t* do_hm = nullptr);

// gemm_emu.hip, f16x2 form: is it on (HOISDF_EMU_FORM, process-wide); row magnitudes (below) of a row-major f32 matrix measured by
// the library: M words, plain stores (no zeroing needed); head magnitudes of an attention operand (groups = columns / 64, L rows per sample)
bool emu_form_h2();
int emu_rowmag_launch(const float* x, long ld, long M, int K, uint32_t* words, hipStream_t st);
int emu_mag_measure(const float* x, long ld, long M, int K, uint32_t* words, hipStream_t st);      // the same, traced as "a chain, once for all its readers"
int emu_headmag_launch(const float* x, long ld, long M, int groups, int L, uint32_t* words, hipStream_t st);   // zeroes, then folds
uint32_t* mag_scratch(hipStream_t st, long words);      // stream-ordered scratch for words the library measures itself (per device and stream)

// ---- row magnitudes (f16x2 form; round 6: one word per ROW instead of 256 words per matrix).  A kernel that writes a matrix which
// a later contraction reads as its row operand leaves each row's largest magnitude behind while it still holds the values: one u32
// per row, the IEEE bits of max |x[row][:]| (|x| as bits orders like |x|), ZERO before the producer(s) run, folded with an unsigned
// atomic max - order-independent, hence deterministic; several producers may share an array (the column tiles of a GEMM, the heads
// of an attention output, dQ / dK / dV of one [dq | dk | dv] matrix).  The forward / grad-input contraction scales EVERY ROW by its
// own power of two (the scale leaves again in the epilogue, per output row): a row's rounding depends on that row alone - samples of
// a batch do not see each other, and a row far below the matrix maximum keeps its 22 bits.  The grad-weight contracts OVER the rows:
// each of its row slices takes the largest word of its own rows.  An operand without words gets hoisdf's own pass (emu_rowmag_launch:
// one more read of the matrix).  A word smaller than the row's true maximum (a stale array) cannot produce Inf / NaN: the kernels that
// split run with MODE.FP16_OVFL set, so an f16 piece saturates at +-65504 (4 x head room above the promised [2^13, 2^14), then clipping).
__device__ __forceinline__ uint32_t mag_bits(float v) { return __builtin_bit_cast(uint32_t, v) & 0x7fffffffu; }
__device__ __forceinline__ uint32_t mag_bits4(const float4& v) {
  return max(max(mag_bits(v.x), mag_bits(v.y)), max(mag_bits(v.z), mag_bits(v.w)));
}
// f16 conversions of this wave saturate instead of overflowing to Inf (MODE.FP16_OVFL, hwreg(HW_REG_MODE, 23, 1)): once, at kernel entry
__device__ __forceinline__ void f16_saturate_on() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }
// all 64 lanes of a wave hold (part of) ONE row: fold the wave's maximum into that row's word; words may be null
__device__ __forceinline__ void rowmag_publish_wave(uint32_t* words, long row, uint32_t m);
// groups of G consecutive lanes (G = 4, 8, 16, 32, 64) each hold (part of) one row: every group's maximum ends in all of its lanes.
// Within a 16-lane row the exchange is DPP (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: no LDS traffic).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
template <int G>
__device__ __forceinline__ uint32_t group_max_u32(uint32_t m) {
  static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group size");
  m = max(m, dpp_u32<0xB1>(m));
  m = max(m, dpp_u32<0x4E>(m));
  if (G >= 8) m = max(m, dpp_u32<0x141>(m));
  if (G >= 16) m = max(m, dpp_u32<0x140>(m));
  if (G >= 32) m = max(m, (uint32_t)__shfl_xor((int)m, 16, 64));
  if (G >= 64) m = max(m, (uint32_t)__shfl_xor((int)m, 32, 64));
  return m;
}
__device__ __forceinline__ void rowmag_publish_wave(uint32_t* words, long row, uint32_t m) {
  if (!words) return;
  m = group_max_u32<64>(m);
  if ((threadIdx.x & 63) == 0) atomicMax(words + row, m);
}
// power-of-two operand scale from the largest magnitude (bits of |x|max): max |x| s in [2^13, 2^14); zero / denormal / huge maxima clamp
__device__ __forceinline__ uint32_t mag_exp(uint32_t amax_bits) { return min(max((amax_bits >> 23) & 0xffu, 14u), 254u); }
__device__ __forceinline__ float mag_scale(uint32_t amax_bits) { return __builtin_bit_cast(float, (267u - mag_exp(amax_bits)) << 23); }
__device__ __forceinline__ float mag_inv_scale(uint32_t amax_bits) { return __builtin_bit_cast(float, (mag_exp(amax_bits) - 13u) << 23); }
__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, uint32_t* red4) {      // 256 threads; red4 = four shared words
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return max(max(red4[0], red4[1]), max(red4[2], red4[3]));
}
// ---- head magnitudes: the operands of the emulated attention (Q, K, V, dO: column slices of 64 per head, L rows per sample) take ONE
// scale per (sample, head) - the softmax of a (sample, head) never sees another one's scale.  Layout of the words of a matrix with nc
// 64-column groups and nb samples: word[group * nb + sample], zero before the producer runs (the GEMM whose epilogue writes the matrix
// folds each wave tile's maximum into the word(s) of the samples its rows belong to; hoisdf_head_mag_measure for a matrix nobody described).
__device__ __forceinline__ long head_words(long groups, long nb) { return groups * nb; }
// the entries below with magnitude words (null = none): x_mag / dy_mag describe the row operand, y_mag / dx_mag receive the output's
// (y_heads / dx_heads: head magnitudes of the output, samples of head_L rows; null = not wanted)
int linear_fwd_emu_mag(const float* x, int ldx, const void* w_image, const float* bias, float* y, int ldy, long M, int N, int K, int act,
                       float drop_p, uint64_t seed, uint32_t* relu_bits, const uint32_t* x_mag, uint32_t* y_mag, void* stream,
                       uint32_t* y_heads = nullptr, int head_L = 0);
int linear_bwd_input_emu_mag(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const void* wt_image, float* dx, int lddx,
                             long M, int N, int K, int accumulate, const uint32_t* dy_mag, uint32_t* dx_mag, void* stream,
                             uint32_t* dx_heads = nullptr, int head_L = 0);
int linear_bwd_weight_emu_mag(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x, int ldx, float* dW, int lddw,
                              float* db, long M, int N, int K, float* workspace, long workspace_floats, const uint32_t* dy_mag,
                              const uint32_t* x_mag, void* stream);
int project_gather_fwd_mag(const hoisdf_pyramid* pyr, const float* points, const int32_t* sample_idx, long n_rows, int rows_per_sample,
                           const float* center, const float* cam_intr, float scale, int img_h, int img_w, float* feat, int ldf, float* cam_out,
                           float* uv_out, uint32_t* feat_mag, void* stream);
int posenc_fwd_mag(const float* points, long n_rows, float* x0, int ldx0, int col0, float* pe, uint32_t* x0_mag, void* stream);
int sdf_head_bwd_mag(const float* dsdf, const float* sdf_raw, const float* h, int ldh, const float* w, float* dh, int lddh, float* dw,
                     float* db, long n_rows, int K, float clamp, uint32_t* dh_mag, void* stream);
int add_layernorm_fwd_mag(const float* x, const float* r, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long M,
                          int D, float eps, float drop_p, uint64_t seed, uint32_t* y_mag, void* stream);
int add_layernorm_bwd_mag(const float* dy, const float* x, const float* r, const float* gamma, const float* mean, const float* rstd,
                          const float* dx_add, float* dx, float* dr, float* dgamma, float* dbeta, long M, int D, float drop_p, uint64_t seed,
                          uint32_t* dx_mag, uint32_t* dr_mag, void* stream);

bool deterministic_mode();
bool gemm_emu_mode();             // hoisdf_set_gemm_emu: ... as fp32 emulated on the bf16 MFMA pipe (default on)
struct DetScratch {
  float* part;          // [gridDim.x][ncols] partials
  unsigned* ticket;     // arrival counter, zero between launches
  int on;
};
DetScratch det_scratch(size_t floats);   // on = 0 when the mode is off (or the scratch cannot be allocated)

// All threads of the block call this with the block's column sums in shared memory: src[0..nA) belong to dstA,
// src[nA..nA+nB) to dstB (pre-zeroed destinations).  Either one float atomic per column (default) or the ordered
// last-block reduction.
__device__ __forceinline__ void block_column_sum(float* dstA, int nA, float* dstB, int nB,
                                                 const float* src, unsigned* s_last /* one shared word */,
                                                 const DetScratch ds) {
  const int n = nA + nB;
  if (!ds.on) {
    for (int c = threadIdx.x; c < n; c += blockDim.x) atomicAdd(c < nA ? dstA + c : dstB + (c - nA), src[c]);
    return;
  }
  for (int c = threadIdx.x; c < n; c += blockDim.x) ds.part[(size_t)blockIdx.x * n + c] = src[c];
  __threadfence();                              // agent-scope release of this block's partials
  __syncthreads();
  if (threadIdx.x == 0) *s_last = (atomicAdd(ds.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
  __syncthreads();
  if (!*s_last) return;
  __threadfence();                              // acquire: the other blocks' partials (other XCDs' L2 included)
  const volatile float* part = ds.part;
  for (int c = threadIdx.x; c < n; c += blockDim.x) {
    float acc = 0.f;
    for (unsigned b = 0; b < gridDim.x; ++b) acc += part[(size_t)b * n + c];
    if (c < nA) dstA[c] += acc; else dstB[c - nA] += acc;
  }
  if (threadIdx.x == 0) *ds.ticket = 0u;
}

// ---- wave-level reductions (64 lanes) -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a linear block id: consecutive logical ids land on the same
// XCD (block b is dispatched to XCD b % 8), so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int nx = 8;
  int xcd = bid % nx, loc = bid / nx;
  int q = nblk / nx, r = nblk % nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}

}  // namespace hoisdf
