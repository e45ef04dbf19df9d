// Helpers shared by the host-side chains of the coarse entries (layers.hip, sdf_query.hip): a bump allocator over caller buffers
// (with a null base it only measures - the size queries run the same carving code as the real calls) and the linear-layer
// dispatch of the library defaults (fp32 emulated on the bf16 pipe from 2048 rows up, otherwise the exact-f32 kernels).
#pragma once
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "common.h"

namespace hoisdf {
namespace {

constexpr long EMU_MIN_ROWS = 2048;      // below: a handful of tiles, latency-bound - the exact-f32 kernel
constexpr long EMU_DW_MIN_ROWS = 8192;   // grad-weight contracts over the rows: >= 32 slabs per slice at 256 slices

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// bump allocator over a caller buffer; with base == nullptr it only measures
struct Bump {
  char* base; long cap; long off = 0; bool overflow = false;
  Bump(void* b, long c) : base(static_cast<char*>(b)), cap(c) {}
  void* take(long bytes) {
    off = (off + 255) & ~255L;
    const long at = off;
    off += bytes;
    if (!base) return nullptr;
    if (off > cap) { overflow = true; return nullptr; }
    return base + at;
  }
  float* floats(long n) { return static_cast<float*>(take(n * 4)); }
};

// Whether a forward call left row magnitudes in its saved block depends on process-wide switches (hoisdf_set_gemm_emu, the form): the
// forward RECORDS what it did per block (keyed by the address of the block's magnitude region) and the backward of that block asks
// here instead of re-reading the switches - one flipped in between (bench.py and the cfg setters do flip them at run time) would
// otherwise make the backward trust words nobody wrote.  -1: no record (a host that never ran the forward through this library).
struct SavedMags {
  std::mutex mu;
  std::unordered_map<const void*, int> on;
  void put(const void* key, bool v) { if (!key) return; std::lock_guard<std::mutex> l(mu); if (on.size() > 8192) on.clear(); on[key] = v ? 1 : 0; }
  int get(const void* key) { std::lock_guard<std::mutex> l(mu); auto it = on.find(key); return it == on.end() ? -1 : it->second; }
};
inline SavedMags& saved_mags() { static SavedMags s; return s; }

struct Ctx {
  hipStream_t st; void* stream;
  Bump* ws;
  bool dry;          // measuring pass: no launches
  bool emu;          // library mode at entry
  int rc = HOISDF_OK;
  bool ok() const { return rc == HOISDF_OK; }
};

// HOISDF_EMU_SMALL=0: small-M linear layers stay on the exact-f32 tiled kernel (the round-3 flow; A/B runs)
inline bool emu_small_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("HOISDF_EMU_SMALL"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on == 1;
}
// small row counts (decoder stack, heads): the one-wave-per-tile emulated form (gemm_emu_small.hip)
inline bool emu_small(const Ctx& c, long M, const float* a, long lda, const float* W, long ldw, int N, int K) {
  if (!c.emu || !emu_small_enabled() || M > hoisdf_linear_emu_small_max_rows() || N % 4 || K % 4 || lda % 4 || ldw % 4) return false;
  return c.dry || hoisdf_linear_emu_small_supported(a, lda, W, ldw, M, N, K);
}
bool emu_rows(const Ctx& c, long M, const float* a, long lda, int contraction) {
  // (dry pass: pointers are null - assume aligned, which the real pass then checks again; the workspace is an upper bound)
  return c.emu && M >= EMU_MIN_ROWS && contraction % 4 == 0 && lda % 4 == 0 && (c.dry || al16(a));
}
const void* image_of(Ctx& c, const void* given, const float* W, int ldw, int N, int K, int transpose) {
  if (given) return given;
  void* img = c.ws->take(hoisdf_linear_emu_image_bytes(transpose ? K : N, transpose ? N : K));
  if (c.dry) return nullptr;
  if (!img) { c.rc = HOISDF_ERR_WORKSPACE; return nullptr; }
  c.rc = hoisdf_linear_emu_prepare(W, ldw, N, K, transpose, img, c.stream);
  return img;
}
// K_pad >= K (default K): columns K .. K_pad - 1 of x are zero padding; the emulated form contracts over K_pad (they meet the zero
// fill of the weight image, which is built from the K real columns) so that a ragged K (289) still takes the bf16 pipe
void lin_fwd(Ctx& c, const float* x, int ldx, const float* W, int ldw, const void* img, const float* b, float* y, int ldy, long M, int N,
             int K, int act, float p, uint64_t seed, uint32_t* bits, int K_pad = 0, const uint32_t* x_mag = nullptr, uint32_t* y_mag = nullptr,
             uint32_t* y_heads = nullptr, int head_L = 0) {
  // x_mag / y_mag: row magnitudes of x (null: unknown) and for y (null: not wanted; zero on entry) - common.h; only the emulated
  // tiled form reads / writes them: a caller that hands y_mag on must know that form ran (emu_rows of the same arguments);
  // y_heads: head magnitudes of y (samples of head_L rows), the same way
  if (!c.ok()) return;
  if (K_pad < K || (K_pad + 15) / 16 != (K + 15) / 16) K_pad = K;
  if (emu_rows(c, M, x, ldx, K_pad)) {
    const void* im = image_of(c, img, W, ldw, N, K, 0);
    if (c.dry || !c.ok()) return;
    c.rc = linear_fwd_emu_mag(x, ldx, im, b, y, ldy, M, N, K_pad, act, p, seed, bits, x_mag, y_mag, c.stream, y_heads, head_L);
    return;
  }
  if (c.dry) return;
  if (emu_small(c, M, x, ldx, W, ldw, N, K)) { c.rc = hoisdf_linear_fwd_emu_small(x, ldx, W, ldw, b, y, ldy, M, N, K, act, p, seed, bits, c.stream); return; }
  c.rc = hoisdf_linear_fwd(x, ldx, W, ldw, b, y, ldy, M, N, K, act, p, seed, bits, c.stream);
}
void lin_bwd_input(Ctx& c, const float* dy, int lddy, const uint32_t* bits, float p, const float* W, int ldw, const void* img_t, float* dx,
                   int lddx, long M, int N, int K, int accumulate, const uint32_t* dy_mag = nullptr, uint32_t* dx_mag = nullptr,
                   uint32_t* dx_heads = nullptr, int head_L = 0) {
  if (!c.ok()) return;
  if (!bits) p = 0.f;
  if (emu_rows(c, M, dy, lddy, N)) {
    const void* im = image_of(c, img_t, W, ldw, N, K, 1);
    if (c.dry || !c.ok()) return;
    c.rc = linear_bwd_input_emu_mag(dy, lddy, bits, p, im, dx, lddx, M, N, K, accumulate, dy_mag, accumulate ? nullptr : dx_mag, c.stream,
                                    accumulate ? nullptr : dx_heads, head_L);
    return;
  }
  if (c.dry) return;
  if (emu_small(c, M, dy, lddy, W, ldw, N, K)) { c.rc = hoisdf_linear_bwd_input_emu_small(dy, lddy, bits, p, W, ldw, dx, lddx, M, N, K, accumulate, c.stream); return; }
  c.rc = hoisdf_linear_bwd_input(dy, lddy, bits, p, W, ldw, dx, lddx, M, N, K, accumulate, c.stream);
}
// dW / db zero on entry (the exact-f32 kernel accumulates, the emulated one overwrites)
// K_pad > K: dW is a dense [N][K_pad] and columns K .. K_pad - 1 of x are zero padding - both forms contract the padded width
// (the pad columns of dW come out zero)
// dy_mag / x_mag (f16x2 form): magnitude words of dy / x; with either one given the grad-weight runs in the f16x2 form (a missing one
// is measured by the library), with neither in the bf16x3 form, which needs no magnitudes (no weight image is involved: the two
// forms coexist per call)
void lin_bwd_weight(Ctx& c, const float* dy, int lddy, const uint32_t* bits, float p, const float* x, int ldx, float* dW, float* db, long M,
                    int N, int K, int K_pad = 0, const uint32_t* dy_mag = nullptr, const uint32_t* x_mag = nullptr) {
  if (!c.ok()) return;
  if (!bits) p = 0.f;
  if (K_pad < K) K_pad = K;
  const bool emu = c.emu && M >= EMU_DW_MIN_ROWS && (N < K ? N : K) >= 64 && N % 4 == 0 && K_pad % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 &&
                   (c.dry || (al16(dy) && al16(x) && al16(dW)));
  if (emu) {
    const long nws = hoisdf_linear_bwd_weight_emu_workspace(M, N, K_pad);
    float* w = c.ws->floats(nws > 4 ? nws : 4);   // (scratch of consecutive calls is not recycled: earlier launches may still read theirs)
    if (!c.dry) {
      if (!w) { c.rc = HOISDF_ERR_WORKSPACE; return; }
      if (dy_mag || x_mag) c.rc = linear_bwd_weight_emu_mag(dy, lddy, bits, p, x, ldx, dW, K_pad, db, M, N, K_pad, w, nws, dy_mag, x_mag, c.stream);
      else c.rc = hoisdf_linear_bwd_weight_emu(dy, lddy, bits, p, x, ldx, dW, K_pad, db, M, N, K_pad, w, nws, c.stream);
    }
    return;
  }
  if (c.emu && emu_small_enabled() && M <= hoisdf_linear_emu_small_max_rows()) {       // (no alignment demands: scalar loads)
    if (!c.dry) c.rc = hoisdf_linear_bwd_weight_emu_small(dy, lddy, bits, p, x, ldx, dW, K_pad, db, M, N, K_pad, c.stream);
    return;
  }
  long nws = 0; float* w = nullptr;
  if (deterministic_mode()) {
    nws = hoisdf_linear_bwd_weight_workspace(M, N, K_pad);
    if (nws > 0) w = c.ws->floats(nws);
    if (!c.dry && nws > 0 && !w) { c.rc = HOISDF_ERR_WORKSPACE; return; }
  }
  if (c.dry) return;
  c.rc = hoisdf_linear_bwd_weight(dy, lddy, bits, p, x, ldx, dW, K_pad, db, M, N, K_pad, w, nws, c.stream);
}


}  // namespace
}  // namespace hoisdf
