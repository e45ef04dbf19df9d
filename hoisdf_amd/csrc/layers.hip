// Coarse C entries of SURVEY.md section 8(b): one transformer encoder layer (common/nets/transformer.py:286-302,
// TransformerEncoderLayer.forward_post, plus the stack's inter_norm of the layer output, :117-131) forward and backward as ONE
// call each.  Host-side chains of the per-op launches in this library - in-projection, attention, out-projection, residual +
// dropout + LayerNorm, FFN with the fused ReLU / dropout epilogue and sign bitmap, second LayerNorm, inter_norm - on the caller's
// stream, over caller-provided buffers: `saved` holds what the backward re-reads (activations, softmax statistics, the ReLU
// bitmap), `workspace` is scratch.  The arithmetic follows the library defaults: linear layers of >= 2048 rows as fp32 emulated
// on the bf16 MFMA pipe (hoisdf_linear_*_emu; hoisdf_set_gemm_emu(0) / HOISDF_GEMM=f32: the exact-f32 kernels), attention as
// the descriptor says.  The opt-in reduced-operand modes (split precision, f16 eval attention) are not offered here.
// hoisdf_amd/ops.py's encoder_layer autograd node is a thin wrapper of these two calls.
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

// dst[g][r][:] (group stride dst_gs floats) (+)= src[g][r][:] (group stride src_gs) for r < rows, E floats a row (E % 4 == 0)
__global__ void rows_copy_add_kernel(float* __restrict__ dst, long dst_gs, const float* __restrict__ src, long src_gs, int rows, int E4,
                                     long total, int add) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long per = (long)rows * E4;
  const long g = i / per, r = i - g * per;
  float4* d = reinterpret_cast<float4*>(dst + g * dst_gs) + r;
  const float4 s = reinterpret_cast<const float4*>(src + g * src_gs)[r];
  if (add) { float4 t = *d; t.x += s.x; t.y += s.y; t.z += s.z; t.w += s.w; *d = t; }
  else *d = s;
}
int rows_copy_add(float* dst, long dst_gs, const float* src, long src_gs, int groups, int rows, int E, int add, hipStream_t st) {
  const long total = (long)groups * rows * (E / 4);
  if (total == 0) return HOISDF_OK;
  hipLaunchKernelGGL(rows_copy_add_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, dst, dst_gs, src, src_gs, rows, E / 4, total, add);
  return check_launch("encoder_layer rows copy/add");
}

struct Ctx {
  hipStream_t st; void* stream;
  Bump* ws;
  bool dry;          // measuring pass: no launches
  bool emu;          // library mode at entry
  int rc = HOISDF_OK;
  bool ok() const { return rc == HOISDF_OK; }
};

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
void lin_fwd(Ctx& c, const float* x, int ldx, const float* W, int ldw, const void* img, const float* b, float* y, int ldy, long M, int N,
             int K, int act, float p, uint64_t seed, uint32_t* bits) {
  if (!c.ok()) return;
  if (emu_rows(c, M, x, ldx, K)) {
    const void* im = image_of(c, img, W, ldw, N, K, 0);
    if (c.dry || !c.ok()) return;
    c.rc = hoisdf_linear_fwd_emu(x, ldx, im, b, y, ldy, M, N, K, act, p, seed, bits, c.stream);
    return;
  }
  if (c.dry) return;
  c.rc = hoisdf_linear_fwd(x, ldx, W, ldw, b, y, ldy, M, N, K, act, p, seed, bits, c.stream);
}
void lin_bwd_input(Ctx& c, const float* dy, int lddy, const uint32_t* bits, float p, const float* W, int ldw, const void* img_t, float* dx,
                   int lddx, long M, int N, int K, int accumulate) {
  if (!c.ok()) return;
  if (!bits) p = 0.f;
  if (emu_rows(c, M, dy, lddy, N)) {
    const void* im = image_of(c, img_t, W, ldw, N, K, 1);
    if (c.dry || !c.ok()) return;
    c.rc = hoisdf_linear_bwd_input_emu(dy, lddy, bits, p, im, dx, lddx, M, N, K, accumulate, c.stream);
    return;
  }
  if (c.dry) return;
  c.rc = hoisdf_linear_bwd_input(dy, lddy, bits, p, W, ldw, dx, lddx, M, N, K, accumulate, c.stream);
}
// dW / db zero on entry (the exact-f32 kernel accumulates, the emulated one overwrites)
void lin_bwd_weight(Ctx& c, const float* dy, int lddy, const uint32_t* bits, float p, const float* x, int ldx, float* dW, float* db, long M,
                    int N, int K) {
  if (!c.ok()) return;
  if (!bits) p = 0.f;
  const bool emu = c.emu && M >= EMU_DW_MIN_ROWS && (N < K ? N : K) >= 64 && N % 4 == 0 && K % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 &&
                   (c.dry || (al16(dy) && al16(x) && al16(dW)));
  if (emu) {
    const long nws = hoisdf_linear_bwd_weight_emu_workspace(M, N, K);
    float* w = c.ws->floats(nws > 4 ? nws : 4);   // (scratch of consecutive calls is not recycled: earlier launches may still read theirs)
    if (!c.dry) {
      if (!w) { c.rc = HOISDF_ERR_WORKSPACE; return; }
      c.rc = hoisdf_linear_bwd_weight_emu(dy, lddy, bits, p, x, ldx, dW, K, db, M, N, K, w, nws, c.stream);
    }
    return;
  }
  long nws = 0; float* w = nullptr;
  if (deterministic_mode()) {
    nws = hoisdf_linear_bwd_weight_workspace(M, N, K);
    if (nws > 0) w = c.ws->floats(nws);
    if (!c.dry && nws > 0 && !w) { c.rc = HOISDF_ERR_WORKSPACE; return; }
  }
  if (c.dry) return;
  c.rc = hoisdf_linear_bwd_weight(dy, lddy, bits, p, x, ldx, dW, K, db, M, N, K, w, nws, c.stream);
}

struct Geo {
  int B, S, E, F, H, nq, ni; bool full; long M, Ms; float eps, p; int att, att_bwd_emu;
};
int geometry(const hoisdf_encoder_layer_desc* d, Geo& g) {
  HOISDF_REQUIRE(d, HOISDF_ERR_INVALID, "encoder_layer: null descriptor");
  HOISDF_REQUIRE(d->B > 0 && d->S > 0 && d->E > 0 && d->F > 0 && d->H > 0 && d->E % d->H == 0 && d->E % 4 == 0 && d->F % 4 == 0,
                 HOISDF_ERR_INVALID, "encoder_layer: bad sizes B=%d S=%d E=%d F=%d H=%d", d->B, d->S, d->E, d->F, d->H);
  HOISDF_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, HOISDF_ERR_INVALID, "encoder_layer: drop_p=%f", d->drop_p);
  HOISDF_REQUIRE(d->attention == 0 || d->attention == 2, HOISDF_ERR_INVALID, "encoder_layer: attention must be 0 (exact f32) or 2 (emulated fp32)");
  g.B = d->B; g.S = d->S; g.E = d->E; g.F = d->F; g.H = d->H;
  g.nq = (d->n_query <= 0 || d->n_query >= d->S) ? d->S : d->n_query;
  g.ni = (d->n_inter <= 0 || d->n_inter >= g.nq) ? g.nq : d->n_inter;
  g.full = g.nq == g.S;
  g.M = (long)g.B * g.nq; g.Ms = (long)g.B * g.S;
  g.eps = d->eps; g.p = d->drop_p;
  g.att = g.nq < 32 ? 0 : d->attention;                     // (the emulated kernels tile 32 queries)
  g.att_bwd_emu = g.att == 2 && d->attention_bwd_emulated && d->training;
  return HOISDF_OK;
}

// what the forward leaves for the backward, carved from `saved` in this order
struct Saved {
  float* qkv; float* qbuf; float* kvbuf; float* xq; float* o; float* lse; float* a; float* x1; float* h; uint32_t* bits; float* f; float* st;
  void* planes; long planes_bytes;
};
void carve_saved(const Geo& g, Bump& b, Saved& s) {
  const int E = g.E;
  s.qkv = s.qbuf = s.kvbuf = s.xq = nullptr;
  if (g.full) s.qkv = b.floats(g.Ms * 3 * E);
  else { s.qbuf = b.floats(g.M * E); s.kvbuf = b.floats(g.Ms * 2 * E); s.xq = b.floats(g.M * E); }
  s.o = b.floats(g.M * E); s.lse = b.floats((long)g.B * g.H * g.nq); s.a = b.floats(g.M * E); s.x1 = b.floats(g.M * E);
  s.h = b.floats(g.M * g.F); s.bits = static_cast<uint32_t*>(b.take(g.M * ((g.F + 31) / 32) * 4)); s.f = b.floats(g.M * E);
  s.st = b.floats(6 * g.M);
  s.planes_bytes = g.att_bwd_emu ? hoisdf_attention_emu_workspace(g.B, g.H, g.nq, g.S, 2) : 0;
  s.planes = s.planes_bytes ? b.take(s.planes_bytes) : nullptr;
}

int forward(const float* x, const hoisdf_encoder_layer_weights* w, const hoisdf_encoder_layer_desc* d, const Geo& g, float* x_out, float* y_out,
            Bump& saved, Bump& ws, bool dry, void* stream) {
  Ctx c{as_stream(stream), stream, &ws, dry, gemm_emu_mode()};
  Saved s;
  carve_saved(g, d->training ? saved : ws, s);
  const int E = g.E, F = g.F;
  const float *q, *k, *v; int ldq, ldkv;
  const float* xq2 = x;
  if (g.full) {
    lin_fwd(c, x, E, w->w_in, E, w->img_in, w->b_in, s.qkv, 3 * E, g.Ms, 3 * E, E, 0, 0.f, 0, nullptr);
    q = s.qkv; k = s.qkv + E; v = s.qkv + 2 * E; ldq = ldkv = 3 * E;
  } else {
    if (!dry && c.ok()) c.rc = rows_copy_add(s.xq, (long)g.nq * E, x, (long)g.S * E, g.B, g.nq, E, 0, c.st);
    xq2 = s.xq;
    lin_fwd(c, s.xq, E, w->w_in, E, w->img_in_q, w->b_in, s.qbuf, E, g.M, E, E, 0, 0.f, 0, nullptr);
    lin_fwd(c, x, E, w->w_in + (size_t)E * E, E, w->img_in_kv, w->b_in ? w->b_in + E : nullptr, s.kvbuf, 2 * E, g.Ms, 2 * E, E, 0, 0.f, 0, nullptr);
    q = s.qbuf; k = s.kvbuf; v = s.kvbuf + E; ldq = E; ldkv = 2 * E;
  }
  if (g.att == 2) {
    void* aw = s.planes; long ab = s.planes_bytes;
    if (!aw) { ab = hoisdf_attention_emu_workspace(g.B, g.H, g.nq, g.S, 0); aw = ws.take(ab); }
    if (!dry && c.ok()) {
      if (!aw) c.rc = HOISDF_ERR_WORKSPACE;
      else c.rc = hoisdf_attention_fwd_emu(q, ldq, k, ldkv, v, ldkv, s.o, E, s.lse, g.B, g.H, g.nq, g.S, g.S, g.p, d->seed[0], aw, ab,
                                           s.planes ? 1 : 0, stream);
    }
  } else if (!dry && c.ok()) {
    c.rc = hoisdf_attention_fwd(q, ldq, k, ldkv, v, ldkv, s.o, E, s.lse, g.B, g.H, g.nq, g.S, g.S, g.p, d->seed[0], stream);
  }
  lin_fwd(c, s.o, E, w->w_out, E, w->img_out, w->b_out, s.a, E, g.M, E, E, 0, 0.f, 0, nullptr);
  float* st = s.st;
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_fwd(xq2, s.a, w->g1, w->be1, s.x1, st, st + g.M, g.M, E, g.eps, g.p, d->seed[1], stream);
  lin_fwd(c, s.x1, E, w->w1, E, w->img_1, w->b1, s.h, F, g.M, F, E, 1, g.p, d->seed[2], s.bits);
  lin_fwd(c, s.h, F, w->w2, F, w->img_2, w->b2, s.f, E, g.M, E, F, 0, 0.f, 0, nullptr);
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_fwd(s.x1, s.f, w->g2, w->be2, x_out, st + 2 * g.M, st + 3 * g.M, g.M, E, g.eps, g.p, d->seed[3], stream);
  if (w->g3 && !dry && c.ok()) {
    if (g.ni == g.nq) c.rc = hoisdf_add_layernorm_fwd(x_out, nullptr, w->g3, w->be3, y_out, st + 4 * g.M, st + 5 * g.M, g.M, E, g.eps, 0.f, 0, stream);
    else c.rc = hoisdf_layernorm_rows_fwd(x_out, w->g3, w->be3, y_out, st + 4 * g.M, st + 5 * g.M, g.B, g.nq, g.ni, E, g.eps, stream);
  }
  if (c.ok() && !dry && (ws.overflow || saved.overflow)) { set_error("encoder_layer_fwd: workspace / saved buffer too small"); return HOISDF_ERR_WORKSPACE; }
  return c.rc;
}

int backward(const float* x, const float* x_out, const hoisdf_encoder_layer_weights* w, const hoisdf_encoder_layer_desc* d, const Geo& g,
             Bump& saved, const float* g_x_out, const float* g_y, float* dx, const hoisdf_encoder_layer_grads* G, Bump& ws, bool dry,
             void* stream) {
  Ctx c{as_stream(stream), stream, &ws, dry, gemm_emu_mode()};
  Saved s;
  carve_saved(g, saved, s);
  const int E = g.E, F = g.F;
  const long M = g.M, Ms = g.Ms;
  const float* st = s.st;
  // inter_norm backward joins the gradient of the layer output itself
  const float* dx2 = g_x_out;
  if (g_y && w->g3) {
    float* t = ws.floats(M * E);
    if (!dry && c.ok()) {
      if (!t) c.rc = HOISDF_ERR_WORKSPACE;
      else if (g.ni == g.nq) c.rc = hoisdf_add_layernorm_bwd(g_y, x_out, nullptr, w->g3, st + 4 * M, st + 5 * M, g_x_out, t, nullptr, G->dg3, G->dbe3, M, E, 0.f, 0, stream);
      else c.rc = hoisdf_layernorm_rows_bwd(g_y, x_out, w->g3, st + 4 * M, st + 5 * M, g_x_out, t, G->dg3, G->dbe3, g.B, g.nq, g.ni, E, stream);
    }
    dx2 = t;
  }
  float* dx1 = ws.floats(M * E); float* df = ws.floats(M * E); float* dh = ws.floats(M * F);
  float* dxq = g.full ? dx : ws.floats(M * E);                      // all rows produced: the residual gradient IS dx
  float* da = ws.floats(M * E); float* dO = ws.floats(M * E);
  float* delta = ws.floats((long)g.B * g.H * g.nq);
  if (!dry && (!dx1 || !df || !dh || !dxq || !da || !dO || !delta)) { set_error("encoder_layer_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
  if (!dry && !dx2) { set_error("encoder_layer_bwd: no upstream gradient (g_x_out and g_y both null)"); return HOISDF_ERR_INVALID; }
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_bwd(dx2, s.x1, s.f, w->g2, st + 2 * M, st + 3 * M, nullptr, dx1, df, G->dg2, G->dbe2, M, E, g.p, d->seed[3], stream);
  lin_bwd_input(c, df, E, nullptr, 0.f, w->w2, F, w->img_t_2, dh, F, M, E, F, 0);
  lin_bwd_weight(c, df, E, nullptr, 0.f, s.h, F, G->dw2, G->db2, M, E, F);
  lin_bwd_input(c, dh, F, s.bits, g.p, w->w1, E, w->img_t_1, dx1, E, M, F, E, 1);            // dx1 += : the FFN branch joins the residual
  lin_bwd_weight(c, dh, F, s.bits, g.p, s.x1, E, G->dw1, G->db1, M, F, E);
  const float* xq2 = g.full ? x : s.xq;
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_bwd(dx1, xq2, s.a, w->g1, st, st + M, nullptr, dxq, da, G->dg1, G->dbe1, M, E, g.p, d->seed[1], stream);
  lin_bwd_input(c, da, E, nullptr, 0.f, w->w_out, E, w->img_t_out, dO, E, M, E, E, 0);
  lin_bwd_weight(c, da, E, nullptr, 0.f, s.o, E, G->dw_out, G->db_out, M, E, E);
  auto attn_bwd = [&](const float* q, int ldq, const float* k, const float* v, int ldkv, float* dq, float* dk, float* dv) {
    if (g.att_bwd_emu) {
      const long ab = hoisdf_attention_bwd_emu_workspace(g.B, g.H, g.nq, g.S, 1);
      void* aw = ws.take(ab);
      if (dry || !c.ok()) return;
      if (!aw) { c.rc = HOISDF_ERR_WORKSPACE; return; }
      c.rc = hoisdf_attention_bwd_emu(q, ldq, k, ldkv, v, ldkv, s.o, E, dO, E, s.lse, delta, dq, dk, dv, g.B, g.H, g.nq, g.S, g.S, g.p, d->seed[0],
                                      s.planes, aw, ab, stream);
    } else if (!dry && c.ok()) {
      c.rc = hoisdf_attention_bwd(q, ldq, k, ldkv, v, ldkv, s.o, E, dO, E, s.lse, delta, dq, dk, dv, g.B, g.H, g.nq, g.S, g.S, g.p, d->seed[0], stream);
    }
  };
  if (g.full) {
    float* dqkv = ws.floats(Ms * 3 * E);
    if (!dry && !dqkv) { set_error("encoder_layer_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
    attn_bwd(s.qkv, 3 * E, s.qkv + E, s.qkv + 2 * E, 3 * E, dqkv, dqkv + E, dqkv + 2 * E);
    lin_bwd_input(c, dqkv, 3 * E, nullptr, 0.f, w->w_in, E, w->img_t_in, dxq, E, Ms, 3 * E, E, 1);   // += : attention branch joins the residual
    lin_bwd_weight(c, dqkv, 3 * E, nullptr, 0.f, x, E, G->dw_in, G->db_in, Ms, 3 * E, E);
  } else {
    float* dq = ws.floats(M * E); float* dkv = ws.floats(Ms * 2 * E);
    if (!dry && (!dq || !dkv)) { set_error("encoder_layer_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
    attn_bwd(s.qbuf, E, s.kvbuf, s.kvbuf + E, 2 * E, dq, dkv, dkv + E);
    lin_bwd_input(c, dq, E, nullptr, 0.f, w->w_in, E, w->img_t_in_q, dxq, E, M, E, E, 1);
    lin_bwd_weight(c, dq, E, nullptr, 0.f, s.xq, E, G->dw_in, G->db_in, M, E, E);
    lin_bwd_input(c, dkv, 2 * E, nullptr, 0.f, w->w_in + (size_t)E * E, E, w->img_t_in_kv, dx, E, Ms, 2 * E, E, 0);
    lin_bwd_weight(c, dkv, 2 * E, nullptr, 0.f, x, E, G->dw_in + (size_t)E * E, G->db_in ? G->db_in + E : nullptr, Ms, 2 * E, E);
    if (!dry && c.ok()) c.rc = rows_copy_add(dx, (long)g.S * E, dxq, (long)g.nq * E, g.B, g.nq, E, 1, c.st);
  }
  if (c.ok() && !dry && ws.overflow) { set_error("encoder_layer_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
  return c.rc;
}

}  // namespace
}  // namespace hoisdf

using namespace hoisdf;

extern "C" long hoisdf_encoder_layer_saved_bytes(const hoisdf_encoder_layer_desc* d) {
  Geo g;
  if (geometry(d, g) != HOISDF_OK) return -1;
  Bump b(nullptr, 0); Saved s;
  carve_saved(g, b, s);
  return b.off + 256;
}

extern "C" long hoisdf_encoder_layer_workspace_bytes(const hoisdf_encoder_layer_desc* d, int backward_pass) {
  Geo g;
  if (geometry(d, g) != HOISDF_OK) return -1;
  // measuring pass with no weight images given: the upper bound (images the caller does pass are simply not built)
  hoisdf_encoder_layer_weights w{};
  hoisdf_encoder_layer_grads G{};
  float dummy = 0.f;
  w.g3 = &dummy;
  Bump saved(nullptr, 0), ws(nullptr, 0);
  if (backward_pass) (void)backward(nullptr, nullptr, &w, d, g, saved, &dummy, &dummy, nullptr, &G, ws, true, nullptr);
  else (void)forward(nullptr, &w, d, g, nullptr, nullptr, saved, ws, true, nullptr);
  return ws.off + 256;
}

extern "C" int hoisdf_encoder_layer_fwd(const float* x, const hoisdf_encoder_layer_weights* w, const hoisdf_encoder_layer_desc* d, float* x_out,
                                        float* y_out, void* saved, long saved_bytes, void* workspace, long workspace_bytes, void* stream) {
  Geo g;
  if (int rc = geometry(d, g)) return rc;
  HOISDF_REQUIRE(x && w && x_out && w->w_in && w->w_out && w->w1 && w->w2 && w->g1 && w->be1 && w->g2 && w->be2, HOISDF_ERR_INVALID,
                 "encoder_layer_fwd: null pointer");
  HOISDF_REQUIRE(!w->g3 || (w->be3 && y_out), HOISDF_ERR_INVALID, "encoder_layer_fwd: inter_norm needs be3 and y_out");
  HOISDF_REQUIRE(!d->training || saved, HOISDF_ERR_WORKSPACE, "encoder_layer_fwd: training needs the saved buffer (hoisdf_encoder_layer_saved_bytes)");
  HOISDF_REQUIRE(workspace || workspace_bytes == 0, HOISDF_ERR_WORKSPACE, "encoder_layer_fwd: null workspace");
  HOISDF_REQUIRE(al16(x) && al16(x_out) && al16(saved) && al16(workspace), HOISDF_ERR_INVALID, "encoder_layer_fwd: buffers must be 16-byte aligned");
  Bump sv(saved, saved_bytes), ws(workspace, workspace_bytes);
  static char none;                                  // (a real pass never measures: a null buffer is an empty one)
  if (!sv.base) { sv.base = &none; sv.cap = 0; }
  if (!ws.base) { ws.base = &none; ws.cap = 0; }
  const int rc = forward(x, w, d, g, x_out, y_out, sv, ws, false, stream);
  if (rc == HOISDF_ERR_WORKSPACE) set_error("encoder_layer_fwd: workspace (%ld bytes) or saved buffer (%ld bytes) too small", workspace_bytes, saved_bytes);
  return rc;
}

extern "C" int hoisdf_encoder_layer_bwd(const float* x, const float* x_out, const hoisdf_encoder_layer_weights* w, const hoisdf_encoder_layer_desc* d,
                                        const void* saved, long saved_bytes, const float* g_x_out, const float* g_y, float* dx,
                                        const hoisdf_encoder_layer_grads* grads, void* workspace, long workspace_bytes, void* stream) {
  Geo g;
  if (int rc = geometry(d, g)) return rc;
  HOISDF_REQUIRE(x && x_out && w && saved && dx && grads && workspace, HOISDF_ERR_INVALID, "encoder_layer_bwd: null pointer");
  HOISDF_REQUIRE(grads->dw_in && grads->db_in && grads->dw_out && grads->db_out && grads->dg1 && grads->dbe1 && grads->dw1 && grads->db1 &&
                     grads->dw2 && grads->db2 && grads->dg2 && grads->dbe2 && (!w->g3 || !g_y || (grads->dg3 && grads->dbe3)),
                 HOISDF_ERR_INVALID, "encoder_layer_bwd: every parameter gradient buffer is required (zero-filled)");
  HOISDF_REQUIRE(d->training, HOISDF_ERR_INVALID, "encoder_layer_bwd: the forward call must have run with training = 1");
  HOISDF_REQUIRE(al16(x) && al16(dx) && al16(saved) && al16(workspace), HOISDF_ERR_INVALID, "encoder_layer_bwd: buffers must be 16-byte aligned");
  Bump sv(const_cast<void*>(saved), saved_bytes), ws(workspace, workspace_bytes);
  const int rc = backward(x, x_out, w, d, g, sv, g_x_out, g_y, dx, grads, ws, false, stream);
  if (rc == HOISDF_OK && sv.overflow) { set_error("encoder_layer_bwd: saved buffer too small"); return HOISDF_ERR_WORKSPACE; }
  if (rc == HOISDF_ERR_WORKSPACE) set_error("encoder_layer_bwd: workspace (%ld bytes) too small", workspace_bytes);
  return rc;
}
