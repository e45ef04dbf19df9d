// Coarse C entries of SURVEY.md section 8(b): one transformer encoder layer (common/nets/transformer.py:286-302,
// TransformerEncoderLayer.forward_post, plus the stack's inter_norm of the layer output, :117-131) forward and backward as ONE
// call each; further down the same for one decoder layer (:366-395).  Host-side chains of the per-op launches in this library - in-projection, attention, out-projection, residual +
// dropout + LayerNorm, FFN with the fused ReLU / dropout epilogue and sign bitmap, second LayerNorm, inter_norm - on the caller's
// stream, over caller-provided buffers: `saved` holds what the backward re-reads (activations, softmax statistics, the ReLU
// bitmap), `workspace` is scratch.  The arithmetic follows the library defaults: linear layers of >= 2048 rows as fp32 emulated
// on the bf16 MFMA pipe (hoisdf_linear_*_emu; hoisdf_set_gemm_emu(0) / HOISDF_GEMM=f32: the exact-f32 kernels), attention as
// the descriptor says.  The opt-in reduced-operand mode (f16 eval attention) is not offered here.
// hoisdf_amd/ops.py's encoder_layer / decoder_layer autograd nodes are thin wrappers of these calls.
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "chain.h"

namespace hoisdf {
namespace {

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

struct Geo {
  int B, S, E, F, H, nq, ni; bool full; long M, Ms; float eps, p; int att, att_bwd_emu;
  bool fused_qkv;      // the in-projection writes the attention planes directly (linear_fwd_emu_qkv): no f32 q / k / v, no conversion pass
  bool att_h2;         // the attention forward in the f16x2 form: the in-projection writes f32 q / k / v and leaves their magnitude words, the
                       // conversion pass makes two scaled f16 planes per operand from them (attention_fwd_emu_mag)
};
// HOISDF_ATTN_FORM=b3: the attention forward of the layers stays in the bf16x3 form (A/B runs)
bool attn_h2_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("HOISDF_ATTN_FORM"); on = (e && (e[0] == 'b' || e[0] == 'B')) ? 0 : 1; }
  return on == 1;
}
// HOISDF_QKV_PLANES=0: the in-projection writes f32 q / k / v and the attention entry converts them (the round-3 flow; A/B runs)
bool qkv_planes_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("HOISDF_QKV_PLANES"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on == 1;
}
// The layout of `saved` depends on g.fused_qkv, which reads process-wide switches (hoisdf_set_gemm_emu, HOISDF_QKV_PLANES): the
// forward records its decision per saved buffer and the backward of that buffer takes it from here, so a switch flipped between a
// layer's forward and its backward (bench.py and the cfg.gemm_emu setter do flip it at run time) cannot make the two carve differently.
struct SavedForms {
  std::mutex mu;
  std::unordered_map<const void*, int> form;
  void put(const void* saved, int flags) { std::lock_guard<std::mutex> l(mu); if (form.size() > 4096) form.clear(); form[saved] = flags; }
  int take(const void* saved) {
    std::lock_guard<std::mutex> l(mu);
    auto it = form.find(saved);
    if (it == form.end()) return -1;
    const int f = it->second; form.erase(it);
    return f;
  }
};
SavedForms& saved_forms() { static SavedForms s; return s; }

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
  // emulated attention forward (and, in training, the emulated backward over the kept planes: nothing else reads q / k / v),
  // heads of 64, whole 128-token wave tiles per sample, the in-projection on the emulated GEMM
  // (f16x2 attention: where the layer's contractions run in the f16x2 form - the same test as layer_mags() below)
  g.att_h2 = attn_h2_enabled() && g.att == 2 && g.E == g.H * 64 && gemm_emu_mode() && emu_form_h2() && g.M >= EMU_MIN_ROWS;
  g.fused_qkv = !g.att_h2 && qkv_planes_enabled() && g.att == 2 && (g.att_bwd_emu || !d->training) && g.E == g.H * 64 && g.S % 128 == 0 &&
                g.nq % 128 == 0 && gemm_emu_mode() && g.M >= EMU_MIN_ROWS;
  return HOISDF_OK;
}

// what the forward leaves for the backward, carved from `saved` in this order
struct Saved {
  float* qkv; float* qbuf; float* kvbuf; float* xq; float* o; float* lse; float* a; float* x1; float* h; uint32_t* bits; float* f; float* st;
  void* planes; long planes_bytes;
  // row magnitudes (common.h) of o, x1, h, x_out (M words each) and the head magnitudes of the projected q, k, v (3 H B words: the
  // groups of [q | k | v], or of q followed by those of [k | v]) - one region, zeroed by the forward before its first launch
  uint32_t* mag; long mag_bytes;
};
enum { MAG_O = 0, MAG_X1 = 1, MAG_H = 2, MAG_XOUT = 3, MAG_ROWS = 4 };
void carve_saved(const Geo& g, Bump& b, Saved& s) {
  const int E = g.E;
  s.qkv = s.qbuf = s.kvbuf = s.xq = nullptr;
  if (g.full) { if (!g.fused_qkv) s.qkv = b.floats(g.Ms * 3 * E); }
  else {
    if (!g.fused_qkv) { s.qbuf = b.floats(g.M * E); s.kvbuf = b.floats(g.Ms * 2 * E); }
    s.xq = b.floats(g.M * E);
  }
  s.o = b.floats(g.M * E); s.lse = b.floats((long)g.B * g.H * g.nq); s.a = b.floats(g.M * E); s.x1 = b.floats(g.M * E);
  s.h = b.floats(g.M * g.F); s.bits = static_cast<uint32_t*>(b.take(g.M * ((g.F + 31) / 32) * 4)); s.f = b.floats(g.M * E);
  s.st = b.floats(6 * g.M);
  s.planes_bytes = g.att_bwd_emu ? hoisdf_attention_emu_workspace(g.B, g.H, g.nq, g.S, 2) : 0;     // (bf16x3 or, with att_h2, f16x2 planes: the backward runs in the forward's form)
  s.planes = s.planes_bytes ? b.take(s.planes_bytes) : nullptr;
  s.mag_bytes = ((long)MAG_ROWS * g.M + 3L * g.H * g.B) * 4;
  s.mag = static_cast<uint32_t*>(b.take(s.mag_bytes));
}
// the head magnitudes inside Saved::mag: q's groups, then k's, then v's (H B words each)
uint32_t* head_q(const Geo& g, uint32_t* mag) { return mag ? mag + (long)MAG_ROWS * g.M : nullptr; }
uint32_t* head_k(const Geo& g, uint32_t* mag) { return mag ? head_q(g, mag) + (long)g.H * g.B : nullptr; }
uint32_t* head_v(const Geo& g, uint32_t* mag) { return mag ? head_q(g, mag) + 2L * g.H * g.B : nullptr; }
// magnitude words travel between the kernels of a layer when its contractions run in the f16x2 form (every row count of a layer's
// contractions is M or Ms >= M: one test covers them all)
bool layer_mags(const Ctx& c, const Geo& g) { return c.emu && emu_form_h2() && g.M >= EMU_MIN_ROWS; }

int forward(const float* x, const hoisdf_encoder_layer_weights* w, const hoisdf_encoder_layer_desc* d, const Geo& g, float* x_out, float* y_out,
            Bump& saved, Bump& ws, bool dry, void* stream) {
  Ctx c{as_stream(stream), stream, &ws, dry, gemm_emu_mode()};
  Saved s;
  carve_saved(g, d->training ? saved : ws, s);
  const int E = g.E, F = g.F;
  const float *q = nullptr, *k = nullptr, *v = nullptr; int ldq = 0, ldkv = 0;
  const float* xq2 = x;
  void* aw = s.planes; long ab = s.planes_bytes;
  if (g.att == 2 && !aw) { ab = hoisdf_attention_emu_workspace(g.B, g.H, g.nq, g.S, 0); aw = ws.take(ab); }
  const bool mags = layer_mags(c, g) && (dry || s.mag);
  auto mg = [&](int i) -> uint32_t* { return mags && s.mag ? s.mag + (long)i * g.M : nullptr; };
  uint32_t* const smag = mags ? s.mag : nullptr;
  const uint32_t* xm = mags ? d->x_mag : nullptr;             // the caller's row magnitudes of x (B S words; null: the in-projection measures it)
  if (mags && !dry && c.ok() && hipMemsetAsync(s.mag, 0, (size_t)s.mag_bytes, c.st) != hipSuccess) c.rc = HOISDF_ERR_LAUNCH;
  if (g.fused_qkv) {
    // in-projection straight into the attention planes (same values as the f32 matrix + conversion pass: bit-identical planes)
    QkvPlanes pq{}, pkv{};
    const void* im = nullptr; const void* im_q = nullptr; const void* im_kv = nullptr;
    if (g.full) im = image_of(c, w->img_in, w->w_in, E, 3 * E, E, 0);
    else {
      if (!dry && c.ok()) c.rc = rows_copy_add(s.xq, (long)g.nq * E, x, (long)g.S * E, g.B, g.nq, E, 0, c.st);
      xq2 = s.xq;
      im_q = image_of(c, w->img_in_q, w->w_in, E, E, E, 0);
      im_kv = image_of(c, w->img_in_kv, w->w_in + (size_t)E * E, E, 2 * E, E, 0);
    }
    if (!dry && c.ok()) {
      if (!aw) c.rc = HOISDF_ERR_WORKSPACE;
      else {
        attention_emu_plane_targets(aw, g.B, g.H, g.nq, g.S, s.planes ? 1 : 0, pq, pkv);
        if (g.full) c.rc = linear_fwd_emu_qkv(x, E, im, w->b_in, g.Ms, 3 * E, E, pq, stream, xm);
        else {
          c.rc = linear_fwd_emu_qkv(s.xq, E, im_q, w->b_in, g.M, E, E, pq, stream, nullptr);  // (a row subset of x: measured)
          pkv.col0 = E;
          if (c.ok()) c.rc = linear_fwd_emu_qkv(x, E, im_kv, w->b_in ? w->b_in + E : nullptr, g.Ms, 2 * E, E, pkv, stream, xm);
        }
      }
    }
  } else if (g.full) {
    lin_fwd(c, x, E, w->w_in, E, w->img_in, w->b_in, s.qkv, 3 * E, g.Ms, 3 * E, E, 0, 0.f, 0, nullptr, 0, xm, nullptr,
            g.att_h2 ? head_q(g, smag) : nullptr, g.S);
    q = s.qkv; k = s.qkv + E; v = s.qkv + 2 * E; ldq = ldkv = 3 * E;
  } else {
    if (!dry && c.ok()) c.rc = rows_copy_add(s.xq, (long)g.nq * E, x, (long)g.S * E, g.B, g.nq, E, 0, c.st);
    xq2 = s.xq;
    lin_fwd(c, s.xq, E, w->w_in, E, w->img_in_q, w->b_in, s.qbuf, E, g.M, E, E, 0, 0.f, 0, nullptr, 0, nullptr, nullptr,
            g.att_h2 ? head_q(g, smag) : nullptr, g.nq);
    lin_fwd(c, x, E, w->w_in + (size_t)E * E, E, w->img_in_kv, w->b_in ? w->b_in + E : nullptr, s.kvbuf, 2 * E, g.Ms, 2 * E, E, 0, 0.f, 0, nullptr, 0, xm, nullptr,
            g.att_h2 ? head_k(g, smag) : nullptr, g.S);
    q = s.qbuf; k = s.kvbuf; v = s.kvbuf + E; ldq = E; ldkv = 2 * E;
  }
  const uint32_t* o_mag = g.att == 2 ? mg(MAG_O) : nullptr;        // (the exact-f32 attention leaves none: the out-projection measures o itself)
  if (g.fused_qkv) {
    if (!dry && c.ok())
      c.rc = attention_fwd_emu_planes(s.o, E, s.lse, g.B, g.H, g.nq, g.S, g.S, g.p, d->seed[0], aw, s.planes ? 1 : 0, stream, mg(MAG_O));
  } else if (g.att == 2) {
    if (!dry && c.ok()) {
      if (!aw) c.rc = HOISDF_ERR_WORKSPACE;
      else {
        const bool hm = g.att_h2 && smag;                                   // (without the head magnitudes: the bf16x3 form)
        c.rc = attention_fwd_emu_mag(q, ldq, k, ldkv, v, ldkv, s.o, E, s.lse, g.B, g.H, g.nq, g.S, g.S, g.p, d->seed[0], aw, ab,
                                     s.planes ? 1 : 0, mg(MAG_O), stream, hm ? head_q(g, smag) : nullptr, hm ? head_k(g, smag) : nullptr,
                                     hm ? head_v(g, smag) : nullptr);
      }
    }
  } else if (!dry && c.ok()) {
    c.rc = hoisdf_attention_fwd(q, ldq, k, ldkv, v, ldkv, s.o, E, s.lse, g.B, g.H, g.nq, g.S, g.S, g.p, d->seed[0], stream);
  }
  lin_fwd(c, s.o, E, w->w_out, E, w->img_out, w->b_out, s.a, E, g.M, E, E, 0, 0.f, 0, nullptr, 0, o_mag, nullptr);
  float* st = s.st;
  if (!dry && c.ok()) c.rc = add_layernorm_fwd_mag(xq2, s.a, w->g1, w->be1, s.x1, st, st + g.M, g.M, E, g.eps, g.p, d->seed[1], mg(MAG_X1), stream);
  lin_fwd(c, s.x1, E, w->w1, E, w->img_1, w->b1, s.h, F, g.M, F, E, 1, g.p, d->seed[2], s.bits, 0, mg(MAG_X1), mg(MAG_H));
  lin_fwd(c, s.h, F, w->w2, F, w->img_2, w->b2, s.f, E, g.M, E, F, 0, 0.f, 0, nullptr, 0, mg(MAG_H), nullptr);
  if (!dry && c.ok()) c.rc = add_layernorm_fwd_mag(s.x1, s.f, w->g2, w->be2, x_out, st + 2 * g.M, st + 3 * g.M, g.M, E, g.eps, g.p, d->seed[3], mg(MAG_XOUT), stream);
  if (w->g3 && !dry && c.ok()) {
    if (g.ni == g.nq) c.rc = hoisdf_add_layernorm_fwd(x_out, nullptr, w->g3, w->be3, y_out, st + 4 * g.M, st + 5 * g.M, g.M, E, g.eps, 0.f, 0, stream);
    else c.rc = hoisdf_layernorm_rows_fwd(x_out, w->g3, w->be3, y_out, st + 4 * g.M, st + 5 * g.M, g.B, g.nq, g.ni, E, g.eps, stream);
  }
  if (c.ok() && !dry && (ws.overflow || saved.overflow)) { set_error("encoder_layer_fwd: workspace / saved buffer too small"); return HOISDF_ERR_WORKSPACE; }
  return c.rc;
}

int backward(const float* x, const float* x_out, const hoisdf_encoder_layer_weights* w, const hoisdf_encoder_layer_desc* d, const Geo& g,
             Bump& saved, const float* g_x_out, const float* g_y, float* dx, const hoisdf_encoder_layer_grads* G, Bump& ws, bool dry,
             void* stream, bool fwd_mags = true) {
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
  // row magnitudes of df, dh, da (M words each), of [dq | dk | dv] (Ms words) and the head magnitudes of dO (H B words) (common.h): written
  // by the kernel that produces the matrix, read by the contraction that consumes it
  enum { MAG_DF = 0, MAG_DH = 1, MAG_DA = 2, MAG_BROWS = 3 };
  const long bmag_bytes = ((long)MAG_BROWS * M + Ms + (long)g.H * g.B) * 4;
  uint32_t* bmag = static_cast<uint32_t*>(ws.take(bmag_bytes));
  const bool mags = layer_mags(c, g) && (dry || bmag);
  auto mg = [&](int i) -> uint32_t* { return mags && bmag ? bmag + (long)i * M : nullptr; };
  uint32_t* const mag_dqkv = mags && bmag ? bmag + (long)MAG_BROWS * M : nullptr;
  uint32_t* const head_do = mags && bmag ? bmag + (long)MAG_BROWS * M + Ms : nullptr;
  auto fm = [&](int i) -> const uint32_t* { return mags && fwd_mags && s.mag ? s.mag + (long)i * M : nullptr; };       // the forward's (o, x1, h, x_out)
  uint32_t* const smag = mags && fwd_mags ? s.mag : nullptr;
  if (mags && !dry && c.ok() && hipMemsetAsync(bmag, 0, (size_t)bmag_bytes, c.st) != hipSuccess) c.rc = HOISDF_ERR_LAUNCH;
  if (!dry && c.ok()) c.rc = add_layernorm_bwd_mag(dx2, s.x1, s.f, w->g2, st + 2 * M, st + 3 * M, nullptr, dx1, df, G->dg2, G->dbe2, M, E, g.p, d->seed[3], nullptr, mg(MAG_DF), stream);
  lin_bwd_input(c, df, E, nullptr, 0.f, w->w2, F, w->img_t_2, dh, F, M, E, F, 0, mg(MAG_DF), mg(MAG_DH));
  lin_bwd_weight(c, df, E, nullptr, 0.f, s.h, F, G->dw2, G->db2, M, E, F, 0, mg(MAG_DF), fm(MAG_H));
  lin_bwd_input(c, dh, F, s.bits, g.p, w->w1, E, w->img_t_1, dx1, E, M, F, E, 1, mg(MAG_DH), nullptr);            // dx1 += : the FFN branch joins the residual
  lin_bwd_weight(c, dh, F, s.bits, g.p, s.x1, E, G->dw1, G->db1, M, F, E, 0, mg(MAG_DH), fm(MAG_X1));
  const float* xq2 = g.full ? x : s.xq;
  if (!dry && c.ok()) c.rc = add_layernorm_bwd_mag(dx1, xq2, s.a, w->g1, st, st + M, nullptr, dxq, da, G->dg1, G->dbe1, M, E, g.p, d->seed[1], nullptr, mg(MAG_DA), stream);
  lin_bwd_input(c, da, E, nullptr, 0.f, w->w_out, E, w->img_t_out, dO, E, M, E, E, 0, mg(MAG_DA), nullptr, g.att_h2 ? head_do : nullptr, g.nq);
  lin_bwd_weight(c, da, E, nullptr, 0.f, s.o, E, G->dw_out, G->db_out, M, E, E, 0, mg(MAG_DA), g.att == 2 ? fm(MAG_O) : nullptr);
  const bool amag = mags && g.att_bwd_emu && g.full;       // (separate q / kv matrices: the two kernels' words would have to be told apart)
  auto attn_bwd = [&](const float* q, int ldq, const float* k, const float* v, int ldkv, float* dq, float* dk, float* dv) {
    if (g.att_bwd_emu) {
      const long ab = hoisdf_attention_bwd_emu_workspace(g.B, g.H, g.nq, g.S, s.planes ? 1 : 0);
      void* aw = ws.take(ab);
      if (dry || !c.ok()) return;
      if (!aw) { c.rc = HOISDF_ERR_WORKSPACE; return; }
      // (f16x2 form: the head magnitudes of the projected matrices from the forward, dO's from the out-projection's grad-input)
      const bool hm = g.att_h2 && smag && head_do;
      if (g.att_h2 && !hm) { set_error("encoder_layer_bwd: the forward ran the f16x2 attention but the head magnitudes are not available"); c.rc = HOISDF_ERR_INVALID; return; }
      c.rc = attention_bwd_emu_mag(q, ldq, k, ldkv, v, ldkv, s.o, E, dO, E, s.lse, delta, dq, dk, dv, g.B, g.H, g.nq, g.S, g.S, g.p, d->seed[0],
                                   s.planes, aw, ab, amag ? mag_dqkv : nullptr, stream, hm ? head_q(g, smag) : nullptr,
                                   hm ? head_k(g, smag) : nullptr, hm ? head_v(g, smag) : nullptr, hm ? head_do : nullptr);
    } else if (!dry && c.ok()) {
      c.rc = hoisdf_attention_bwd(q, ldq, k, ldkv, v, ldkv, s.o, E, dO, E, s.lse, delta, dq, dk, dv, g.B, g.H, g.nq, g.S, g.S, g.p, d->seed[0], stream);
    }
  };
  if (g.full) {
    float* dqkv = ws.floats(Ms * 3 * E);
    if (!dry && !dqkv) { set_error("encoder_layer_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
    if (g.fused_qkv) attn_bwd(nullptr, 3 * E, nullptr, nullptr, 3 * E, dqkv, dqkv + E, dqkv + 2 * E);       // (the kept planes are the operands)
    else attn_bwd(s.qkv, 3 * E, s.qkv + E, s.qkv + 2 * E, 3 * E, dqkv, dqkv + E, dqkv + 2 * E);
    lin_bwd_input(c, dqkv, 3 * E, nullptr, 0.f, w->w_in, E, w->img_t_in, dxq, E, Ms, 3 * E, E, 1, amag ? mag_dqkv : nullptr, nullptr);   // += : attention branch joins the residual
    lin_bwd_weight(c, dqkv, 3 * E, nullptr, 0.f, x, E, G->dw_in, G->db_in, Ms, 3 * E, E, 0, amag ? mag_dqkv : nullptr, amag ? d->x_mag : nullptr);
  } else {
    float* dq = ws.floats(M * E); float* dkv = ws.floats(Ms * 2 * E);
    if (!dry && (!dq || !dkv)) { set_error("encoder_layer_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
    if (g.fused_qkv) attn_bwd(nullptr, E, nullptr, nullptr, 2 * E, dq, dkv, dkv + E);
    else attn_bwd(s.qbuf, E, s.kvbuf, s.kvbuf + E, 2 * E, dq, dkv, dkv + E);
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

extern "C" const uint32_t* hoisdf_encoder_layer_out_mag(const hoisdf_encoder_layer_desc* d, const void* saved) {
  Geo g;
  if (!saved || geometry(d, g) != HOISDF_OK || !d->training) return nullptr;
  Ctx c{nullptr, nullptr, nullptr, false, gemm_emu_mode()};
  if (!layer_mags(c, g)) return nullptr;
  Bump b(const_cast<void*>(saved), 1L << 62); Saved s;
  carve_saved(g, b, s);
  return s.mag ? s.mag + (long)MAG_XOUT * g.M : nullptr;
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
  if (d->training) {
    Ctx c0{nullptr, nullptr, nullptr, false, gemm_emu_mode()};
    saved_forms().put(saved, (g.fused_qkv ? 1 : 0) | (g.att_h2 ? 2 : 0) | (layer_mags(c0, g) ? 4 : 0));     // (4: row / head magnitudes were left in `saved`)
  }
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
  const int recorded = saved_forms().take(saved);              // what the forward of THIS saved buffer decided (-1: unknown host, recompute)
  if (recorded >= 0) { g.fused_qkv = (recorded & 1) != 0; g.att_h2 = (recorded & 2) != 0; }
  const bool fwd_mags = recorded < 0 || (recorded & 4) != 0;   // (a forward that ran with the emulation off left no magnitudes to trust)
  Bump sv(const_cast<void*>(saved), saved_bytes), ws(workspace, workspace_bytes);
  const int rc = backward(x, x_out, w, d, g, sv, g_x_out, g_y, dx, grads, ws, false, stream, fwd_mags);
  if (rc == HOISDF_OK && sv.overflow) { set_error("encoder_layer_bwd: saved buffer too small"); return HOISDF_ERR_WORKSPACE; }
  if (rc == HOISDF_ERR_WORKSPACE) set_error("encoder_layer_bwd: workspace (%ld bytes) too small", workspace_bytes);
  return rc;
}

// ============================================================================================================================
// one transformer DECODER layer per call (common/nets/transformer.py:366-395, TransformerDecoderLayer.forward_post as the hand
// stack uses it: Q = 17 MANO queries per sample, masked self-attention over the queries with q = k = tgt + query_pos, v = tgt;
// cross-attention of tgt + query_pos over the first kv_len rows of the encoder memory; FFN; three post-norms; + the stack's norm
// of the layer output, :150-163).  The per-op entries in the order nets/blocks.py issues them.
// ============================================================================================================================
namespace hoisdf {
namespace {

// out[b][q][:] = a[b][q][:] + p[q][:]
__global__ void add_bcast_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ p, int rows_p, int E4, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long per = (long)rows_p * E4;
  const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(p)[i % per];
  reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}
// dp[q][:] += sum_b (g1[b][q][:] + g2[b][q][:])   (one thread per (q, 4 columns), samples in order: fixed summation order)
__global__ void sum_bcast_grad_kernel(float* __restrict__ dp, const float* __restrict__ g1, const float* __restrict__ g2, int B, int rows_p, int E4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_p * E4) return;
  float4 acc = reinterpret_cast<float4*>(dp)[i];
  for (int b = 0; b < B; ++b) {
    const float4 x = reinterpret_cast<const float4*>(g1)[(long)b * rows_p * E4 + i], y = reinterpret_cast<const float4*>(g2)[(long)b * rows_p * E4 + i];
    acc.x += x.x + y.x; acc.y += x.y + y.y; acc.z += x.z + y.z; acc.w += x.w + y.w;
  }
  reinterpret_cast<float4*>(dp)[i] = acc;
}

struct DGeo { int B, Q, S, E, F, H, kv; long M, Ms; float eps, p; };
int dgeometry(const hoisdf_decoder_layer_desc* d, DGeo& g) {
  HOISDF_REQUIRE(d, HOISDF_ERR_INVALID, "decoder_layer: null descriptor");
  HOISDF_REQUIRE(d->B > 0 && d->Q > 0 && d->Q <= 64 && d->S > 0 && d->E > 0 && d->F > 0 && d->H > 0 && d->E % d->H == 0 && d->E % 4 == 0 &&
                     d->F % 4 == 0 && d->kv_len > 0 && d->kv_len <= d->S,
                 HOISDF_ERR_INVALID, "decoder_layer: bad sizes B=%d Q=%d S=%d E=%d F=%d H=%d kv_len=%d", d->B, d->Q, d->S, d->E, d->F, d->H, d->kv_len);
  HOISDF_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, HOISDF_ERR_INVALID, "decoder_layer: drop_p=%f", d->drop_p);
  g.B = d->B; g.Q = d->Q; g.S = d->S; g.E = d->E; g.F = d->F; g.H = d->H; g.kv = d->kv_len;
  g.M = (long)d->B * d->Q; g.Ms = (long)d->B * d->S; g.eps = d->eps; g.p = d->drop_p;
  return HOISDF_OK;
}
struct DSaved {
  float *tq1, *qk, *v, *probs, *o1, *a1, *x1, *tq2, *q2, *kv, *o2, *lse2, *a2, *x2, *h, *f, *st; uint32_t* bits;
};
void dcarve(const DGeo& g, Bump& b, DSaved& s) {
  const long M = g.M; const int E = g.E;
  s.tq1 = b.floats(M * E); s.qk = b.floats(M * 2 * E); s.v = b.floats(M * E); s.probs = b.floats((long)g.B * g.H * g.Q * g.Q);
  s.o1 = b.floats(M * E); s.a1 = b.floats(M * E); s.x1 = b.floats(M * E); s.tq2 = b.floats(M * E); s.q2 = b.floats(M * E);
  s.kv = b.floats(g.Ms * 2 * E); s.o2 = b.floats(M * E); s.lse2 = b.floats((long)g.B * g.H * g.Q); s.a2 = b.floats(M * E);
  s.x2 = b.floats(M * E); s.h = b.floats(M * g.F); s.bits = static_cast<uint32_t*>(b.take(M * ((g.F + 31) / 32) * 4)); s.f = b.floats(M * E);
  s.st = b.floats(8 * M);
}
int add_bcast(float* out, const float* a, const float* p, const DGeo& g, hipStream_t st) {
  const long total = g.M * (g.E / 4);
  hipLaunchKernelGGL(add_bcast_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, out, a, p, g.Q, g.E / 4, total);
  return check_launch("decoder_layer add");
}

int dforward(const float* tgt, const float* memory, const float* qpos, const uint8_t* mask, const hoisdf_decoder_layer_weights* w,
             const hoisdf_decoder_layer_desc* d, const DGeo& g, float* out, float* y_out, Bump& saved, Bump& ws, bool dry, void* stream) {
  Ctx c{as_stream(stream), stream, &ws, dry, gemm_emu_mode()};
  DSaved s;
  dcarve(g, d->training ? saved : ws, s);
  const int E = g.E, F = g.F;
  const long M = g.M;
  float* st = s.st;
  // masked self-attention over the queries
  if (!dry && c.ok()) c.rc = add_bcast(s.tq1, tgt, qpos, g, c.st);
  lin_fwd(c, s.tq1, E, w->sa_w_in, E, nullptr, w->sa_b_in, s.qk, 2 * E, M, 2 * E, E, 0, 0.f, 0, nullptr);
  lin_fwd(c, tgt, E, w->sa_w_in + (size_t)2 * E * E, E, nullptr, w->sa_b_in ? w->sa_b_in + 2 * E : nullptr, s.v, E, M, E, E, 0, 0.f, 0, nullptr);
  if (!dry && c.ok())
    c.rc = hoisdf_attention_small_fwd(s.qk, 2 * E, s.qk + E, 2 * E, s.v, E, mask, s.o1, E, s.probs, g.B, g.H, g.Q, g.Q, g.p, d->seed[0], stream);
  lin_fwd(c, s.o1, E, w->sa_w_out, E, nullptr, w->sa_b_out, s.a1, E, M, E, E, 0, 0.f, 0, nullptr);
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_fwd(tgt, s.a1, w->g1, w->be1, s.x1, st, st + M, M, E, g.eps, g.p, d->seed[1], stream);
  // cross-attention over the encoder memory (keys < kv_len)
  if (!dry && c.ok()) c.rc = add_bcast(s.tq2, s.x1, qpos, g, c.st);
  lin_fwd(c, s.tq2, E, w->ca_w_in, E, nullptr, w->ca_b_in, s.q2, E, M, E, E, 0, 0.f, 0, nullptr);
  lin_fwd(c, memory, E, w->ca_w_in + (size_t)E * E, E, w->img_ca_kv, w->ca_b_in ? w->ca_b_in + E : nullptr, s.kv, 2 * E, g.Ms, 2 * E, E, 0, 0.f, 0, nullptr);
  if (!dry && c.ok())
    c.rc = hoisdf_attention_fwd(s.q2, E, s.kv, 2 * E, s.kv + E, 2 * E, s.o2, E, s.lse2, g.B, g.H, g.Q, g.S, g.kv, g.p, d->seed[2], stream);
  lin_fwd(c, s.o2, E, w->ca_w_out, E, nullptr, w->ca_b_out, s.a2, E, M, E, E, 0, 0.f, 0, nullptr);
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_fwd(s.x1, s.a2, w->g2, w->be2, s.x2, st + 2 * M, st + 3 * M, M, E, g.eps, g.p, d->seed[3], stream);
  // FFN
  lin_fwd(c, s.x2, E, w->w1, E, nullptr, w->b1, s.h, F, M, F, E, 1, g.p, d->seed[4], s.bits);
  lin_fwd(c, s.h, F, w->w2, F, nullptr, w->b2, s.f, E, M, E, F, 0, 0.f, 0, nullptr);
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_fwd(s.x2, s.f, w->g3, w->be3, out, st + 4 * M, st + 5 * M, M, E, g.eps, g.p, d->seed[5], stream);
  if (w->g4 && !dry && c.ok()) c.rc = hoisdf_add_layernorm_fwd(out, nullptr, w->g4, w->be4, y_out, st + 6 * M, st + 7 * M, M, E, g.eps, 0.f, 0, stream);
  return c.rc;
}

int dbackward(const float* tgt, const float* memory, const uint8_t* mask, const float* out, const hoisdf_decoder_layer_weights* w,
              const hoisdf_decoder_layer_desc* d, const DGeo& g, Bump& saved, const float* g_out, const float* g_y, float* d_tgt, float* d_memory,
              int accumulate_memory, float* d_qpos, const hoisdf_decoder_layer_grads* G, Bump& ws, bool dry, void* stream) {
  Ctx c{as_stream(stream), stream, &ws, dry, gemm_emu_mode()};
  DSaved s;
  dcarve(g, saved, s);
  (void)mask;
  const int E = g.E, F = g.F;
  const long M = g.M, Ms = g.Ms;
  const float* st = s.st;
  const float* dx3 = g_out;
  if (g_y && w->g4) {
    float* t = ws.floats(M * E);
    if (!dry && c.ok()) {
      if (!t) c.rc = HOISDF_ERR_WORKSPACE;
      else c.rc = hoisdf_add_layernorm_bwd(g_y, out, nullptr, w->g4, st + 6 * M, st + 7 * M, g_out, t, nullptr, G->dg4, G->dbe4, M, E, 0.f, 0, stream);
    }
    dx3 = t;
  }
  float* dx2 = ws.floats(M * E); float* df = ws.floats(M * E); float* dh = ws.floats(M * F);
  float* dx1 = ws.floats(M * E); float* da2 = ws.floats(M * E); float* do2 = ws.floats(M * E); float* dq2 = ws.floats(M * E);
  float* dkv = ws.floats(Ms * 2 * E); float* delta = ws.floats((long)g.B * g.H * g.Q); float* dtq2 = ws.floats(M * E);
  float* da1 = ws.floats(M * E); float* do1 = ws.floats(M * E); float* dqk = ws.floats(M * 2 * E); float* dv = ws.floats(M * E);
  float* dtq1 = ws.floats(M * E);
  if (!dry && (!dx2 || !df || !dh || !dx1 || !da2 || !do2 || !dq2 || !dkv || !delta || !dtq2 || !da1 || !do1 || !dqk || !dv || !dtq1)) {
    set_error("decoder_layer_bwd: workspace too small");
    return HOISDF_ERR_WORKSPACE;
  }
  if (!dry && !dx3) { set_error("decoder_layer_bwd: no upstream gradient (g_out and g_y both null)"); return HOISDF_ERR_INVALID; }
  // FFN + norm3
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_bwd(dx3, s.x2, s.f, w->g3, st + 4 * M, st + 5 * M, nullptr, dx2, df, G->dg3, G->dbe3, M, E, g.p, d->seed[5], stream);
  lin_bwd_input(c, df, E, nullptr, 0.f, w->w2, F, nullptr, dh, F, M, E, F, 0);
  lin_bwd_weight(c, df, E, nullptr, 0.f, s.h, F, G->dw2, G->db2, M, E, F);
  lin_bwd_input(c, dh, F, s.bits, g.p, w->w1, E, nullptr, dx2, E, M, F, E, 1);
  lin_bwd_weight(c, dh, F, s.bits, g.p, s.x2, E, G->dw1, G->db1, M, F, E);
  // cross-attention + norm2
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_bwd(dx2, s.x1, s.a2, w->g2, st + 2 * M, st + 3 * M, nullptr, dx1, da2, G->dg2, G->dbe2, M, E, g.p, d->seed[3], stream);
  lin_bwd_input(c, da2, E, nullptr, 0.f, w->ca_w_out, E, nullptr, do2, E, M, E, E, 0);
  lin_bwd_weight(c, da2, E, nullptr, 0.f, s.o2, E, G->dca_w_out, G->dca_b_out, M, E, E);
  if (!dry && c.ok())
    c.rc = hoisdf_attention_bwd(s.q2, E, s.kv, 2 * E, s.kv + E, 2 * E, s.o2, E, do2, E, s.lse2, delta, dq2, dkv, dkv + E, g.B, g.H, g.Q, g.S, g.kv, g.p,
                                d->seed[2], stream);
  lin_bwd_input(c, dq2, E, nullptr, 0.f, w->ca_w_in, E, nullptr, dtq2, E, M, E, E, 0);
  lin_bwd_weight(c, dq2, E, nullptr, 0.f, s.tq2, E, G->dca_w_in, G->dca_b_in, M, E, E);
  lin_bwd_input(c, dkv, 2 * E, nullptr, 0.f, w->ca_w_in + (size_t)E * E, E, w->img_t_ca_kv, d_memory, E, Ms, 2 * E, E, accumulate_memory ? 1 : 0);
  lin_bwd_weight(c, dkv, 2 * E, nullptr, 0.f, memory, E, G->dca_w_in + (size_t)E * E, G->dca_b_in ? G->dca_b_in + E : nullptr, Ms, 2 * E, E);
  if (!dry && c.ok()) c.rc = rows_copy_add(dx1, M * E, dtq2, M * E, 1, (int)M, E, 1, c.st);          // x1 also fed tq2 = x1 + query_pos
  // self-attention + norm1
  if (!dry && c.ok()) c.rc = hoisdf_add_layernorm_bwd(dx1, tgt, s.a1, w->g1, st, st + M, nullptr, d_tgt, da1, G->dg1, G->dbe1, M, E, g.p, d->seed[1], stream);
  lin_bwd_input(c, da1, E, nullptr, 0.f, w->sa_w_out, E, nullptr, do1, E, M, E, E, 0);
  lin_bwd_weight(c, da1, E, nullptr, 0.f, s.o1, E, G->dsa_w_out, G->dsa_b_out, M, E, E);
  if (!dry && c.ok()) {
    // (the small backward accumulates dK / dV over the queries with atomics: zeroed here)
    if (hipMemsetAsync(dqk, 0, sizeof(float) * M * 2 * E, c.st) != hipSuccess || hipMemsetAsync(dv, 0, sizeof(float) * M * E, c.st) != hipSuccess) {
      set_error("decoder_layer_bwd: clearing the self-attention gradients failed");
      return HOISDF_ERR_LAUNCH;
    }
    c.rc = hoisdf_attention_small_bwd(s.qk, 2 * E, s.qk + E, 2 * E, s.v, E, s.probs, do1, E, dqk, dqk + E, dv, g.B, g.H, g.Q, g.Q, g.p, d->seed[0], stream);
  }
  lin_bwd_input(c, dqk, 2 * E, nullptr, 0.f, w->sa_w_in, E, nullptr, dtq1, E, M, 2 * E, E, 0);
  lin_bwd_weight(c, dqk, 2 * E, nullptr, 0.f, s.tq1, E, G->dsa_w_in, G->dsa_b_in, M, 2 * E, E);
  lin_bwd_input(c, dv, E, nullptr, 0.f, w->sa_w_in + (size_t)2 * E * E, E, nullptr, d_tgt, E, M, E, E, 1);
  lin_bwd_weight(c, dv, E, nullptr, 0.f, tgt, E, G->dsa_w_in + (size_t)2 * E * E, G->dsa_b_in ? G->dsa_b_in + 2 * E : nullptr, M, E, E);
  if (!dry && c.ok()) c.rc = rows_copy_add(d_tgt, M * E, dtq1, M * E, 1, (int)M, E, 1, c.st);        // tgt also fed tq1 = tgt + query_pos
  if (d_qpos && !dry && c.ok()) {
    hipLaunchKernelGGL(sum_bcast_grad_kernel, dim3((unsigned)cdiv((long)g.Q * (E / 4), 256)), dim3(256), 0, c.st, d_qpos, dtq1, dtq2, g.B, g.Q, E / 4);
    c.rc = check_launch("decoder_layer query_pos gradient");
  }
  if (c.ok() && !dry && ws.overflow) { set_error("decoder_layer_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
  return c.rc;
}

}  // namespace
}  // namespace hoisdf

extern "C" long hoisdf_decoder_layer_saved_bytes(const hoisdf_decoder_layer_desc* d) {
  DGeo g;
  if (dgeometry(d, g) != HOISDF_OK) return -1;
  Bump b(nullptr, 0); DSaved s;
  dcarve(g, b, s);
  return b.off + 256;
}

extern "C" long hoisdf_decoder_layer_workspace_bytes(const hoisdf_decoder_layer_desc* d, int backward_pass) {
  DGeo g;
  if (dgeometry(d, g) != HOISDF_OK) return -1;
  hoisdf_decoder_layer_weights w{};
  hoisdf_decoder_layer_grads G{};
  float dummy = 0.f;
  w.g4 = &dummy;
  Bump saved(nullptr, 0), ws(nullptr, 0);
  if (backward_pass) (void)dbackward(nullptr, nullptr, nullptr, nullptr, &w, d, g, saved, &dummy, &dummy, nullptr, nullptr, 0, nullptr, &G, ws, true, nullptr);
  else (void)dforward(nullptr, nullptr, nullptr, nullptr, &w, d, g, nullptr, nullptr, saved, ws, true, nullptr);
  return ws.off + 256;
}

extern "C" int hoisdf_decoder_layer_fwd(const float* tgt, const float* memory, const float* query_pos, const uint8_t* tgt_mask,
                                        const hoisdf_decoder_layer_weights* w, const hoisdf_decoder_layer_desc* d, float* out, float* y_out,
                                        void* saved, long saved_bytes, void* workspace, long workspace_bytes, void* stream) {
  DGeo g;
  if (int rc = dgeometry(d, g)) return rc;
  HOISDF_REQUIRE(tgt && memory && query_pos && tgt_mask && w && out && w->sa_w_in && w->sa_w_out && w->ca_w_in && w->ca_w_out && w->w1 && w->w2 &&
                     w->g1 && w->be1 && w->g2 && w->be2 && w->g3 && w->be3,
                 HOISDF_ERR_INVALID, "decoder_layer_fwd: null pointer");
  HOISDF_REQUIRE(!w->g4 || (w->be4 && y_out), HOISDF_ERR_INVALID, "decoder_layer_fwd: the stack norm needs be4 and y_out");
  HOISDF_REQUIRE(!d->training || saved, HOISDF_ERR_WORKSPACE, "decoder_layer_fwd: training needs the saved buffer (hoisdf_decoder_layer_saved_bytes)");
  HOISDF_REQUIRE(al16(tgt) && al16(memory) && al16(query_pos) && al16(out) && al16(saved) && al16(workspace), HOISDF_ERR_INVALID,
                 "decoder_layer_fwd: buffers must be 16-byte aligned");
  static char none;
  Bump sv(saved, saved_bytes), ws(workspace, workspace_bytes);
  if (!sv.base) { sv.base = &none; sv.cap = 0; }
  if (!ws.base) { ws.base = &none; ws.cap = 0; }
  int rc = dforward(tgt, memory, query_pos, tgt_mask, w, d, g, out, y_out, sv, ws, false, stream);
  if (rc == HOISDF_OK && (sv.overflow || ws.overflow)) rc = HOISDF_ERR_WORKSPACE;
  if (rc == HOISDF_ERR_WORKSPACE) set_error("decoder_layer_fwd: workspace (%ld bytes) or saved buffer (%ld bytes) too small", workspace_bytes, saved_bytes);
  return rc;
}

extern "C" int hoisdf_decoder_layer_bwd(const float* tgt, const float* memory, const uint8_t* tgt_mask, const float* out,
                                        const hoisdf_decoder_layer_weights* w, const hoisdf_decoder_layer_desc* d, const void* saved, long saved_bytes,
                                        const float* g_out, const float* g_y, float* d_tgt, float* d_memory, int accumulate_memory,
                                        float* d_query_pos, const hoisdf_decoder_layer_grads* grads, void* workspace, long workspace_bytes,
                                        void* stream) {
  DGeo g;
  if (int rc = dgeometry(d, g)) return rc;
  HOISDF_REQUIRE(tgt && memory && out && w && saved && d_tgt && d_memory && grads && workspace, HOISDF_ERR_INVALID, "decoder_layer_bwd: null pointer");
  HOISDF_REQUIRE(grads->dsa_w_in && grads->dsa_b_in && grads->dsa_w_out && grads->dsa_b_out && grads->dca_w_in && grads->dca_b_in && grads->dca_w_out &&
                     grads->dca_b_out && grads->dw1 && grads->db1 && grads->dw2 && grads->db2 && grads->dg1 && grads->dbe1 && grads->dg2 &&
                     grads->dbe2 && grads->dg3 && grads->dbe3 && (!w->g4 || !g_y || (grads->dg4 && grads->dbe4)),
                 HOISDF_ERR_INVALID, "decoder_layer_bwd: every parameter gradient buffer is required (zero-filled)");
  HOISDF_REQUIRE(d->training, HOISDF_ERR_INVALID, "decoder_layer_bwd: the forward call must have run with training = 1");
  HOISDF_REQUIRE(al16(tgt) && al16(memory) && al16(d_tgt) && al16(d_memory) && al16(saved) && al16(workspace) && al16(d_query_pos), HOISDF_ERR_INVALID,
                 "decoder_layer_bwd: buffers must be 16-byte aligned");
  Bump sv(const_cast<void*>(saved), saved_bytes), ws(workspace, workspace_bytes);
  int rc = dbackward(tgt, memory, tgt_mask, out, w, d, g, sv, g_out, g_y, d_tgt, d_memory, accumulate_memory, d_query_pos, grads, ws, false, stream);
  if (rc == HOISDF_OK && sv.overflow) rc = HOISDF_ERR_WORKSPACE;
  if (rc == HOISDF_ERR_WORKSPACE) set_error("decoder_layer_bwd: workspace (%ld bytes) or saved buffer too small", workspace_bytes);
  return rc;
}
