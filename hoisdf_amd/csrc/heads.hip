// Coarse C entries of SURVEY.md section 8(b) for the two MLP-shaped stages either side of the transformers:
//   hoisdf_tokens_fwd / _bwd      K1 + K7 + K8: [project + gather ->] linear_transformerin (C -> 1024 -> 512 -> 256 -> D - 33,
//                                 ReLU after all four, main/model.py:58-62,145-179) -> sigma gate + token rows (main/model.py:
//                                 123-126,520-562) in one call per direction;
//   hoisdf_heads_vote_fwd / _bwd  K11 + K12: linear_handvote (D -> D -> D -> D -> 3 J) and linear_handcls (D -> D -> D -> J) on the
//                                 intermediate hand rows of ALL encoder depths (main/model.py:587-593) + the vote aggregation and
//                                 the JointvoteLoss reductions (common/nets/loss.py:31-61) in one call per direction.
// Host-side chains of this library's launches on the caller's stream over caller-provided `saved` / `workspace` buffers (the size
// queries run the same carving code without launching), linear layers by the library defaults (fp32 emulated on the bf16 pipe from
// 2048 rows up).  hoisdf_amd/ops.py's tokens / heads_vote autograd nodes are thin wrappers.
#include "chain.h"

using namespace hoisdf;

namespace {
constexpr int MLP_MAX = HOISDF_MLP_MAX_LAYERS;
inline long bits_words(int N) { return (N + 31) / 32; }

struct MlpSaved { float* h[MLP_MAX]; uint32_t* bits[MLP_MAX]; uint32_t* mag; };

bool mlp_ok(const hoisdf_mlp* m) {
  if (!m || m->n_layers < 1 || m->n_layers > MLP_MAX) return false;
  for (int i = 0; i <= m->n_layers; ++i) if (m->dims[i] <= 0) return false;
  return true;
}
// hidden activations h[i] (output of layer i, i < last) and the ReLU sign maps of every activated layer; magnitude words (common.h)
// of the operands of the chain's contractions: array 0 = the input x (when the chain measured it itself), i = h[i - 1]
void mlp_carve(const hoisdf_mlp* m, long M, Bump& b, MlpSaved& s) {
  const int last = m->n_layers - 1;
  for (int i = 0; i < m->n_layers; ++i) {
    s.h[i] = i < last ? b.floats(M * m->dims[i + 1]) : nullptr;
    const bool act = i < last || m->act_last;
    s.bits[i] = act ? static_cast<uint32_t*>(b.take(M * bits_words(m->dims[i + 1]) * 4)) : nullptr;
  }
  s.mag = static_cast<uint32_t*>(b.take((long)m->n_layers * M * 4));       // (row magnitudes: M words per array)
}
inline bool chain_mags_on(const Ctx& c, long M) { return c.emu && emu_form_h2() && M >= EMU_MIN_ROWS; }
// n zero-filled row-magnitude arrays (M words each) from the workspace (null: not in the f16x2 form; a measuring pass only reserves the bytes)
uint32_t* chain_mags(Ctx& c, long M, int n) {
  if (!chain_mags_on(c, M)) return nullptr;
  uint32_t* p = static_cast<uint32_t*>(c.ws->take((long)n * M * 4));
  if (c.dry || !p || !c.ok()) return nullptr;
  if (hipMemsetAsync(p, 0, (size_t)n * M * 4, c.st) != hipSuccess) { c.rc = HOISDF_ERR_LAUNCH; return nullptr; }
  return p;
}
// x_mag: the caller's magnitude words of x; null = measured here ONCE (array 0 of s.mag) for the layer-0 contraction and, in the
// backward, its grad-weight.  The backward must be given the same x_mag.
void mlp_forward(Ctx& c, const hoisdf_mlp* m, const float* x, int ldx, long M, const MlpSaved& s, float* y, int ldy, const uint32_t* x_mag = nullptr) {
  const int last = m->n_layers - 1;
  const float* in = x; int ldin = ldx;
  uint32_t* mg = chain_mags_on(c, M) && !c.dry && s.mag && c.ok() ? s.mag : nullptr;
  if (!c.dry) saved_mags().put(s.mag, mg != nullptr);            // (what the backward of this block may rely on: chain.h)
  if (mg && hipMemsetAsync(mg, 0, (size_t)m->n_layers * M * 4, c.st) != hipSuccess) { c.rc = HOISDF_ERR_LAUNCH; return; }
  const uint32_t* in_mag = mg ? x_mag : nullptr;
  if (mg && !in_mag && emu_rows(c, M, x, ldx, m->dims[0])) {
    if ((c.rc = emu_mag_measure(x, ldx, M, m->dims[0], mg, c.st)) != HOISDF_OK) return;
    in_mag = mg;
  }
  for (int i = 0; i < m->n_layers; ++i) {
    float* out = i < last ? s.h[i] : y;
    const int ldo = i < last ? m->dims[i + 1] : ldy;
    // (the tiled emulated form is the one that writes the words: the same test lin_fwd makes)
    uint32_t* out_mag = mg && i < last && emu_rows(c, M, in, ldin, m->dims[i]) ? mg + (long)(i + 1) * M : nullptr;
    lin_fwd(c, in, ldin, m->w[i], m->dims[i], m->img[i], m->b[i], out, ldo, M, m->dims[i + 1], m->dims[i], s.bits[i] ? 1 : 0, 0.f, 0, s.bits[i], 0,
            in_mag, out_mag);
    in = out; ldin = ldo; in_mag = out_mag;
  }
}
// dy [M][ld] = gradient of the last layer's (post-activation) output; dx (optional) receives / accumulates the input gradient
void mlp_backward(Ctx& c, const hoisdf_mlp* m, const hoisdf_mlp_grads* G, const float* x, int ldx, long M, const MlpSaved& s, const float* dy,
                  int lddy, float* dx, int lddx, int accumulate_dx, const uint32_t* x_mag = nullptr) {
  const float* g = dy; int ldg = lddy;
  const int last = m->n_layers - 1;
  uint32_t* mg = chain_mags(c, M, m->n_layers + 1);         // array i = the gradient entering layer i (i = n_layers: dy)
  const bool fwd_mags = mg && s.mag && saved_mags().get(s.mag) != 0;      // (0: the forward of this block ran without them)
  const uint32_t* g_mag = nullptr;
  if (mg && emu_rows(c, M, dy, lddy, m->dims[last + 1])) {   // dy feeds two contractions: measured once
    if ((c.rc = emu_mag_measure(dy, lddy, M, m->dims[last + 1], mg + (long)(last + 1) * M, c.st)) != HOISDF_OK) return;
    g_mag = mg + (long)(last + 1) * M;
  }
  for (int i = last; i >= 0; --i) {
    const float* in = i > 0 ? s.h[i - 1] : x;
    const int ldin = i > 0 ? m->dims[i] : ldx;
    // the words of the layer's input: h[i - 1]'s from the forward's epilogue; x's from the caller or from the forward's own pass
    const uint32_t* in_mag = !fwd_mags ? nullptr : i > 0 ? (emu_rows(c, M, i > 1 ? s.h[i - 2] : x, i > 1 ? m->dims[i - 1] : ldx, m->dims[i - 1]) ? s.mag + (long)i * M : nullptr)
                                                         : (x_mag ? x_mag : (emu_rows(c, M, x, ldx, m->dims[0]) ? s.mag : nullptr));
    lin_bwd_weight(c, g, ldg, s.bits[i], 0.f, in, ldin, G->dw[i], G->db[i], M, m->dims[i + 1], m->dims[i], 0, g_mag, g_mag ? in_mag : nullptr);
    if (i == 0 && !dx) break;
    float* gin = i > 0 ? c.ws->floats(M * m->dims[i]) : dx;
    const int ldgin = i > 0 ? m->dims[i] : lddx;
    if (!c.dry && c.ok() && !gin) { c.rc = HOISDF_ERR_WORKSPACE; return; }
    uint32_t* gin_mag = mg && i > 0 && emu_rows(c, M, g, ldg, m->dims[i + 1]) ? mg + (long)i * M : nullptr;
    lin_bwd_input(c, g, ldg, s.bits[i], 0.f, m->w[i], m->dims[i], m->img_t[i], gin, ldgin, M, m->dims[i + 1], m->dims[i], i == 0 ? accumulate_dx : 0,
                  g_mag, gin_mag);
    g = gin; ldg = ldgin; g_mag = gin_mag;
  }
}
// the words mlp_forward(m, x) left for x in its saved block (null when it did not measure)
const uint32_t* enc_mag(const Ctx& c, const hoisdf_mlp* m, const float* x, int ldx, long M, const MlpSaved& s) {
  return chain_mags_on(c, M) && !c.dry && s.mag && saved_mags().get(s.mag) != 0 && emu_rows(c, M, x, ldx, m->dims[0]) ? s.mag : nullptr;
}
bool grads_ok(const hoisdf_mlp* m, const hoisdf_mlp_grads* G) {
  if (!G) return false;
  for (int i = 0; i < m->n_layers; ++i) if (!G->dw[i] || !G->db[i]) return false;
  return true;
}

// ---------------------------------------------------------------- tokens ----------------------------------------------------------------
struct TokSaved { float* feat; float* fea; MlpSaved mlp; };
void tok_carve(const hoisdf_mlp* m, long M, bool own_feat, Bump& b, TokSaved& s) {
  s.feat = own_feat ? b.floats(M * m->dims[0]) : nullptr;
  s.fea = b.floats(M * m->dims[m->n_layers]);
  mlp_carve(m, M, b, s.mlp);
}

// ---------------------------------------------------------------- heads + vote ----------------------------------------------------------
struct VoteSaved { float *off, *cls, *stats; MlpSaved vote, cl; };
void vote_carve(const hoisdf_mlp* mv, const hoisdf_mlp* mc, long M, int LB, int J, Bump& b, VoteSaved& s) {
  s.off = b.floats(M * 3 * J); s.cls = b.floats(M * J); s.stats = b.floats((long)LB * J * 2);
  mlp_carve(mv, M, b, s.vote); mlp_carve(mc, M, b, s.cl);
}
}  // namespace

extern "C" long hoisdf_tokens_saved_bytes(const hoisdf_mlp* mlp, long n_rows, int gather_inside) {
  if (!mlp_ok(mlp) || n_rows <= 0) return 0;
  Bump b(nullptr, 0); TokSaved s;
  tok_carve(mlp, n_rows, gather_inside != 0, b, s);
  return b.off + 256;
}

static int tokens_backward(const hoisdf_mlp* mlp, const hoisdf_mlp_grads* G, const float* feat, long M, TokSaved& s, const float* dtok,
                           const float* sdf, const float* beta, float* dfeat, int accumulate_dfeat, float* dbeta, int B, int P, int S, int row0,
                           int D, Bump& ws, bool dry, void* stream) {
  Ctx c{as_stream(stream), stream, &ws, dry, gemm_emu_mode()};
  const int F = mlp->dims[mlp->n_layers];
  float* dfea = ws.floats(M * F);
  float* part = ws.floats(hoisdf_token_build_bwd_partials());
  if (!dry) {
    if (!dfea || !part) { set_error("tokens_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
    c.rc = hoisdf_token_build_bwd_ordered(dtok, s.fea, F, sdf, beta, dfea, F, dbeta, part, B, P, S, row0, D, stream);
  }
  mlp_backward(c, mlp, G, feat, mlp->dims[0], M, s.mlp, dfea, F, dfeat, mlp->dims[0], accumulate_dfeat);
  if (c.ok() && !dry && ws.overflow) { set_error("tokens_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
  return c.rc;
}

extern "C" long hoisdf_tokens_workspace_bytes(const hoisdf_mlp* mlp, long n_rows, int backward_pass) {
  if (!mlp_ok(mlp) || n_rows <= 0) return 0;
  Bump ws(nullptr, 0), sv(nullptr, 0);
  TokSaved s; tok_carve(mlp, n_rows, false, sv, s);
  Ctx c{nullptr, nullptr, &ws, true, gemm_emu_mode()};
  if (backward_pass) {
    hoisdf_mlp_grads G{};
    (void)tokens_backward(mlp, &G, nullptr, n_rows, s, nullptr, nullptr, nullptr, reinterpret_cast<float*>(1), 0, nullptr, 1, 1, 1, 0, 1, ws, true, nullptr);
  } else {
    mlp_forward(c, mlp, nullptr, mlp->dims[0], n_rows, s.mlp, nullptr, mlp->dims[mlp->n_layers]);
  }
  return ws.off + 256;
}

extern "C" int hoisdf_tokens_fwd(const hoisdf_pyramid* pyr, const float* points, const float* center, const float* cam_intr, float scale,
                                 int img_h, int img_w, const float* feat_in, const float* cam_in, const hoisdf_mlp* mlp, const float* pe,
                                 const float* sdf, const float* beta_ptr, float* tok, float* fea_out, float* cam_out, int B, int P, int S,
                                 int row0, int D, void* saved, long saved_bytes, void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(mlp_ok(mlp) && center && pe && sdf && beta_ptr && tok && saved && workspace, HOISDF_ERR_INVALID, "tokens_fwd: null pointer / bad MLP");
  HOISDF_REQUIRE(B > 0 && P > 0 && S >= row0 + P && row0 >= 0 && mlp->act_last && mlp->dims[mlp->n_layers] == D - 33, HOISDF_ERR_INVALID,
                 "tokens_fwd: bad sizes (the MLP must end in D - 33 = %d activated columns)", D - 33);
  const bool gather = feat_in == nullptr;
  HOISDF_REQUIRE(gather ? (pyr && points && cam_intr && cam_out) : (cam_in != nullptr), HOISDF_ERR_INVALID,
                 "tokens_fwd: either gathered rows + their camera points, or a pyramid + points + cam_out");
  const long M = (long)B * P;
  HOISDF_REQUIRE(saved_bytes >= hoisdf_tokens_saved_bytes(mlp, M, gather), HOISDF_ERR_WORKSPACE, "tokens_fwd: saved buffer of %ld bytes, need %ld", saved_bytes,
                 hoisdf_tokens_saved_bytes(mlp, M, gather));
  Bump sv(saved, saved_bytes), ws(workspace, workspace_bytes);
  TokSaved s; tok_carve(mlp, M, gather, sv, s);
  Ctx c{as_stream(stream), stream, &ws, false, gemm_emu_mode()};
  const int C = mlp->dims[0], F = D - 33;
  const float* feat = feat_in;
  const float* cam = cam_in;
  if (gather) {
    c.rc = hoisdf_project_gather_fwd(pyr, points, nullptr, M, P, center, cam_intr, scale, img_h, img_w, s.feat, C, cam_out, nullptr, stream);
    feat = s.feat; cam = cam_out;
  }
  mlp_forward(c, mlp, feat, C, M, s.mlp, s.fea, F);
  if (c.ok()) c.rc = hoisdf_token_build_fwd(cam, center, pe, s.fea, F, sdf, beta_ptr, tok, B, P, S, row0, D, stream);
  if (c.ok() && fea_out && hipMemcpyAsync(fea_out, s.fea, sizeof(float) * M * F, hipMemcpyDeviceToDevice, c.st) != hipSuccess) c.rc = HOISDF_ERR_LAUNCH;
  if (c.ok() && (sv.overflow || ws.overflow)) c.rc = HOISDF_ERR_WORKSPACE;
  if (c.rc == HOISDF_ERR_WORKSPACE) set_error("tokens_fwd: workspace (%ld bytes) too small", workspace_bytes);
  return c.rc;
}

extern "C" int hoisdf_tokens_bwd(const hoisdf_pyramid_grad* dpyr, const float* points, const float* center, const float* cam_intr, float scale,
                                 int img_h, int img_w, const float* feat_in, const hoisdf_mlp* mlp, const float* sdf, const float* beta_ptr,
                                 const float* dtok, const void* saved, long saved_bytes, const hoisdf_mlp_grads* grads, float* dfeat,
                                 int accumulate_dfeat, float* dbeta, int B, int P, int S, int row0, int D, void* workspace, long workspace_bytes,
                                 void* stream) {
  HOISDF_REQUIRE(mlp_ok(mlp) && grads_ok(mlp, grads) && sdf && beta_ptr && dtok && saved && dbeta && workspace, HOISDF_ERR_INVALID,
                 "tokens_bwd: null pointer (every weight-gradient buffer is required, zero-filled)");
  const bool gather = feat_in == nullptr;
  const long M = (long)B * P;
  HOISDF_REQUIRE(B > 0 && P > 0 && saved_bytes >= hoisdf_tokens_saved_bytes(mlp, M, gather), HOISDF_ERR_INVALID, "tokens_bwd: bad sizes / saved buffer");
  HOISDF_REQUIRE(gather ? (points && center && cam_intr) : true, HOISDF_ERR_INVALID, "tokens_bwd: the gather's arguments are needed again");
  Bump sv(const_cast<void*>(saved), saved_bytes), ws(workspace, workspace_bytes);
  TokSaved s; tok_carve(mlp, M, gather, sv, s);
  const int C = mlp->dims[0];
  float* df = dfeat;
  int acc = accumulate_dfeat;
  if (gather && dpyr && !df) {                 // the gathered rows were ours: their gradient lives in the workspace and is scattered below
    df = ws.floats(M * C); acc = 0;
    if (!df) { set_error("tokens_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
  }
  int rc = tokens_backward(mlp, grads, gather ? s.feat : feat_in, M, s, dtok, sdf, beta_ptr, df, acc, dbeta, B, P, S, row0, D, ws, false, stream);
  if (rc == HOISDF_OK && gather && dpyr && df)
    rc = hoisdf_project_gather_bwd(dpyr, points, nullptr, M, P, center, cam_intr, scale, img_h, img_w, df, C, stream);
  return rc;
}

extern "C" long hoisdf_heads_vote_saved_bytes(const hoisdf_mlp* vote, const hoisdf_mlp* cls, int L, int B, int P, int J) {
  if (!mlp_ok(vote) || !mlp_ok(cls) || L <= 0 || B <= 0 || P <= 0 || J <= 0) return 0;
  Bump b(nullptr, 0); VoteSaved s;
  vote_carve(vote, cls, (long)L * B * P, L * B, J, b, s);
  return b.off + 256;
}

static int heads_vote_backward(const hoisdf_mlp* mv, const hoisdf_mlp* mc, const hoisdf_mlp_grads* Gv, const hoisdf_mlp_grads* Gc, const float* enc,
                               const float* pts, const float* gt, float radius, const float* joints, VoteSaved& s, const float* djoints,
                               const float* dl3d, const float* dbce, float* denc, int L, int B, int P, int J, Bump& ws, bool dry, void* stream) {
  Ctx c{as_stream(stream), stream, &ws, dry, gemm_emu_mode()};
  const long M = (long)L * B * P;
  float* doff = ws.floats(M * 3 * J); float* dcls = ws.floats(M * J);
  if (!dry) {
    if (!doff || !dcls) { set_error("heads_vote_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
    c.rc = hoisdf_vote_loss_bwd(s.off, s.cls, pts, gt, radius, joints, s.stats, djoints, dl3d, dbce, doff, dcls, L, B, P, J, stream);
  }
  const int E = mv->dims[0];
  mlp_backward(c, mv, Gv, enc, E, M, s.vote, doff, 3 * J, denc, E, 0);
  mlp_backward(c, mc, Gc, enc, E, M, s.cl, dcls, J, denc, E, 1, enc_mag(c, mv, enc, E, M, s.vote));
  if (c.ok() && !dry && ws.overflow) { set_error("heads_vote_bwd: workspace too small"); return HOISDF_ERR_WORKSPACE; }
  return c.rc;
}

extern "C" long hoisdf_heads_vote_workspace_bytes(const hoisdf_mlp* vote, const hoisdf_mlp* cls, int L, int B, int P, int J, int backward_pass) {
  if (!mlp_ok(vote) || !mlp_ok(cls) || L <= 0 || B <= 0 || P <= 0 || J <= 0) return 0;
  const long M = (long)L * B * P;
  Bump ws(nullptr, 0), sv(nullptr, 0);
  VoteSaved s; vote_carve(vote, cls, M, L * B, J, sv, s);
  if (backward_pass) {
    hoisdf_mlp_grads G{};
    (void)heads_vote_backward(vote, cls, &G, &G, nullptr, nullptr, nullptr, 0.f, nullptr, s, nullptr, nullptr, nullptr, reinterpret_cast<float*>(1), L, B, P, J, ws,
                              true, nullptr);
  } else {
    Ctx c{nullptr, nullptr, &ws, true, gemm_emu_mode()};
    mlp_forward(c, vote, nullptr, vote->dims[0], M, s.vote, nullptr, 3 * J);
    mlp_forward(c, cls, nullptr, cls->dims[0], M, s.cl, nullptr, J);
  }
  return ws.off + 256;
}

extern "C" int hoisdf_heads_vote_fwd(const float* enc, const hoisdf_mlp* vote, const hoisdf_mlp* cls, const float* pts, const float* joint_gt_mm,
                                     float radius, float* joints, float* l3d_sum, float* bce_sum, float* near_sum, int L, int B, int P, int J,
                                     void* saved, long saved_bytes, void* workspace, long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(enc && mlp_ok(vote) && mlp_ok(cls) && pts && joint_gt_mm && joints && l3d_sum && bce_sum && near_sum && saved && workspace,
                 HOISDF_ERR_INVALID, "heads_vote_fwd: null pointer / bad MLP");
  HOISDF_REQUIRE(L > 0 && B > 0 && P > 0 && J > 0 && vote->dims[vote->n_layers] == 3 * J && cls->dims[cls->n_layers] == J &&
                     vote->dims[0] == cls->dims[0] && !vote->act_last && !cls->act_last,
                 HOISDF_ERR_INVALID, "heads_vote_fwd: the vote MLP must end in 3 J = %d, the class MLP in J = %d plain columns", 3 * J, J);
  HOISDF_REQUIRE(saved_bytes >= hoisdf_heads_vote_saved_bytes(vote, cls, L, B, P, J), HOISDF_ERR_WORKSPACE, "heads_vote_fwd: saved buffer too small");
  const long M = (long)L * B * P;
  Bump sv(saved, saved_bytes), ws(workspace, workspace_bytes);
  VoteSaved s; vote_carve(vote, cls, M, L * B, J, sv, s);
  Ctx c{as_stream(stream), stream, &ws, false, gemm_emu_mode()};
  const int E = vote->dims[0];
  mlp_forward(c, vote, enc, E, M, s.vote, s.off, 3 * J);
  mlp_forward(c, cls, enc, E, M, s.cl, s.cls, J, enc_mag(c, vote, enc, E, M, s.vote));        // (enc is measured once, by the first chain)
  if (c.ok()) c.rc = hoisdf_vote_loss_fwd(s.off, s.cls, pts, joint_gt_mm, radius, joints, s.stats, l3d_sum, bce_sum, near_sum, L, B, P, J, stream);
  if (c.ok() && (sv.overflow || ws.overflow)) c.rc = HOISDF_ERR_WORKSPACE;
  if (c.rc == HOISDF_ERR_WORKSPACE) set_error("heads_vote_fwd: workspace (%ld bytes) too small", workspace_bytes);
  return c.rc;
}

extern "C" int hoisdf_heads_vote_bwd(const float* enc, const hoisdf_mlp* vote, const hoisdf_mlp* cls, const float* pts, const float* joint_gt_mm,
                                     float radius, const float* joints, const void* saved, long saved_bytes, const float* djoints,
                                     const float* dl3d_sum, const float* dbce_sum, const hoisdf_mlp_grads* vote_grads,
                                     const hoisdf_mlp_grads* cls_grads, float* denc, int L, int B, int P, int J, void* workspace,
                                     long workspace_bytes, void* stream) {
  HOISDF_REQUIRE(enc && mlp_ok(vote) && mlp_ok(cls) && grads_ok(vote, vote_grads) && grads_ok(cls, cls_grads) && pts && joint_gt_mm && joints && saved &&
                     denc && workspace,
                 HOISDF_ERR_INVALID, "heads_vote_bwd: null pointer (every weight-gradient buffer is required, zero-filled)");
  HOISDF_REQUIRE(L > 0 && B > 0 && P > 0 && J > 0 && saved_bytes >= hoisdf_heads_vote_saved_bytes(vote, cls, L, B, P, J), HOISDF_ERR_INVALID,
                 "heads_vote_bwd: bad sizes / saved buffer");
  Bump sv(const_cast<void*>(saved), saved_bytes), ws(workspace, workspace_bytes);
  VoteSaved s; vote_carve(vote, cls, (long)L * B * P, L * B, J, sv, s);
  return heads_vote_backward(vote, cls, vote_grads, cls_grads, enc, pts, joint_gt_mm, radius, joints, s, djoints, dl3d_sum, dbce_sum, denc, L, B, P, J, ws, false,
                             stream);
}
