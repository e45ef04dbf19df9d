/* A C host with no Python and no torch runs ONE transformer encoder layer through the coarse entries of include/hoisdf.h
 * (hoisdf_encoder_layer_fwd / _bwd) on values dumped from fixture g5 (the reference's own TransformerEncoderLayer output and
 * the stack's inter_norm of it, tests/golden/make_golden.py) and checks
 *   - the layer output and its inter_norm against the fixture,
 *   - the backward against a central finite difference of the forward along a random direction (dropout off).
 * File layout (little-endian): int32 B, S, E, F, H; then float32 arrays x[B][S][E], w_in, b_in, w_out, b_out, g1, be1, w1, b1,
 * w2, b2, g2, be2, g3, be3, expected x_out[B][S][E], expected y[B][S][E].
 * Built and run by tests/test_gpu_model.py::test_c_host_encoder_layer (hipcc). */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hoisdf.h"

#define CHECK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "hip error line %d\n", __LINE__); return 2; } } while (0)
#define CALL(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "line %d: status %d: %s\n", __LINE__, rc_, hoisdf_last_error()); return 1; } } while (0)

static float* upload(FILE* f, long n, float** host) {
  float* h = (float*)malloc(sizeof(float) * n);
  if (fread(h, sizeof(float), n, f) != (size_t)n) { fprintf(stderr, "short read\n"); exit(2); }
  float* d = NULL;
  if (hipMalloc((void**)&d, sizeof(float) * n) != hipSuccess || hipMemcpy(d, h, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) exit(2);
  if (host) *host = h; else free(h);
  return d;
}
static float* dzeros(long n) {
  float* d = NULL;
  if (hipMalloc((void**)&d, sizeof(float) * n) != hipSuccess || hipMemset(d, 0, sizeof(float) * n) != hipSuccess) exit(2);
  return d;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <dump.bin>\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  int hdr[5];
  if (fread(hdr, sizeof(int), 5, f) != 5) return 2;
  const int B = hdr[0], S = hdr[1], E = hdr[2], F = hdr[3], H = hdr[4];
  const long n = (long)B * S * E;
  float *hx, *hexp_x, *hexp_y;
  hoisdf_encoder_layer_weights w;
  memset(&w, 0, sizeof(w));                               /* no weight images: the library builds what it needs */
  float* x = upload(f, n, &hx);
  w.w_in = upload(f, 3L * E * E, NULL); w.b_in = upload(f, 3L * E, NULL); w.w_out = upload(f, (long)E * E, NULL); w.b_out = upload(f, E, NULL);
  w.g1 = upload(f, E, NULL); w.be1 = upload(f, E, NULL); w.w1 = upload(f, (long)F * E, NULL); w.b1 = upload(f, F, NULL);
  w.w2 = upload(f, (long)E * F, NULL); w.b2 = upload(f, E, NULL); w.g2 = upload(f, E, NULL); w.be2 = upload(f, E, NULL);
  w.g3 = upload(f, E, NULL); w.be3 = upload(f, E, NULL);
  float* exp_x = upload(f, n, &hexp_x); float* exp_y = upload(f, n, &hexp_y);
  (void)exp_x; (void)exp_y;
  fclose(f);

  hoisdf_encoder_layer_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.S = S; d.E = E; d.F = F; d.H = H; d.eps = 1e-5f; d.drop_p = 0.f; d.attention = 2; d.training = 1;
  const long nsaved = hoisdf_encoder_layer_saved_bytes(&d), nws_f = hoisdf_encoder_layer_workspace_bytes(&d, 0),
             nws_b = hoisdf_encoder_layer_workspace_bytes(&d, 1);
  if (nsaved <= 0 || nws_f <= 0 || nws_b <= 0) { fprintf(stderr, "size query failed: %s\n", hoisdf_last_error()); return 1; }
  void *saved, *ws;
  CHECK(hipMalloc(&saved, nsaved)); CHECK(hipMalloc(&ws, nws_f > nws_b ? nws_f : nws_b));
  float *x_out = dzeros(n), *y_out = dzeros(n);
  hipStream_t st; CHECK(hipStreamCreate(&st));
  printf("%s: encoder layer B=%d S=%d E=%d F=%d H=%d, saved %ld B, workspace %ld / %ld B\n", hoisdf_version(), B, S, E, F, H, nsaved, nws_f, nws_b);
  CALL(hoisdf_encoder_layer_fwd(x, &w, &d, x_out, y_out, saved, nsaved, ws, nws_f, st));
  CHECK(hipStreamSynchronize(st));
  float *hxo = (float*)malloc(sizeof(float) * n), *hyo = (float*)malloc(sizeof(float) * n);
  CHECK(hipMemcpy(hxo, x_out, sizeof(float) * n, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hyo, y_out, sizeof(float) * n, hipMemcpyDeviceToHost));
  double ex = 0, ey = 0;
  for (long i = 0; i < n; ++i) { double a = fabs(hxo[i] - hexp_x[i]), b = fabs(hyo[i] - hexp_y[i]); if (a > ex) ex = a; if (b > ey) ey = b; }
  printf("forward vs the reference fixture: layer output max abs err %.3e, inter_norm %.3e\n", ex, ey);
  if (!(ex < 2e-5 && ey < 2e-5)) return 1;                /* activations are O(1); the CPU oracle's own bar on g5 is 1e-5 .. 5e-5 */

  /* too small a workspace is refused before anything is launched into it */
  if (hoisdf_encoder_layer_fwd(x, &w, &d, x_out, y_out, saved, nsaved, ws, 1024, st) != HOISDF_ERR_WORKSPACE) return 1;

  /* backward of L = sum(x_out * gx) + sum(y * gy) against a central difference along direction v */
  float *hg = (float*)malloc(sizeof(float) * 2 * n), *hv = (float*)malloc(sizeof(float) * n);
  unsigned s = 777u;
  for (long i = 0; i < 2 * n; ++i) { s = s * 1664525u + 1013904223u; hg[i] = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
  for (long i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hv[i] = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
  float *gx = dzeros(n), *gy = dzeros(n), *dx = dzeros(n);
  CHECK(hipMemcpy(gx, hg, sizeof(float) * n, hipMemcpyHostToDevice)); CHECK(hipMemcpy(gy, hg + n, sizeof(float) * n, hipMemcpyHostToDevice));
  hoisdf_encoder_layer_grads G;
  G.dw_in = dzeros(3L * E * E); G.db_in = dzeros(3L * E); G.dw_out = dzeros((long)E * E); G.db_out = dzeros(E); G.dg1 = dzeros(E); G.dbe1 = dzeros(E);
  G.dw1 = dzeros((long)F * E); G.db1 = dzeros(F); G.dw2 = dzeros((long)E * F); G.db2 = dzeros(E); G.dg2 = dzeros(E); G.dbe2 = dzeros(E);
  G.dg3 = dzeros(E); G.dbe3 = dzeros(E);
  CALL(hoisdf_encoder_layer_bwd(x, x_out, &w, &d, saved, nsaved, gx, gy, dx, &G, ws, nws_b, st));
  CHECK(hipStreamSynchronize(st));
  float* hdx = (float*)malloc(sizeof(float) * n);
  CHECK(hipMemcpy(hdx, dx, sizeof(float) * n, hipMemcpyDeviceToHost));
  double analytic = 0;
  for (long i = 0; i < n; ++i) analytic += (double)hdx[i] * hv[i];
  double L[2];
  const float h = 2e-3f;
  float* xp = (float*)malloc(sizeof(float) * n);
  d.training = 0;
  for (int sgn = 0; sgn < 2; ++sgn) {
    for (long i = 0; i < n; ++i) xp[i] = hx[i] + (sgn ? -h : h) * hv[i];
    CHECK(hipMemcpy(x, xp, sizeof(float) * n, hipMemcpyHostToDevice));
    const long nws0 = hoisdf_encoder_layer_workspace_bytes(&d, 0);
    void* ws0; CHECK(hipMalloc(&ws0, nws0));
    CALL(hoisdf_encoder_layer_fwd(x, &w, &d, x_out, y_out, NULL, 0, ws0, nws0, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(hxo, x_out, sizeof(float) * n, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hyo, y_out, sizeof(float) * n, hipMemcpyDeviceToHost));
    double acc = 0;
    for (long i = 0; i < n; ++i) acc += (double)hxo[i] * hg[i] + (double)hyo[i] * hg[n + i];
    L[sgn] = acc;
    CHECK(hipFree(ws0));
  }
  const double numeric = (L[0] - L[1]) / (2.0 * h);
  printf("backward: <dx, v> = %.6f, central difference %.6f\n", analytic, numeric);
  if (!(fabs(analytic - numeric) <= 2e-2 * fabs(numeric) + 2e-2)) return 1;
  float hdb[8];
  CHECK(hipMemcpy(hdb, G.db_out, sizeof(float) * 8, hipMemcpyDeviceToHost));
  double sdb = 0; for (int i = 0; i < 8; ++i) sdb += fabs(hdb[i]);
  if (!(sdb > 0) || sdb != sdb) return 1;                  /* parameter gradients were written */
  printf("c host encoder layer ok\n");
  return 0;
}
