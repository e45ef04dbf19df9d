/* A C host with no Python and no torch: y = relu(x W^T + b) through libhoisdf_hip.so, checked against a scalar
 * loop; then posenc.  Demonstrates the drop-in boundary of include/hoisdf.h (plain pointers, sizes, stream).
 * Built and run by tests/test_gpu_model.py::test_c_host_linear (hipcc). */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "hoisdf.h"

#define CHECK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "hip error line %d\n", __LINE__); return 2; } } while (0)

int main(void) {
  const int M = 300, N = 223, K = 289;
  float *x = (float*)malloc(sizeof(float) * M * K), *W = (float*)malloc(sizeof(float) * N * K), *b = (float*)malloc(sizeof(float) * N);
  float *y = (float*)malloc(sizeof(float) * M * N);
  unsigned s = 12345u;
  for (int i = 0; i < M * K; ++i) { s = s * 1664525u + 1013904223u; x[i] = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
  for (int i = 0; i < N * K; ++i) { s = s * 1664525u + 1013904223u; W[i] = (((s >> 8) & 0xFFFF) / 65536.0f - 0.5f) * 0.1f; }
  for (int i = 0; i < N; ++i) b[i] = 0.01f * (float)(i % 7 - 3);
  float *dx, *dW, *db, *dy;
  CHECK(hipMalloc((void**)&dx, sizeof(float) * M * K)); CHECK(hipMalloc((void**)&dW, sizeof(float) * N * K));
  CHECK(hipMalloc((void**)&db, sizeof(float) * N)); CHECK(hipMalloc((void**)&dy, sizeof(float) * M * N));
  CHECK(hipMemcpy(dx, x, sizeof(float) * M * K, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dW, W, sizeof(float) * N * K, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, b, sizeof(float) * N, hipMemcpyHostToDevice));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  printf("%s\n", hoisdf_version());
  int rc = hoisdf_linear_fwd(dx, K, dW, K, db, dy, N, M, N, K, /*act=*/1, /*drop_p=*/0.f, /*seed=*/0, /*relu_bits=*/NULL, st);
  if (rc) { fprintf(stderr, "hoisdf_linear_fwd: %d %s\n", rc, hoisdf_last_error()); return 1; }
  CHECK(hipStreamSynchronize(st));
  CHECK(hipMemcpy(y, dy, sizeof(float) * M * N, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double acc = b[n];
      for (int k = 0; k < K; ++k) acc += (double)x[m * K + k] * (double)W[n * K + k];
      if (acc < 0) acc = 0;
      double e = fabs(acc - (double)y[m * N + n]);
      if (e > worst) worst = e;
    }
  printf("linear_fwd %dx%dx%d: max abs err vs double loop %.3e\n", M, N, K, worst);
  if (!(worst < 2e-5)) return 1;
  /* argument validation never touches the device */
  if (hoisdf_linear_fwd(NULL, K, dW, K, db, dy, N, M, N, K, 1, 0.f, 0, NULL, st) != HOISDF_ERR_INVALID) return 1;
  printf("c host ok\n");
  return 0;
}
