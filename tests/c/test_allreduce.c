/* C host of libhoisdf_rccl.so: one rank, all-reduce of a device buffer must return it unchanged (sum over 1 rank).
 * Built and run by tests/test_gpu_model.py::test_c_host_allreduce (hipcc). */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "hoisdf_collective.h"

int main(void) {
  const long n = 1 << 20;
  float* h = (float*)malloc(n * sizeof(float));
  for (long i = 0; i < n; ++i) h[i] = (float)(i % 1000) * 0.25f;
  float* d;
  if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return 2;
  hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
  hoisdf_coll_id id;
  void* comm = 0;
  hipStream_t st;
  hipStreamCreate(&st);
  if (hoisdf_coll_unique_id(&id) || hoisdf_coll_init(&comm, 1, 0, &id) || hoisdf_allreduce(comm, d, n, st)) {
    fprintf(stderr, "FAILED: %s\n", hoisdf_coll_last_error());
    return 1;
  }
  hipStreamSynchronize(st);
  float* r = (float*)malloc(n * sizeof(float));
  hipMemcpy(r, d, n * sizeof(float), hipMemcpyDeviceToHost);
  for (long i = 0; i < n; ++i)
    if (r[i] != h[i]) { fprintf(stderr, "mismatch at %ld\n", i); return 1; }
  if (hoisdf_allreduce(0, d, n, st) == 0) { fprintf(stderr, "null comm accepted\n"); return 1; }
  hoisdf_coll_destroy(comm);
  printf("allreduce ok (%ld floats, 1 rank)\n", n);
  return 0;
}
