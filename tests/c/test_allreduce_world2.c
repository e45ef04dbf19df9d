/* C host of libhoisdf_rccl.so, world size 2, rendezvous over a FILE (what a launcher without MPI / torch does):
 *   rank 0: hoisdf_coll_unique_id -> write the 256-byte token to <path>.tmp, rename to <path>
 *   rank 1: poll for <path>, read the token
 *   both  : hipSetDevice(rank % device count), hoisdf_coll_init(world = 2), all-reduce of rank-dependent data, check the sum.
 * usage: test_allreduce_world2 <rank> <path>.  Prints "id <checksum>" once the token is in hand (the plumbing this test is
 * about), then "allreduce ok" - or "init refused: <message>" when RCCL will not put two ranks on the one visible GPU.
 * Built and run by tests/test_gpu_model.py::test_c_host_allreduce_two_ranks_file_rendezvous. */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "hoisdf_collective.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int rank = atoi(argv[1]);
  const char* path = argv[2];
  hoisdf_coll_id id;
  if (rank == 0) {
    if (hoisdf_coll_unique_id(&id)) { fprintf(stderr, "FAILED: %s\n", hoisdf_coll_last_error()); return 1; }
    char tmp[1024];
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    FILE* f = fopen(tmp, "wb");
    if (!f || fwrite(&id, sizeof id, 1, f) != 1) return 2;
    fclose(f);
    if (rename(tmp, path)) return 2;
  } else {
    FILE* f = 0;
    for (int i = 0; i < 600 && !(f = fopen(path, "rb")); ++i) usleep(50000);
    if (!f || fread(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "FAILED: no rendezvous file\n"); return 2; }
    fclose(f);
  }
  unsigned sum = 0;
  for (unsigned i = 0; i < sizeof id; ++i) sum = sum * 131u + (unsigned char)id.bytes[i];
  printf("id %u\n", sum);
  fflush(stdout);
  int ndev = 0;
  hipGetDeviceCount(&ndev);
  if (ndev < 1 || hipSetDevice(rank % ndev) != hipSuccess) return 2;
  const long n = 1 << 18;
  float* h = (float*)malloc(n * sizeof(float));
  for (long i = 0; i < n; ++i) h[i] = (float)((i % 997) + 1000 * rank);
  float* d;
  if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return 2;
  hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
  void* comm = 0;
  if (hoisdf_coll_init(&comm, 2, rank, &id)) {
    printf("init refused: %s\n", hoisdf_coll_last_error());
    return 0;
  }
  hipStream_t st;
  hipStreamCreate(&st);
  if (hoisdf_allreduce(comm, d, n, st)) { fprintf(stderr, "FAILED: %s\n", hoisdf_coll_last_error()); return 1; }
  hipStreamSynchronize(st);
  hipMemcpy(h, d, n * sizeof(float), hipMemcpyDeviceToHost);
  for (long i = 0; i < n; ++i)
    if (h[i] != (float)(2 * (i % 997) + 1000)) { fprintf(stderr, "mismatch at %ld: %f\n", i, h[i]); return 1; }
  hoisdf_coll_destroy(comm);
  printf("allreduce ok (%ld floats, 2 ranks)\n", n);
  return 0;
}
