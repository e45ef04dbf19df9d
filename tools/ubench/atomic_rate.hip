// Float-atomic throughput probe: does it matter whether the workgroups that add into one region
// sit on the same XCD (block id % 8) or on all eight?  hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
// each block adds `per_block` floats (contiguous, 256 threads x per_block/256) into one region
__global__ void k(float* buf, int nreg, int region_floats, int per_block, int mode, int reps) {
  const int b = blockIdx.x;
  int region;
  if (mode == 0) region = (b & 7) + 8 * ((b >> 3) % (nreg / 8));   // same-XCD blocks share a region
  else region = (b >> 3) % nreg;                                     // 8 XCDs hit every region
  float* base = buf + (size_t)region * region_floats;
  for (int r = 0; r < reps; ++r) {
    const int off = ((b * 7 + r) * per_block) % region_floats;
    for (int i = threadIdx.x; i < per_block; i += 256) atomicAdd(base + off + i, 1.0f);
  }
}
__global__ void kstore(float* buf, int nreg, int region_floats, int per_block, int mode, int reps) {
  const int b = blockIdx.x;
  int region = (mode == 0) ? (b & 7) + 8 * ((b >> 3) % (nreg / 8)) : (b >> 3) % nreg;
  float* base = buf + (size_t)region * region_floats;
  for (int r = 0; r < reps; ++r) {
    const int off = ((b * 7 + r) * per_block) % region_floats;
    for (int i = threadIdx.x; i < per_block; i += 256) base[off + i] = 1.0f;
  }
}
int main() {
  const int nreg = 32, region_floats = 131072;   // 32 regions x 512 KB (a dQ slab of one (b, head) at L = 2048)
  float* d; hipMalloc(&d, (size_t)nreg * region_floats * 4); hipMemset(d, 0, (size_t)nreg * region_floats * 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int mode = 0; mode < 2; ++mode)
    for (int which = 0; which < 2; ++which) {
      const int blocks = 4096, per_block = 2048, reps = 64;
      if (which == 0) k<<<blocks, 256>>>(d, nreg, region_floats, per_block, mode, 2); else kstore<<<blocks, 256>>>(d, nreg, region_floats, per_block, mode, 2);
      hipEventRecord(s);
      if (which == 0) k<<<blocks, 256>>>(d, nreg, region_floats, per_block, mode, reps); else kstore<<<blocks, 256>>>(d, nreg, region_floats, per_block, mode, reps);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      double n = (double)blocks * per_block * reps;
      printf("%s %s: %.1f G floats/s (%.2f ms)\n", which == 0 ? "atomicAdd" : "store    ", mode == 0 ? "XCD-local regions" : "cross-XCD regions", n / ms / 1e6, ms);
    }
  return 0;
}
