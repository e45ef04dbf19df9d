// probe of ds_read_b64_tr_b16 (gfx950): LDS holds u16 "addresses" (element index); every lane reads 8 bytes through the transpose
// read from an address chosen by a pattern and prints what it got.  Build: hipcc --offload-arch=gfx950 ds_tr_probe.hip -o ds_tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int pattern, int pitch, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int off;      // element offset (u16 units), must be 8-byte aligned => multiple of 4
  if (pattern == 0) off = l * 4;                               // lane i -> consecutive 8-byte pieces of a flat array
  else if (pattern == 1) off = (l % 16) * pitch + (l / 16) * 4; // lane i -> row i%16, piece i/16 (pitch u16 per row)
  else off = ((l % 16) / 4) * pitch + ((l % 4) * 4) + (l / 16) * 16;   // 4 rows x 4 pieces per 16-lane group
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off));
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = v[i];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int pat = 0; pat < 3; ++pat) {
    const int pitch = 64;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pat, pitch, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d (pitch %d):\n", pat, pitch);
    for (int l = 0; l < 64; ++l) { printf(" l%02d:[%4d %4d %4d %4d]", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]); if (l % 4 == 3) printf("\n"); }
  }
  return 0;
}
