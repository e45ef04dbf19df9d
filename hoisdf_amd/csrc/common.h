// Shared device/host helpers for the HOISDF gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hoisdf.h"

namespace hoisdf {

// ---- error plumbing (thread-local message, int codes; never throws) -------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define HOISDF_REQUIRE(cond, code, ...)                 \
  do {                                                  \
    if (!(cond)) {                                      \
      ::hoisdf::set_error(__VA_ARGS__);                 \
      return (code);                                    \
    }                                                   \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- counter-based RNG for dropout ----------------------------------------------------
// keep(idx) is a pure function of (seed, idx) so the backward pass regenerates the mask.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float rand01(uint64_t seed, uint64_t idx) {
  uint32_t h = mix32((uint32_t)idx ^ mix32((uint32_t)(idx >> 32) + (uint32_t)seed) ^
                     mix32((uint32_t)(seed >> 32) + 0x9E3779B9U));
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// multiplier applied to a kept element; 0 for a dropped one.  p == 0 -> exactly 1.
__device__ __forceinline__ float drop_scale(float p, float inv_keep, uint64_t seed, uint64_t idx) {
  if (p <= 0.f) return 1.f;
  return rand01(seed, idx) >= p ? inv_keep : 0.f;
}

// ---- wave-level reductions (64 lanes) -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a linear block id: consecutive logical ids land on the same
// XCD (block b is dispatched to XCD b % 8), so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int nx = 8;
  int xcd = bid % nx, loc = bid / nx;
  int q = nblk / nx, r = nblk % nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}

}  // namespace hoisdf
