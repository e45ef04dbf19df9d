// K6: per-sample selection of the k candidates with the smallest |sdf| (the reference sorts all
// ~20-30 k survivors with torch.sort and keeps the head: main/model.py:345-352).
// One 1024-thread workgroup per sample:
//   1. 4-pass 8-bit radix select on the uint bit pattern of |sdf| (monotonic for non-negative
//      floats) -> exact k-th smallest key T and how many ties at T to take;
//   2. ordered wave-ballot compaction of {key < T} U {first ties at T} into LDS (row order);
//   3. bitonic sort of the k (key,row) pairs in LDS -> ascending (|sdf|, row), like the reference's
//      sorted head;  HBM traffic = 2 reads of the candidate keys + k indices written.
#include "common.h"

namespace hoisdf {

constexpr int SEL_T = 1024;
constexpr int SEL_MAXK = 8192;

__device__ __forceinline__ uint32_t abs_key(float v) { return __float_as_uint(fabsf(v)); }

__global__ __launch_bounds__(SEL_T) void select_kernel(const float* __restrict__ sdf, const int32_t* __restrict__ offsets,
                                                       const int32_t* __restrict__ counts, int k, int kpow2,
                                                       int32_t* __restrict__ sel) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long pairs[];   // kpow2 entries
  __shared__ int hist[256];
  __shared__ uint32_t s_prefix;
  __shared__ int s_remaining;
  __shared__ int wave_lt[16], wave_eq[16];
  __shared__ int run_out, run_eq;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = offsets[b], n = counts[b];
  const float* x = sdf + base;
  if (n < k) {        // caller checks this on the host and raises; keep the output defined
    for (int i = tid; i < k; i += SEL_T) sel[(size_t)b * k + i] = -1;
    return;
  }
  // ---- 1. radix select ----------------------------------------------------------------------
  if (tid == 0) { s_prefix = 0u; s_remaining = k; }
  uint32_t maskbits = 0u;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += SEL_T) hist[i] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    for (int i = tid; i < n; i += SEL_T) {
      const uint32_t key = abs_key(x[i]);
      if ((key & maskbits) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int rem = s_remaining, cum = 0, d = 0;
      for (; d < 256; ++d) {
        if (cum + hist[d] >= rem) break;
        cum += hist[d];
      }
      s_prefix = prefix | ((uint32_t)d << shift);
      s_remaining = rem - cum;
    }
    maskbits |= 0xFFu << shift;
    __syncthreads();
  }
  const uint32_t T = s_prefix;
  const int take_eq = s_remaining;          // ties at T to take, lowest rows first
  // ---- 2. ordered compaction into LDS ---------------------------------------------------------
  if (tid == 0) { run_out = 0; run_eq = 0; }
  for (int i = tid; i < kpow2; i += SEL_T) pairs[i] = ~0ULL;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += SEL_T) {
    const int i = c0 + tid;
    uint32_t key = 0xFFFFFFFFu;
    bool lt = false, eq = false;
    if (i < n) { key = abs_key(x[i]); lt = key < T; eq = key == T; }
    const unsigned long long mlt = __ballot(lt), meq = __ballot(eq);
    if (lane == 0) { wave_lt[wave] = __popcll(mlt); wave_eq[wave] = __popcll(meq); }
    __syncthreads();
    int eq_before = run_eq, out_before = run_out;
    for (int w = 0; w < wave; ++w) eq_before += wave_eq[w];
    const unsigned long long below = (1ULL << lane) - 1ULL;
    const int my_eq_rank = eq_before + __popcll(meq & below);
    const bool take = lt || (eq && my_eq_rank < take_eq);
    // output position: everything taken before me in row order
    // taken-before = lt-before + min(eq-before, take_eq)
    int lt_before = 0;
    for (int w = 0; w < wave; ++w) lt_before += wave_lt[w];
    lt_before += __popcll(mlt & below);
    const int eq_taken_before = min(my_eq_rank, take_eq) - min(run_eq, take_eq);
    if (take) {
      const int pos = out_before + lt_before + eq_taken_before;
      pairs[pos] = ((unsigned long long)key << 32) | (uint32_t)i;
    }
    __syncthreads();
    if (tid == 0) {
      int slt = 0, seq = 0;
      for (int w = 0; w < 16; ++w) { slt += wave_lt[w]; seq += wave_eq[w]; }
      const int eq_new = run_eq + seq;
      run_out += slt + (min(eq_new, take_eq) - min(run_eq, take_eq));
      run_eq = eq_new;
    }
    __syncthreads();
  }
  // ---- 3. bitonic sort of kpow2 pairs (padding = all ones sorts last) --------------------------
  for (int size = 2; size <= kpow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (kpow2 >> 1); t += SEL_T) {
        const int lo = ((t / stride) * (stride << 1)) + (t % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long a = pairs[lo], c = pairs[hi];
        if ((a > c) == up) { pairs[lo] = c; pairs[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += SEL_T) sel[(size_t)b * k + i] = base + (int)(uint32_t)(pairs[i] & 0xFFFFFFFFULL);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int lds_,
                                                          const int32_t* __restrict__ sel, long n_sel, int width,
                                                          float* __restrict__ out, int ldo) {
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = n_sel * width;
  for (; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / width;
    const int c = (int)(idx - r * width);
    const int s = sel[r];
    out[(size_t)r * ldo + c] = s >= 0 ? src[(size_t)s * lds_ + c] : 0.f;
  }
}

}  // namespace hoisdf

using namespace hoisdf;

extern "C" int hoisdf_select_smallest_abs(const float* sdf_raw, const int32_t* offsets, const int32_t* counts,
                                          int B, int k, int32_t* sel, void* stream) {
  HOISDF_REQUIRE(sdf_raw && offsets && counts && sel, HOISDF_ERR_INVALID, "select_smallest_abs: null pointer");
  HOISDF_REQUIRE(B > 0 && k > 0 && k <= SEL_MAXK, HOISDF_ERR_INVALID, "select_smallest_abs: k=%d must be in [1, %d]",
                 k, SEL_MAXK);
  int kpow2 = 2;
  while (kpow2 < k) kpow2 <<= 1;
  if ((size_t)kpow2 * 8 > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          kpow2 * 8) != hipSuccess) {
    set_error("select_smallest_abs: cannot raise dynamic LDS to %d bytes", kpow2 * 8);
    return HOISDF_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(select_kernel, dim3(B), dim3(SEL_T), (size_t)kpow2 * 8, as_stream(stream), sdf_raw, offsets,
                     counts, k, kpow2, sel);
  return check_launch("select_smallest_abs");
}

extern "C" int hoisdf_gather_rows(const float* src, int lds_, const int32_t* sel, long n_sel, int width, float* out,
                                  int ldo, void* stream) {
  HOISDF_REQUIRE(src && sel && out && n_sel >= 0 && width > 0 && lds_ >= width && ldo >= width, HOISDF_ERR_INVALID,
                 "gather_rows: bad arguments");
  if (n_sel == 0) return HOISDF_OK;
  long blocks = (n_sel * width + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, lds_, sel,
                     n_sel, width, out, ldo);
  return check_launch("gather_rows");
}
