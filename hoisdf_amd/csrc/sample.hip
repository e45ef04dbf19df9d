// (f2) GPU-side SDF sample selection: the dataset's per-frame `np.random.choice(..., replace=False)` draws
// (reference data/dexycb.py:514-546) done on the device from an HBM-resident store of the `sdf_processed` rows
// ([x y z sdf_hand sdf_obj label] float32, tool/pre_process_sdf.py:140-148).
// Uniform sampling without replacement = "give every eligible row an i.i.d. uniform key and keep the k smallest":
// this kernel writes the keys (counter-based hash of (seed, segment, row) -> 24-bit uniform; ineligible rows
// get +big) and counts the eligible rows per segment; hoisdf_select_smallest_abs then picks the k smallest keys of
// every segment (in ascending key order = a uniformly random order) and hoisdf_gather_rows fetches the rows.
// Not bit-identical to numpy's Mersenne-Twister stream (cannot be); same distribution, tested statistically.
#include "common.h"

namespace hoisdf {

__global__ __launch_bounds__(256) void sdf_sample_keys_kernel(const float* __restrict__ rows, int ld,
                                                              const int64_t* __restrict__ seg_row0,
                                                              const int32_t* __restrict__ seg_len,
                                                              const int32_t* __restrict__ seg_off,
                                                              const int32_t* __restrict__ seg_col, float dist,
                                                              uint64_t seed, float* __restrict__ keys,
                                                              int32_t* __restrict__ eligible) {
  const int s = blockIdx.y;
  const int n = seg_len[s], col = seg_col[s];
  const float* base = rows + (size_t)seg_row0[s] * ld;
  float* out = keys + seg_off[s];
  const uint32_t sk = drop_rowkey(seed, (uint32_t)s);
  int cnt = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const bool ok = col < 0 || fabsf(base[(size_t)i * ld + col]) < dist;
    // 24 random bits -> [0, 1): exactly representable, so key order = integer order
    const uint32_t h = mix32(((uint32_t)i * 0x85EBCA77U) ^ sk) >> 8;
    out[i] = ok ? (float)h * (1.0f / 16777216.0f) : 1e30f;
    cnt += ok;
  }
  cnt = (int)wave_sum((float)cnt);            // <= 2^24 rows per segment: exact in float
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&eligible[s], cnt);
}

}  // namespace hoisdf

using namespace hoisdf;

extern "C" int hoisdf_sdf_sample_keys(const float* rows, int ld, const int64_t* seg_row0, const int32_t* seg_len,
                                      const int32_t* seg_off, const int32_t* seg_col, int n_seg, int max_len,
                                      float dist, uint64_t seed, float* keys, int32_t* eligible, void* stream) {
  HOISDF_REQUIRE(rows && seg_row0 && seg_len && seg_off && seg_col && keys && eligible, HOISDF_ERR_INVALID,
                 "sdf_sample_keys: null pointer");
  HOISDF_REQUIRE(ld >= 5 && n_seg >= 0 && max_len >= 0 && max_len <= (1 << 24), HOISDF_ERR_INVALID,
                 "sdf_sample_keys: ld=%d n_seg=%d max_len=%d", ld, n_seg, max_len);
  if (n_seg == 0 || max_len == 0) return HOISDF_OK;
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(eligible, 0, sizeof(int32_t) * n_seg, st) != hipSuccess) {
    set_error("sdf_sample_keys: memset failed");
    return HOISDF_ERR_LAUNCH;
  }
  int gx = cdiv(max_len, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(sdf_sample_keys_kernel, dim3(gx, n_seg), dim3(256), 0, st, rows, ld, seg_row0, seg_len, seg_off,
                     seg_col, dist, seed, keys, eligible);
  return check_launch("sdf_sample_keys");
}
