// Small-M linear layers as fp32 emulated on the bf16 MFMA pipe (round 4): the 17-query decoder stack and the regression heads run
// ~100 GEMMs of 544 x 256 x 256 (.. x 1024) per training step.  The tiled kernels (gemm.hip, gemm_emu.hip) cut such a problem
// into a handful of 128- / 256-row tiles and take 12-30 us each - latency, not arithmetic.  Here one workgroup owns one 32 x 32
// output tile and its four waves split the contraction; there is no LDS staging of operands and one barrier (the final add of the
// four partial tiles, in wave order): the operands (a few hundred KB, L2-resident) are read straight in the fragment layout of
// v_mfma_f32_32x32x16_bf16 (lane = row / column of the tile, 8 consecutive contraction indices per lane and 16-step), split into
// their three bf16 pieces in registers and fed to the six products of the emulation.  A 544 x 256 output is 136 workgroups, each
// wave 4 load -> split -> MFMA rounds long at K = 256.  Same contracts, epilogues (bias, ReLU, dropout, 1-bit sign map) and
// arithmetic (exact three-way split of both operands, six products, f32 accumulation) as hoisdf_linear_*_emu; reference call
// sites: common/nets/transformer.py:366-395 (decoder layer), common/nets/layer.py:168-201 (MLP).
#include "common.h"

namespace hoisdf {
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#define SPLIT1(x, i)                             \
  do {                                           \
    const __bf16 a_ = (__bf16)(x);               \
    const float r1_ = (x) - (float)a_;           \
    const __bf16 b_ = (__bf16)r1_;               \
    const float r2_ = r1_ - (float)b_;           \
    p0[i] = a_; p1[i] = b_; p2[i] = (__bf16)r2_; \
  } while (0)
struct Frag { bf16x8 p[3]; };
__device__ __forceinline__ Frag split8(const float (&e)[8]) {
  bf16x8 p0, p1, p2;
  SPLIT1(e[0], 0); SPLIT1(e[1], 1); SPLIT1(e[2], 2); SPLIT1(e[3], 3);
  SPLIT1(e[4], 4); SPLIT1(e[5], 5); SPLIT1(e[6], 6); SPLIT1(e[7], 7);
  Frag f; f.p[0] = p0; f.p[1] = p1; f.p[2] = p2;
  return f;
}
// the six products of one 16-step, small terms first
__device__ __forceinline__ f32x16 mul6(const Frag& a, const Frag& b, f32x16 c) {
  c = MFB(a.p[2], b.p[0], c); c = MFB(a.p[0], b.p[2], c); c = MFB(a.p[1], b.p[1], c);
  c = MFB(a.p[1], b.p[0], c); c = MFB(a.p[0], b.p[1], c); c = MFB(a.p[0], b.p[0], c);
  return c;
}

struct SmallArgs {
  const float* A; long lda;             // the row operand: x (forward) or dy (grad-input), [M][lda]
  const float* W; long ldw;             // [N][ldw] weights
  float* C; long ldc;
  const float* bias;
  const uint32_t* abits; int ldbits; float ascale;      // sign bitmap of dy and 1 / keep (grad-input, grad-weight)
  uint32_t* bits_out; int ldbits_out;
  int M, N, K;                          // forward: y[M][N] over K; grad-input: dx[M][K] over N
  int act; float drop_p, inv_keep; uint32_t thresh; uint64_t seed;
  int accumulate;
};

// 8 consecutive floats of a row (two 16-byte loads; lim - k0 is a multiple of 4), zero past the limit
__device__ __forceinline__ void ld8(const float* p, int k0, int lim, float (&e)[8]) {
  const float4 u = k0 < lim ? *reinterpret_cast<const float4*>(p + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 v = k0 + 4 < lim ? *reinterpret_cast<const float4*>(p + k0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  e[0] = u.x; e[1] = u.y; e[2] = u.z; e[3] = u.w; e[4] = v.x; e[5] = v.y; e[6] = v.z; e[7] = v.w;
}

// DX = false: y[m][n] = sum_k x[m][k] W[n][k] (+ bias, ReLU, dropout, sign map);  DX = true: dx[m][k] = sum_n dy_eff[m][n] W[n][k].
// One workgroup per 32 x 32 tile: its four waves take the 16-steps s = wave, wave + 4, ... of the contraction (a 256-long one is
// four dependent load -> split -> MFMA rounds per wave instead of sixteen) and the partial tiles are added in wave order through LDS.
template <bool DX>
__global__ __launch_bounds__(256) void emu_small_kernel(SmallArgs g) {
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int ncol = DX ? g.K : g.N, ncon = DX ? g.N : g.K;          // output columns, contraction length
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int arow = min(m0 + l31, g.M - 1);
  const bool avalid = m0 + l31 < g.M;
  const int bcol = min(n0 + l31, ncol - 1);
  const float* ap = g.A + (size_t)arow * g.lda;
  const int nslab = (ncon + 15) / 16;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ea[8], eb[8], na[8], nb[8];
  uint32_t mw = 0xffu, nmw = 0xffu;
  auto load = [&](int s, float (&a)[8], float (&b)[8], uint32_t& m) {
    const int c0 = s * 16 + 8 * kh;
    ld8(ap, c0, avalid ? ncon : 0, a);
    if (DX) {
      if (g.abits) m = c0 < ncon ? g.abits[(size_t)arow * g.ldbits + (c0 >> 5)] >> (c0 & 31) : 0u;
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = c0 + e < ncon ? g.W[(size_t)(c0 + e) * g.ldw + bcol] : 0.f;
    } else {
      ld8(g.W + (size_t)bcol * g.ldw, c0, ncon, b);
    }
  };
  if (wave < nslab) load(wave, ea, eb, mw);
  for (int s = wave; s < nslab; s += 4) {
    const bool more = s + 4 < nslab;
    if (more) load(s + 4, na, nb, nmw);                            // the wave's next 16-step is in flight under this one's products
    if (DX && g.abits) {
#pragma unroll
      for (int e = 0; e < 8; ++e) ea[e] = (mw >> e) & 1u ? ea[e] : 0.f;
    }
    const Frag fa = split8(ea), fb = split8(eb);
    acc = mul6(fa, fb, acc);
    if (more) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { ea[e] = na[e]; eb[e] = nb[e]; }
      mw = nmw;
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
  // ---- epilogue (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
  const int col = n0 + l31;
  const bool cvalid = col < ncol;
  const float bv = (!DX && g.bias && cvalid) ? g.bias[col] : 0.f;
  const float post = DX && g.abits ? g.ascale : 1.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    float v = acc[r] * post + bv;
    if (!DX) {
      if (g.act == 1) v = fmaxf(v, 0.f);
      if (g.drop_p > 0.f) v *= drop_scale(drop_rowkey(g.seed, (uint32_t)row), (uint32_t)col, g.thresh, g.inv_keep);
      if (g.bits_out) {
        // lanes 0-31: the 32 columns of `row`, lanes 32-63: of row + 4 - one ballot is the two mask words
        const unsigned long long q = __ballot(cvalid && v > 0.f);
        if (l31 == 0 && row < g.M) g.bits_out[(size_t)row * g.ldbits_out + (n0 >> 5)] = (uint32_t)(kh ? q >> 32 : q);
      }
    }
    if (row < g.M && cvalid) {
      float* cp = g.C + (size_t)row * g.ldc + col;
      *cp = g.accumulate ? *cp + v : v;
    }
  }
}

struct SmallDwArgs {
  const float* dy; long lddy;
  const float* x; long ldx;
  const uint32_t* bits; int ldbits; float ascale;
  float* dW; long lddw; float* db;
  int M, N, K;
};

// dW[n][k] = sum_m dy_eff[m][n] x[m][k] (+ db[n] = sum_m dy_eff[m][n]): one workgroup per 32 x 32 tile, its four waves take the
// row slabs s = wave, wave + 4, ... and their partial tiles are added in wave order through LDS (order-fixed, overwrites dW / db)
__global__ __launch_bounds__(256) void emu_small_dw_kernel(SmallDwArgs g) {
  __shared__ float red[3][16][64];
  __shared__ float rdb[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int ncol = min(n0 + l31, g.N - 1), kcol = min(k0 + l31, g.K - 1);
  const bool nvalid = n0 + l31 < g.N, kvalid = k0 + l31 < g.K;
  const int nslab = (g.M + 15) / 16;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;
  float ea[8], eb[8], na[8], nb[8];
  auto load = [&](int s, float (&av)[8], float (&bv)[8]) {
    const int mb = s * 16 + 8 * kh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = mb + e;
      float a = 0.f, b = 0.f;
      if (m < g.M) {
        if (nvalid) {
          a = g.dy[(size_t)m * g.lddy + ncol];
          if (g.bits) a = (g.bits[(size_t)m * g.ldbits + (ncol >> 5)] >> (ncol & 31)) & 1u ? a : 0.f;
        }
        if (kvalid) b = g.x[(size_t)m * g.ldx + kcol];
      }
      av[e] = a; bv[e] = b;
    }
  };
  if (wave < nslab) load(wave, ea, eb);
  for (int s = wave; s < nslab; s += 4) {
    const bool more = s + 4 < nslab;
    if (more) load(s + 4, na, nb);
#pragma unroll
    for (int e = 0; e < 8; ++e) colsum += ea[e];
    const Frag fa = split8(ea), fb = split8(eb);
    acc = mul6(fa, fb, acc);
    if (more) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { ea[e] = na[e]; eb[e] = nb[e]; }
    }
  }
  // ---- partial tiles of waves 1-3 -> LDS, wave 0 adds them in wave order
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  colsum += __shfl_xor(colsum, 32, 64);                           // the two m-halves of the lane's column
  if (kh == 0) rdb[wave][l31] = colsum;
  __syncthreads();
  if (wave != 0) return;
  const float sc = g.bits ? g.ascale : 1.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float v = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
    const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * kh, k = k0 + l31;
    if (n < g.N && k < g.K) g.dW[(size_t)n * g.lddw + k] = v * sc;
  }
  if (g.db && blockIdx.x == 0 && kh == 0 && nvalid) g.db[n0 + l31] = (((rdb[0][l31] + rdb[1][l31]) + rdb[2][l31]) + rdb[3][l31]) * sc;
}
}  // namespace
}  // namespace hoisdf

using namespace hoisdf;

static inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// rows below which the one-wave-per-tile form is the faster one (above: the tiled kernels have enough tiles to fill the chip)
extern "C" int hoisdf_linear_emu_small_max_rows(void) { return 2047; }

extern "C" int hoisdf_linear_emu_small_supported(const float* a, long lda, const float* W, long ldw, long M, int N, int K) {
  return a && W && M >= 1 && M <= hoisdf_linear_emu_small_max_rows() && N >= 1 && K >= 1 && al16p(a) && al16p(W) && lda % 4 == 0 &&
         ldw % 4 == 0 && N % 4 == 0 && K % 4 == 0;
}

extern "C" int hoisdf_linear_fwd_emu_small(const float* x, int ldx, const float* W, int ldw, const float* bias, float* y, int ldy,
                                           long M, int N, int K, int act, float drop_p, uint64_t seed, uint32_t* relu_bits,
                                           void* stream) {
  HOISDF_REQUIRE(M == 0 || (x && W && y), HOISDF_ERR_INVALID, "linear_fwd_emu_small: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N && drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID,
                 "linear_fwd_emu_small: bad sizes M=%ld N=%d K=%d", M, N, K);
  if (M == 0) return HOISDF_OK;
  HOISDF_REQUIRE(hoisdf_linear_emu_small_supported(x, ldx, W, ldw, M, N, K), HOISDF_ERR_INVALID,
                 "linear_fwd_emu_small: M <= %d, 16-byte aligned operands, leading dims / N / K multiples of 4", hoisdf_linear_emu_small_max_rows());
  SmallArgs g{};
  g.A = x; g.lda = ldx; g.W = W; g.ldw = ldw; g.C = y; g.ldc = ldy; g.bias = bias;
  g.bits_out = relu_bits; g.ldbits_out = (N + 31) / 32;
  g.M = (int)M; g.N = N; g.K = K; g.act = act; g.drop_p = drop_p; g.inv_keep = 1.f / (1.f - drop_p);
  g.thresh = drop_threshold(drop_p); g.seed = seed;
  const dim3 grid((unsigned)cdiv(N, 32), (unsigned)cdiv((int)M, 32));
  hipLaunchKernelGGL(emu_small_kernel<false>, grid, dim3(256), 0, as_stream(stream), g);
  return check_launch("linear_fwd_emu_small");
}

extern "C" int hoisdf_linear_bwd_input_emu_small(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* W,
                                                 int ldw, float* dx, int lddx, long M, int N, int K, int accumulate, void* stream) {
  HOISDF_REQUIRE(M == 0 || (dy && W && dx), HOISDF_ERR_INVALID, "linear_bwd_input_emu_small: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldw >= K && lddx >= K && drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID,
                 "linear_bwd_input_emu_small: bad sizes");
  if (M == 0) return HOISDF_OK;
  HOISDF_REQUIRE(hoisdf_linear_emu_small_supported(dy, lddy, W, ldw, M, N, K), HOISDF_ERR_INVALID,
                 "linear_bwd_input_emu_small: M <= %d, 16-byte aligned operands, leading dims / N / K multiples of 4", hoisdf_linear_emu_small_max_rows());
  SmallArgs g{};
  g.A = dy; g.lda = lddy; g.W = W; g.ldw = ldw; g.C = dx; g.ldc = lddx;
  g.abits = relu_bits; g.ldbits = (N + 31) / 32; g.ascale = relu_bits ? 1.f / (1.f - drop_p) : 1.f;
  g.M = (int)M; g.N = N; g.K = K; g.accumulate = accumulate; g.inv_keep = 1.f;
  const dim3 grid((unsigned)cdiv(K, 32), (unsigned)cdiv((int)M, 32));
  hipLaunchKernelGGL(emu_small_kernel<true>, grid, dim3(256), 0, as_stream(stream), g);
  return check_launch("linear_bwd_input_emu_small");
}

extern "C" int hoisdf_linear_bwd_weight_emu_small(const float* dy, int lddy, const uint32_t* relu_bits, float drop_p, const float* x,
                                                  int ldx, float* dW, int lddw, float* db, long M, int N, int K, void* stream) {
  HOISDF_REQUIRE(dW && (M == 0 || (dy && x)), HOISDF_ERR_INVALID, "linear_bwd_weight_emu_small: null pointer");
  HOISDF_REQUIRE(M > 0 && M <= hoisdf_linear_emu_small_max_rows() && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K &&
                     drop_p >= 0.f && drop_p < 1.f,
                 HOISDF_ERR_INVALID, "linear_bwd_weight_emu_small: bad sizes");
  SmallDwArgs g{};
  g.dy = dy; g.lddy = lddy; g.x = x; g.ldx = ldx; g.bits = relu_bits; g.ldbits = (N + 31) / 32;
  g.ascale = relu_bits ? 1.f / (1.f - drop_p) : 1.f;
  g.dW = dW; g.lddw = lddw; g.db = db; g.M = (int)M; g.N = N; g.K = K;
  const dim3 grid((unsigned)cdiv(K, 32), (unsigned)cdiv(N, 32));
  hipLaunchKernelGGL(emu_small_dw_kernel, grid, dim3(256), 0, as_stream(stream), g);
  return check_launch("linear_bwd_weight_emu_small");
}
