// Exact-fp32 GEMM family on the gfx950 f32 MFMA pipe (v_mfma_f32_32x32x2_f32, 157 TF peak,
// bit-for-bit an fmaf chain).  One template serves the three contractions a linear layer needs:
//   forward      y  = x . W^T      A[m][k] k-contiguous, B[n][k] k-contiguous
//   grad input   dx = dy . W       A[m][k] k-contiguous, B[k][n] n-contiguous
//   grad weight  dW = dy^T . x     A[k][m] m-contiguous, B[k][n] n-contiguous   (split-K, atomics)
// Block tile 128x128x32, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator
// VGPRs).  LDS holds both operands k-major ([k][m]) so the MFMA fragment read is always one
// conflict-free ds_read_b32 per operand per k-pair; only the global->LDS staging differs with
// the operand's memory orientation.  Register-staged double buffering: the next tile's global
// loads are in flight during the 64 MFMAs of the current one; one barrier per k-tile.
// Block ids are remapped so that the tiles sharing an A row-panel sit on one XCD (shared L2).
#include "common.h"

namespace hoisdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int LDS_KC = BM + 1;   // k-contiguous source: transposing ds_write_b32, stride = 1 mod 32
constexpr int LDS_MC = BM + 4;   // m-contiguous source: ds_write_b128, 16-byte aligned rows

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  int lda, ldb, ldc;
  int act;
  float drop_p, inv_keep;
  uint64_t seed;
  int splitk, k_per_split, atomic_out;
  int tiles_m, tiles_n;
  int vecA, vecB;
};

// Stage one 128 x 32 operand tile from global memory into registers (4 x float4 per thread).
// KC = true : element (r, k) at src[r*ld + k]   (k contiguous)
// KC = false: element (r, k) at src[k*ld + r]   (r contiguous)
template <bool KC>
__device__ __forceinline__ void stage_load(float4 (&reg)[4], const float* __restrict__ src, int ld,
                                           int r0, int R, int k0, int kend, int vec, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      int r = r0 + (tid >> 3) + 32 * i;
      int k = k0 + (tid & 7) * 4;
      if (r < R) {
        const float* p = src + (size_t)r * ld + k;
        if (vec && k + 3 < kend) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (k + 0 < kend) v.x = p[0];
          if (k + 1 < kend) v.y = p[1];
          if (k + 2 < kend) v.z = p[2];
          if (k + 3 < kend) v.w = p[3];
        }
      }
    } else {
      int k = k0 + (tid >> 5) + 8 * i;
      int r = r0 + (tid & 31) * 4;
      if (k < kend) {
        const float* p = src + (size_t)k * ld + r;
        if (vec && r + 3 < R) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (r + 0 < R) v.x = p[0];
          if (r + 1 < R) v.y = p[1];
          if (r + 2 < R) v.z = p[2];
          if (r + 3 < R) v.w = p[3];
        }
      }
    }
    reg[i] = v;
  }
}

template <bool KC>
__device__ __forceinline__ void stage_store(const float4 (&reg)[4], float* __restrict__ lds, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (KC) {
      int r = (tid >> 3) + 32 * i;
      int k = (tid & 7) * 4;
      lds[(k + 0) * LDS_KC + r] = reg[i].x;
      lds[(k + 1) * LDS_KC + r] = reg[i].y;
      lds[(k + 2) * LDS_KC + r] = reg[i].z;
      lds[(k + 3) * LDS_KC + r] = reg[i].w;
    } else {
      int k = (tid >> 5) + 8 * i;
      int r = (tid & 31) * 4;
      *reinterpret_cast<float4*>(&lds[k * LDS_MC + r]) = reg[i];
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmArgs g) {
  constexpr int SA = A_KC ? LDS_KC : LDS_MC;
  constexpr int SB = B_KC ? LDS_KC : LDS_MC;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * SA + 2 * BK * SB];
  float* As = lds;
  float* Bs = lds + 2 * BK * SA;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int ntile = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  const int split = bid / ntile;
  int t = xcd_remap(bid - split * ntile, ntile);
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rb[4];
  if (nk > 0) {
    stage_load<A_KC>(ra, g.A, g.lda, m0, g.M, kbeg, kend, g.vecA, tid);
    stage_load<B_KC>(rb, g.B, g.ldb, n0, g.N, kbeg, kend, g.vecB, tid);
    stage_store<A_KC>(ra, As, tid);
    stage_store<B_KC>(rb, Bs, tid);
  }
  __syncthreads();

  const int arow = wm * 64 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);
  const int khalf = lane >> 5;

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      stage_load<A_KC>(ra, g.A, g.lda, m0, g.M, kbeg + (kt + 1) * BK, kend, g.vecA, tid);
      stage_load<B_KC>(rb, g.B, g.ldb, n0, g.N, kbeg + (kt + 1) * BK, kend, g.vecB, tid);
    }
    const float* as = As + cur * BK * SA;
    const float* bs = Bs + cur * BK * SB;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a0 = as[(kk + khalf) * SA + arow];
      float a1 = as[(kk + khalf) * SA + arow + 32];
      float b0 = bs[(kk + khalf) * SB + brow];
      float b1 = bs[(kk + khalf) * SB + brow + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      stage_store<A_KC>(ra, As + (cur ^ 1) * BK * SA, tid);
      stage_store<B_KC>(rb, Bs + (cur ^ 1) * BK * SB, tid);
    }
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    if (col >= g.N) continue;
    const float bv = (g.bias != nullptr && split == 0) ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row >= g.M) continue;
        float v = acc[i][j][r] + bv;
        if (g.act == 1) v = fmaxf(v, 0.f);
        if (g.drop_p > 0.f)
          v *= drop_scale(g.drop_p, g.inv_keep, g.seed, (uint64_t)row * (uint64_t)g.N + col);
        float* dst = g.C + (size_t)row * g.ldc + col;
        if (g.atomic_out) atomicAdd(dst, v);
        else *dst = v;
      }
    }
  }
}

static int aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool A_KC, bool B_KC>
static int launch_gemm(GemmArgs g, hipStream_t st) {
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, BN);
  // vector (16-byte) global loads need 16-byte aligned rows along the contiguous dimension
  // (a k-chunk that crosses the end of the contraction range falls back to guarded scalars)
  g.vecA = aligned16(g.A) && (g.lda % 4 == 0) && (A_KC ? (g.k_per_split % 4 == 0) : true);
  g.vecB = aligned16(g.B) && (g.ldb % 4 == 0) && (B_KC ? (g.k_per_split % 4 == 0) : true);
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n * g.splitk));
  hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC>), grid, dim3(NT), 0, st, g);
  return check_launch("gemm_f32");
}

// column sums of dy[M][N] -> db[N] (atomic accumulate); one block handles 256 rows x 64 cols
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, int ld, long M, int N,
                                                     float* __restrict__ db) {
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  long r0 = (long)blockIdx.y * 256;
  float s = 0.f;
  if (col < N) {
    for (int i = w; i < 256; i += 4) {
      long r = r0 + i;
      if (r < M) s += dy[(size_t)r * ld + col];
    }
  }
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && col < N) atomicAdd(&db[col], part[0][threadIdx.x] + part[1][threadIdx.x] +
                                                  part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__global__ void relu_dropout_bwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ dy,
                                        int lddy, float* __restrict__ dp, int ldd, long M, int N,
                                        float inv_keep) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = M * (long)N;
  for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long m = idx / N;
    int n = (int)(idx - m * N);
    float yv = y[(size_t)m * ldy + n];
    dp[(size_t)m * ldd + n] = yv > 0.f ? dy[(size_t)m * lddy + n] * inv_keep : 0.f;
  }
}

}  // namespace hoisdf

using namespace hoisdf;

extern "C" int hoisdf_linear_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias,
                                 float* y, int ldy, long M, int N, int K, int act, float drop_p,
                                 uint64_t seed, void* stream) {
  HOISDF_REQUIRE(x && W && y, HOISDF_ERR_INVALID, "linear_fwd: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, HOISDF_ERR_INVALID,
                 "linear_fwd: bad sizes M=%ld N=%d K=%d ldx=%d ldw=%d ldy=%d", M, N, K, ldx, ldw, ldy);
  HOISDF_REQUIRE(drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID, "linear_fwd: drop_p=%f", drop_p);
  HOISDF_REQUIRE(M < (1L << 31), HOISDF_ERR_INVALID, "linear_fwd: M too large");
  if (M == 0) return HOISDF_OK;
  GemmArgs g{};
  g.A = x; g.B = W; g.C = y; g.bias = bias;
  g.M = (int)M; g.N = N; g.K = K; g.lda = ldx; g.ldb = ldw; g.ldc = ldy;
  g.act = act; g.drop_p = drop_p; g.inv_keep = 1.f / (1.f - drop_p); g.seed = seed;
  g.splitk = 1; g.k_per_split = ((K + BK - 1) / BK) * BK; g.atomic_out = 0;
  return launch_gemm<true, true>(g, as_stream(stream));
}

extern "C" int hoisdf_linear_bwd_input(const float* dy, int lddy, const float* W, int ldw, float* dx,
                                       int lddx, long M, int N, int K, void* stream) {
  HOISDF_REQUIRE(dy && W && dx, HOISDF_ERR_INVALID, "linear_bwd_input: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldw >= K && lddx >= K && M < (1L << 31),
                 HOISDF_ERR_INVALID, "linear_bwd_input: bad sizes");
  if (M == 0) return HOISDF_OK;
  GemmArgs g{};
  // dx[m][k] = sum_n dy[m][n] W[n][k]: contraction index n; "B" is W read as [n][k] = [contract][out]
  g.A = dy; g.B = W; g.C = dx; g.bias = nullptr;
  g.M = (int)M; g.N = K; g.K = N; g.lda = lddy; g.ldb = ldw; g.ldc = lddx;
  g.act = 0; g.drop_p = 0.f; g.inv_keep = 1.f; g.seed = 0;
  g.splitk = 1; g.k_per_split = ((N + BK - 1) / BK) * BK; g.atomic_out = 0;
  return launch_gemm<true, false>(g, as_stream(stream));
}

extern "C" int hoisdf_linear_bwd_weight(const float* dy, int lddy, const float* x, int ldx, float* dW,
                                        int lddw, float* db, long M, int N, int K, void* stream) {
  HOISDF_REQUIRE(dy && x && dW, HOISDF_ERR_INVALID, "linear_bwd_weight: null pointer");
  HOISDF_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K && M < (1L << 31),
                 HOISDF_ERR_INVALID, "linear_bwd_weight: bad sizes");
  if (M == 0) return HOISDF_OK;
  hipStream_t st = as_stream(stream);
  GemmArgs g{};
  // dW[n][k] = sum_m dy[m][n] x[m][k]: out rows n, out cols k, contraction m
  g.A = dy; g.B = x; g.C = dW; g.bias = nullptr;
  g.M = N; g.N = K; g.K = (int)M; g.lda = lddy; g.ldb = ldx; g.ldc = lddw;
  g.act = 0; g.drop_p = 0.f; g.inv_keep = 1.f; g.seed = 0;
  int tiles = cdiv(N, BM) * cdiv(K, BN);
  int ksteps = cdiv(M, BK);
  int want = tiles >= 1024 ? 1 : cdiv(1024, tiles);
  int splitk = want < 1 ? 1 : want;
  if (splitk > ksteps / 4) splitk = ksteps / 4 > 0 ? ksteps / 4 : 1;   // >= 4 k-tiles per split
  int kper = cdiv(ksteps, splitk) * BK;
  splitk = cdiv(M, kper);
  g.splitk = splitk; g.k_per_split = kper; g.atomic_out = 1;
  int rc = launch_gemm<false, false>(g, st);
  if (rc) return rc;
  if (db) {
    dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 256));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, dy, lddy, M, N, db);
    return check_launch("colsum");
  }
  return HOISDF_OK;
}

extern "C" int hoisdf_relu_dropout_bwd(const float* y, int ldy, const float* dy, int lddy, float* dpre,
                                       int ldd, long M, int N, float drop_p, void* stream) {
  HOISDF_REQUIRE(y && dy && dpre && M >= 0 && N > 0 && drop_p >= 0.f && drop_p < 1.f, HOISDF_ERR_INVALID,
                 "relu_dropout_bwd: bad arguments");
  if (M == 0) return HOISDF_OK;
  long total = M * (long)N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), y, ldy, dy,
                     lddy, dpre, ldd, M, N, 1.f / (1.f - drop_p));
  return check_launch("relu_dropout_bwd");
}
