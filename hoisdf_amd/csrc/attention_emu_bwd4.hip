// fp32-EMULATED attention backward, round-5 form: dK / dV / dQ in ONE pass (5 GEMM-equivalents, six bf16 products per product, f32
// accumulation - the arithmetic of attention_emu.hip), no atomics, run-to-run identical.
// reference: common/nets/transformer.py:269,286-302 (autograd backward of nn.MultiheadAttention inside the encoder layers).
//
// Why a new form: the round-3 kernel (emu_attn_bwd_stag_kernel, 8 waves x 16 keys on v_mfma_f32_16x16x32_bf16) reads one A fragment
// from LDS per two 16-cycle MFMAs - with four SIMDs that is the full 256 B / clk of the LDS - and pays a workgroup barrier per
// half-step (PMC round 4: 53 % of the wave cycles parked, MFMA pipe 44 % busy).  Here:
//   * block = 128 keys in FOUR waves of 32 keys, one wave per SIMD (up to 512 VGPR + AGPR), every contraction on
//     v_mfma_f32_32x32x16_bf16: a 1 KB fragment feeds two 32-cycle MFMAs - a quarter of the LDS bytes per MFMA cycle;
//   * K, V fragments (B operands of S / dP), the K^T fragments of the wave's dQ job and the dK / dV accumulators stay in registers;
//   * the wave's own softmax / dropout / split VALU work is pinned behind its OWN MFMAs (on this part a wave's VALU does not hide
//     under another wave's MFMAs, profiles/r03_mfma_valu_overlap.txt), software-pipelined over the query tiles:
//         iteration t:  S(t) | dP(t) | dQ(t - 1) | dV(t) | dK(t)          (24 MFMAs each, attn_bwd4_phase.inc)
//     with ONE workgroup barrier per query tile.  That needs the Q / dO row tiles triple-buffered and the dS^T exchange tile
//     double-buffered (156 KB of LDS).
// Layouts (c = lane & 31, h = lane >> 5; accumulator register r of a 32 x 32 tile holds row CR(r, h) = (r & 3) + 8 (r >> 2) + 4 h, column c):
//   S, dP [32 q x 32 keys]: A = Q / dO rows (row c, d = 16 j + 8 h ..), B = kf / vf (key c, the same d) -> lane = key, 16 queries
//   dV^T, dK^T [64 d x 32 keys] += dO^T / Q^T [d x q] . Pd / dS [q x key]: B = the bf16 triples of Pd / dS STRAIGHT from the lane's
//       registers (k-slot 8 h + i of step jj <-> q = 16 jj + 4 h + (i & 3) + 8 (i >> 2)), A from the same row tiles through
//       ds_read_b64_tr_b16 (two reads of 4 queries x 16 d per fragment)
//   dQ^T [64 d x 32 q] = K^T [d x key] . dS^T [key x q] over the block's 128 keys: every wave writes its dS^T rows (its 32 keys) into
//       the shared tile T^T[128 keys][32 q]; wave w then contracts the d half (w & 1) with the key half (w >> 1) - A = resident K^T
//       fragments, B = T^T through the transpose read - and the two key halves are added through a second LDS tile by all four
//       waves (8 query rows each, 256-byte rows to HBM).  Small products and x0 y0 in separate accumulators.
//   dQ goes to the per-key-block partial buffer [kb][bh][q][64] of the round-3 form; emu_attn_dq_reduce_kernel sums it.
#include <stdlib.h>

#include "attention_emu.h"

namespace hoisdf {
using emu_attn::EmuAttn;
using emu_attn::emu_block;

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int D = 64;
constexpr float LN2 = 0.6931471805599453f;
constexpr int QD_PLANE = 32 * 64;            // bf16 per plane tile [32 q][64 d]
constexpr int QD_BUF = 6 * QD_PLANE;         // one staging buffer: Q planes 0-2, dO planes 0-2 (24 KB)
constexpr int TT_PLANE = 128 * 32;           // bf16 per dS^T plane [128 keys][32 q]
constexpr int TT_BUF = 3 * TT_PLANE;         // 24 KB
constexpr int X_BUF = 2 * 32 * 64;           // floats per dQ exchange buffer [key half][32 q][64 d] (16 KB)
constexpr int TT0 = 3 * QD_BUF;              // bf16 offset of the T^T buffers
constexpr int X0_BYTES = (3 * QD_BUF + 2 * TT_BUF) * 2;
constexpr int ST0_BYTES = X0_BYTES + 2 * X_BUF * 4;
constexpr unsigned B4_LDS_BYTES = ST0_BYTES + 3 * 64 * 4;      // 156 416 B

#define MB(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a_), (b_), (c_), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)

// 16-byte chunk `ch` of row `r` of a [32][64] bf16 row tile.  The XOR (bits: r1, r2, r1 ^ r3) serves the three access patterns
// without bank conflicts (brute-forced against the guide's lane groups): the 16-byte fragment reads of S / dP (16 lanes = rows
// {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} at one chunk), the transpose reads (32 lanes = 4 consecutive rows x 4 chunks) and the
// staging writes (8 lanes = one row)
__device__ __forceinline__ int qd_swz(int r) { return ((r >> 1) & 3) | ((((r >> 1) ^ (r >> 3)) & 1) << 2); }
__device__ __forceinline__ int qd_off(int r, int ch) { return r * 64 + ((ch ^ qd_swz(r)) << 3); }

__device__ __forceinline__ bf16x8 tr8(const __bf16* lo, const __bf16* hi) {
  const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lo));
  const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(hi));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ uint32_t cvt2(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 unpack2(uint32_t w) { return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)}; }
}  // namespace

// CHAIN (experiment, HOISDF_EMU_ATTN_BWD_CHAIN=G, off by default): fewer dQ partials.  The key blocks of a (b, head) run side by side
// on ONE XCD (emu_block) and walk the query tiles at the same pace; within a chain of G consecutive key blocks, block kb adds its
// contribution of a query tile to a running sum that block kb - 1 has already added to - wave w of block kb waits for wave w of block
// kb - 1 through a counter in L2 (same XCD: plain stores + sc1 loads, no fences), reads the 8 rows x 64 d it owns, adds, writes.
// Fixed order = run-to-run identical, and with G = 16 bit-identical to the reduce pass (same association).  Measured (B = 32,
// S = 2048, whole call): G = 16 2.150 ms, G = 8 2.16, G = 4 2.18, G = 2 2.23, partials + reduce 2.155: the 1.07 GB of partial traffic
// is not what bounds the kernel, and a chain costs its members a start-up skew (one L2 round trip per stage, idle time with one
// workgroup per CU).  Kept as the measured answer to "halve the dQ partials" (profiles/r05_attn_bwd_dq_chain_ab.txt); correctness
// relies on the observed block -> XCD placement, which HIP does not promise - hence not the default.
// (Blocks wait on LOWER block indices only, which the dispatcher starts first; the wait is bounded all the same.)
template <bool DROP, bool CHAIN>
__global__ __launch_bounds__(256, 1) void emu_attn_bwd4_kernel(EmuAttn a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  float* const xbuf = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + X0_BYTES);
  float* const stats = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + ST0_BYTES);     // [3][lse 32 (log2 domain) | delta 32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5, jq = lane & 15, b1 = (jq >> 3) & 1;
  int ktile, bh;
  const int nkb = (a.Lk + 127) / 128;
  if (!emu_block(nkb, a.B * a.H, ktile, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int key = ktile * 128 + wave * 32 + c;
  const bool kvalid = key < a.kv_len;
  const int nq = ktile * 128 < a.kv_len ? (a.Lq + 31) / 32 : 0;
  const int dhalf = wave & 1, khalf = wave >> 1;             // this wave's dQ job: d half x key half

  // ---- resident operands ------------------------------------------------------------------------------------------------
  bf16x8 kf[4][3], vf[4][3], ktf[4][3];
  {
    const size_t ro = ((size_t)bh * a.Lkp + key) * D;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kf[j][p] = *reinterpret_cast<const bf16x8*>(a.k[p] + ro + 16 * j + 8 * h);
        vf[j][p] = *reinterpret_cast<const bf16x8*>(a.v[p] + ro + 16 * j + 8 * h);
      }
    // K^T fragments of the dQ job: row d = 32 dhalf + c, k-slots = keys 64 khalf + 16 ks + 8 h + i (2-byte gathers, once per block)
    const size_t ko = ((size_t)bh * a.Lkp + ktile * 128 + 64 * khalf + 8 * h) * D + 32 * dhalf + c;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i)                      // (keys past kv_len contribute nothing to dQ: their dS rows are not masked)
          t[i] = ktile * 128 + 64 * khalf + 16 * ks + 8 * h + i < a.kv_len ? a.k[p][ko + (size_t)(16 * ks + i) * D] : (__bf16)0.f;
        ktf[ks][p] = t;
      }
  }
  // (an empty statement with a "+a" operand re-defines the value IN the accumulator file: without it hipcc keeps a fragment that
  // VALU instructions assembled in VGPRs and copies it over in front of every MFMA that names it)
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" : "+a"(kf[j][p]));
      asm volatile("" : "+a"(vf[j][p]));
      asm volatile("" : "+a"(ktf[j][p]));
    }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

  // ---- per-lane LDS offsets (bf16 elements unless noted) --------------------------------------------------------------------
  const int fswz = qd_swz(c);
  int aoff[4];                               // S / dP A fragments: row c, chunk 2 j + h
#pragma unroll
  for (int j = 0; j < 4; ++j) aoff[j] = c * 64 + (((2 * j + h) ^ fswz) << 3);
  // transpose reads of the row tiles: row 16 jj + 8 s + 4 h + (jq >> 2), chunk 4 mt + 2 (c >> 4) + ((jq >> 1) & 1), half jq & 1;
  // the swizzle of that row is b1 | h << 1 | (b1 ^ s) << 2, so bit 2 of the swizzled chunk is mt ^ s ^ b1: two lane bases, e = mt ^ s
  int trb[2];
  {
    const int low2 = ((((c >> 4) ^ h) & 1) << 1) | ((((jq >> 1) & 1) ^ b1) & 1);
#pragma unroll
    for (int e = 0; e < 2; ++e) trb[e] = (4 * h + (jq >> 2)) * 64 + ((e ^ b1) << 5) + (low2 << 3) + (jq & 1) * 4;
  }
  // T^T[128 keys][32 q] (64-byte rows, 8-byte granules of 4 q at position qg ^ ((key >> 1) & 7)):
  //   reads (B fragments of dQ^T): key row 64 khalf + 16 ks + 8 h + 4 s + (jq >> 2), granule 4 (c >> 4) + (jq & 3)
  int tqb[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int kr = 64 * khalf + 8 * h + 4 * s2 + (jq >> 2);
    tqb[s2] = kr * 32 + (((4 * (c >> 4) + (jq & 3)) ^ ((kr >> 1) & 7)) << 2);
  }
  //   writes: key row 32 wave + c, granule 2 g + h -> tw0 ^ (8 g)
  const int tw0 = (32 * wave + c) * 32 + ((h ^ ((c >> 1) & 7)) << 2);
  // X[khalf][32 q][64 d] floats (256-byte rows, 16-byte granules at position gi ^ (q & 7)):
  //   writes: row c, granule 8 dhalf + 2 g + h -> xw0 ^ (8 g);  reads: rows 8 wave + 4 i + (lane >> 4), granule lane & 15
  const int xw0 = khalf * 2048 + c * 64 + (8 * dhalf << 2) + ((h ^ (c & 7)) << 2);
  const int xr0 = (8 * wave + (lane >> 4)) * 64 + (((lane & 15) ^ (lane >> 4)) << 2);
  const int xr1 = (xr0 + 256) ^ 16;
  const int st_o = qd_off(tid >> 3, tid & 7);               // staging: thread -> row tid >> 3, chunk tid & 7 of each of the six planes

  // ---- staging state ----------------------------------------------------------------------------------------------------------
  u32x4 sg[6];
  float rstat = 0.f;
  const size_t rowbase = (size_t)bh * a.Lqp * D;
  const __bf16* const qb0 = a.q[0] + rowbase; const __bf16* const qb1 = a.q[1] + rowbase; const __bf16* const qb2 = a.q[2] + rowbase;
  const __bf16* const db0 = a.d[0] + rowbase; const __bf16* const db1 = a.d[1] + rowbase; const __bf16* const db2 = a.d[2] + rowbase;
  unsigned goff = (unsigned)((tid >> 3) * D + (tid & 7) * 8);
  const float* const stat_src = (tid < 32 ? a.lse_in : a.delta) + (size_t)bh * a.Lq;
#define B4_LOADP(i_) do { sg[i_] = *reinterpret_cast<const u32x4*>(((i_) == 0 ? qb0 : (i_) == 1 ? qb1 : (i_) == 2 ? qb2 : (i_) == 3 ? db0 : (i_) == 4 ? db1 : db2) + goff); } while (0)
// (GHOST_: a tile past the last one - the pipeline's drain iteration - gets lse = +inf, delta = 0: P = dS = 0)
#define B4_LOADS(QT_, GHOST_)                                                                                          \
  do {                                                                                                                 \
    if (tid < 64) {                                                                                                    \
      const int q_ = (QT_) * 32 + (tid & 31);                                                                          \
      rstat = (q_ < a.Lq && !(GHOST_)) ? stat_src[q_] : (tid < 32 ? INFINITY : 0.f);                                   \
    }                                                                                                                  \
  } while (0)
#define B4_STOREP(i_, BUF_) do { *reinterpret_cast<u32x4*>(lds + (BUF_) + (i_) * QD_PLANE + st_o) = sg[i_]; } while (0)
#define B4_STORES(SB_) do { if (tid < 64) stats[(SB_) + tid] = rstat; } while (0)

  // CHAIN: key blocks [G grp, G grp + G) of a (b, head) form one chain; its running sum is partial number grp
  const int nkb_live = (a.kv_len + 127) / 128;                  // key blocks that take part (the others have nq = 0)
  const int G = CHAIN ? a.chain_group : 1, grp = ktile / G;
  const bool ch_first = ktile - grp * G == 0, ch_last = ktile - grp * G == G - 1 || ktile == nkb_live - 1;
  const bool ch_direct = CHAIN && nkb_live <= G;                // a single chain: its last block writes the caller's dq (scaled) itself
  float* const part = a.dq_part + ((size_t)(CHAIN ? grp : ktile) * a.B * a.H + bh) * a.Lq * D;
  int* const my_flag = CHAIN ? a.dq_flags + ((size_t)bh * nkb + ktile) * 4 + wave : nullptr;
  const int* const up_flag = CHAIN && !ch_first ? a.dq_flags + ((size_t)bh * nkb + ktile - 1) * 4 + wave : nullptr;
  float* const dq_out = CHAIN ? a.dq + (size_t)b * a.Lq * a.ldq + head * D : nullptr;
  if (nq > 0) {
    // ---- prologue: tiles 0 and 1 staged, tile 2 in registers --------------------------------------------------------------
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int qt = min(pt, nq - 1);
      goff = (unsigned)(qt * 32 * D + (tid >> 3) * D + (tid & 7) * 8);
#pragma unroll
      for (int i = 0; i < 6; ++i) B4_LOADP(i);
      B4_LOADS(qt, pt >= nq);
#pragma unroll
      for (int i = 0; i < 6; ++i) B4_STOREP(i, pt * QD_BUF);
      B4_STORES(pt * 64);
    }
    int qt_next = min(2, nq - 1);
    goff = (unsigned)(qt_next * 32 * D + (tid >> 3) * D + (tid & 7) * 8);
#pragma unroll
    for (int i = 0; i < 6; ++i) B4_LOADP(i);
    B4_LOADS(qt_next, 2 >= nq);
    __syncthreads();

    // rotating buffers: Q / dO tiles and statistics (cur = tile t, nxt = t + 1, fre = the one tile t + 2 is staged into), T^T and X by parity
    int qd_cur = 0, qd_nxt = QD_BUF, qd_fre = 2 * QD_BUF;
    int st_cur = 0, st_nxt = 64, st_fre = 128;
    int tt_w = TT0, tt_r = TT0 + TT_BUF;                    // dS^T(t) is written to tt_w, dS^T(t - 1) read from tt_r
    int x_a = 0, x_b = X_BUF;                               // X[t & 1] = x_a: read by the output of tile t - 2; x_b: written with dQ(t - 1)
    // dropout: hash input of element r = hb + CRc(r) * G1 (common.h drop_rowkey / drop_hash), hb moves by 32 G1 per query tile
    uint32_t hb = 0;
    const uint32_t dthr = a.thresh & 0xffff0000u;
    const int hsh = (key & 1) ? 0 : 16;
    if (DROP) hb = drop_rowkey(a.seed, (uint32_t)(bh * a.Lq + 4 * h)) + (uint32_t)(key >> 1) * 0x9E3779B9U;

    f32x16 s, dp, dq;
    f32x4 xo0[2], xo1[2], xprev[2];
    bf16x8 fr[2][3], fq0[4];
    u32x4 pwv[3][2], gwv[3][2];
    f32x2 pe[8], pd[8], xx[8], ff[8];
    float dsc[16], lq[16], dl[16];
    uint32_t hx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dsc[r] = 1.f;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) { pwv[p][i] = u32x4{0u, 0u, 0u, 0u}; gwv[p][i] = u32x4{0u, 0u, 0u, 0u}; }

    // ---- MFMAs through asm: the operand FILE is chosen per statement (hipcc picks one accumulator form per function and then
    // copies accumulators between the files, 200 v_accvgpr_mov / _read per query tile in the builtin version of this kernel):
    // S / dP / dQ accumulate in VGPRs (the softmax reads them), dK / dV and the resident K, V, K^T fragments live in AGPRs.
    // hipcc pads no hazard of an asm statement: every reader of an accumulator sits >= 2 MFMAs behind the chain's last product
    // (an 8-pass result needs 12 states), every VALU-written operand (pw, gw) is produced a phase ahead of its MFMAs.
#define MFMA_SP(acc_, a_, b_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "a"(b_))
#define MFMA_SP0(acc_, a_, b_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc_) : "v"(a_), "a"(b_))
#define MFMA_Q(acc_, a_, b_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc_) : "a"(a_), "v"(b_))
#define MFMA_Q0(acc_, a_, b_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc_) : "a"(a_), "v"(b_))
#define MFMA_VK(acc_, a_, b_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc_) : "v"(a_), "v"(b_))
    // ---- fragment reads -------------------------------------------------------------------------------------------------------
#define FRQ(p_, j_) (*reinterpret_cast<const bf16x8*>(lds + qd_cur + aoff[j_] + (p_) * QD_PLANE))
#define FRD(p_, j_) (*reinterpret_cast<const bf16x8*>(lds + qd_cur + aoff[j_] + (3 + (p_)) * QD_PLANE))
#define FRQN(p_, j_) (*reinterpret_cast<const bf16x8*>(lds + qd_nxt + aoff[j_] + (p_) * QD_PLANE))
#define FRDN(p_, j_) (*reinterpret_cast<const bf16x8*>(lds + qd_nxt + aoff[j_] + (3 + (p_)) * QD_PLANE))
#define FRA(pl_, jj_, mt_) tr8(lds + qd_cur + trb[(mt_)] + (pl_) * QD_PLANE + (16 * (jj_)) * 64, \
                               lds + qd_cur + trb[(mt_) ^ 1] + (pl_) * QD_PLANE + (16 * (jj_) + 8) * 64)
#define FRT(p_, ks_) tr8(lds + tt_r + tqb[0] + (p_) * TT_PLANE + (16 * (ks_)) * 32, lds + tt_r + tqb[1] + (p_) * TT_PLANE + (16 * (ks_)) * 32)
#define PWF(p_, jj_) __builtin_bit_cast(bf16x8, pwv[p_][jj_])
#define GWF(p_, jj_) __builtin_bit_cast(bf16x8, gwv[p_][jj_])
    // ---- units (q_ = element quad: accumulator registers 4 q .. 4 q + 3 = pairs 2 q, 2 q + 1 = queries 8 q + 4 h .. + 3) ------------
    // (the empty asm statements pin a unit's results HERE: the optimiser otherwise sinks them to their consumers)
#define PIN2(x_) asm volatile("" : "+v"(x_))
    // S phase: output of dQ(t - 2) (the two key halves added, 8 query rows per wave); dropout decisions of tile t
#define XOL(i_)                                                                                                        \
  do {                                                                                                                 \
    xo0[i_] = *reinterpret_cast<const f32x4*>(xbuf + x_a + ((i_) ? xr1 : xr0));                                        \
    xo1[i_] = *reinterpret_cast<const f32x4*>(xbuf + x_a + 2048 + ((i_) ? xr1 : xr0));                                 \
  } while (0)
#define XOS(i_)                                                                                                        \
  do {                                                                                                                 \
    const int q_ = (t - 2) * 32 + 8 * wave + 4 * (i_) + (lane >> 4);                                                   \
    if (t >= 2 && q_ < a.Lq) {                                                                                         \
      f32x4 v_ = xo0[i_] + xo1[i_];                                                                                    \
      if (CHAIN) {                                                                                                     \
        if (!ch_first) v_ = xprev[i_] + v_;                                                                            \
        if (ch_last && ch_direct) *reinterpret_cast<f32x4*>(dq_out + (size_t)q_ * a.ldq + 4 * (lane & 15)) = v_ * 0.125f; \
        else *reinterpret_cast<f32x4*>(part + (unsigned)(q_ * D + 4 * (lane & 15))) = v_;                              \
      } else {                                                                                                         \
        *reinterpret_cast<f32x4*>(part + (unsigned)(q_ * D + 4 * (lane & 15))) = v_;                                   \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)
// CHAIN: wait until the same wave of the previous key block has added tile t - 2, then fetch its running sum (sc1: served by L2,
// never a stale L1 line); later, once this wave's own stores are acknowledged, publish the count
#define XOP()                                                                                                          \
  do {                                                                                                                 \
    if (CHAIN && !ch_first && t >= 2) {                                                                                \
      int spins_ = 0;                                                                                                  \
      while (__hip_atomic_load(up_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t - 1 && ++spins_ < (1 << 20))   \
        __builtin_amdgcn_s_sleep(2);                                                                                   \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                               \
        const int q_ = (t - 2) * 32 + 8 * wave + 4 * i_ + (lane >> 4);                                                 \
        const float* src_ = part + (unsigned)(min(q_, a.Lq - 1) * D + 4 * (lane & 15));                                \
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(xprev[i_]) : "v"(src_) : "memory");                  \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)
#define XOW() do { if (CHAIN && !ch_first && t >= 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(xprev[0]), "+v"(xprev[1]) : : "memory"); } while (0)
#define XSIG()                                                                                                         \
  do {                                                                                                                 \
    if (CHAIN && !ch_last && t >= 2) {                                                                                 \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
      if (lane == 0) __hip_atomic_store(my_flag, t - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                   \
    }                                                                                                                  \
  } while (0)
#define CRC(r_) (((r_) & 3) + 8 * ((r_) >> 2))
#define HA(q_) do { if (DROP) { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { const int r_ = 4 * (q_) + i_; uint32_t x_ = hbn + (uint32_t)CRC(r_) * 0x85EBCA77U; x_ ^= x_ >> 15; hx[r_] = x_; PIN2(hx[r_]); } } } while (0)
#define HB(q_) do { if (DROP) { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { const int r_ = 4 * (q_) + i_; hx[r_] *= 0x2C1B3C6DU; PIN2(hx[r_]); } } } while (0)
#define HC(q_) do { if (DROP) { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { const int r_ = 4 * (q_) + i_; const uint32_t x_ = hx[r_] ^ (hx[r_] >> 12); dsc[r_] = (x_ << hsh) >= dthr ? a.inv_keep : 0.f; PIN2(dsc[r_]); } } } while (0)
    // P units: P = exp2(S - lse) (keys past kv_len are NOT masked here: their dK / dV rows are written as zeros at the end and their
    // K^T fragments are zero, so nothing they produce is used), Pd = P * dropout scale, three-way split of Pd -> pw
#define LQ(g_) do { const f32x4 v_ = *reinterpret_cast<const f32x4*>(stats + st_cur + 8 * (g_) + 4 * h); lq[4 * (g_)] = v_.x; lq[4 * (g_) + 1] = v_.y; lq[4 * (g_) + 2] = v_.z; lq[4 * (g_) + 3] = v_.w; } while (0)
#define DL(g_) do { const f32x4 v_ = *reinterpret_cast<const f32x4*>(stats + st_cur + 32 + 8 * (g_) + 4 * h); dl[4 * (g_)] = v_.x; dl[4 * (g_) + 1] = v_.y; dl[4 * (g_) + 2] = v_.z; dl[4 * (g_) + 3] = v_.w; } while (0)
#define PA(q_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                 \
      const int m_ = 2 * (q_) + i_;                                                                                    \
      const float e0_ = __builtin_amdgcn_exp2f(s[2 * m_] - lq[2 * m_]), e1_ = __builtin_amdgcn_exp2f(s[2 * m_ + 1] - lq[2 * m_ + 1]); \
      pe[m_] = f32x2{e0_, e1_};                                                                                        \
      PIN2(pe[m_]);                                                                                                    \
    }                                                                                                                  \
  } while (0)
#define PB(q_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                 \
      const int m_ = 2 * (q_) + i_;                                                                                    \
      pd[m_] = DROP ? f32x2{pe[m_].x * dsc[2 * m_], pe[m_].y * dsc[2 * m_ + 1]} : pe[m_];                              \
      const uint32_t w_ = cvt2(pd[m_]);                                                                                \
      pwv[0][m_ >> 2][m_ & 3] = w_;                                                                                    \
      PIN2(pwv[0][m_ >> 2][m_ & 3]);                                                                                   \
    }                                                                                                                  \
  } while (0)
#define PC(q_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                 \
      const int m_ = 2 * (q_) + i_;                                                                                    \
      xx[m_] = pd[m_] - unpack2(pwv[0][m_ >> 2][m_ & 3]);                                                              \
      PIN2(xx[m_]);                                                                                                    \
    }                                                                                                                  \
  } while (0)
#define PD(q_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                 \
      const int m_ = 2 * (q_) + i_;                                                                                    \
      const uint32_t w_ = cvt2(xx[m_]);                                                                                \
      pwv[1][m_ >> 2][m_ & 3] = w_;                                                                                    \
      ff[m_] = unpack2(w_);                                                                                            \
      PIN2(ff[m_]);                                                                                                    \
    }                                                                                                                  \
  } while (0)
#define PE(q_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                 \
      const int m_ = 2 * (q_) + i_;                                                                                    \
      pwv[2][m_ >> 2][m_ & 3] = cvt2(xx[m_] - ff[m_]);                                                                 \
      PIN2(pwv[2][m_ >> 2][m_ & 3]);                                                                                   \
    }                                                                                                                  \
  } while (0)
    // Q phase: dS = Pd dP - P delta (= P (dP dropout - delta)), three-way split -> gw; dS^T rows of the wave's keys -> T^T
#define QA(q_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                 \
      const int m_ = 2 * (q_) + i_;                                                                                    \
      const f32x2 t_ = pe[m_] * f32x2{dl[2 * m_], dl[2 * m_ + 1]};                                                     \
      xx[m_] = pd[m_] * f32x2{dp[2 * m_], dp[2 * m_ + 1]} - t_;                                                        \
      const uint32_t w_ = cvt2(xx[m_]);                                                                                \
      gwv[0][m_ >> 2][m_ & 3] = w_;                                                                                    \
      PIN2(gwv[0][m_ >> 2][m_ & 3]);                                                                                   \
    }                                                                                                                  \
  } while (0)
#define QB(q_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                 \
      const int m_ = 2 * (q_) + i_;                                                                                    \
      xx[m_] = xx[m_] - unpack2(gwv[0][m_ >> 2][m_ & 3]);                                                              \
      PIN2(xx[m_]);                                                                                                    \
    }                                                                                                                  \
  } while (0)
#define QC(q_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                 \
      const int m_ = 2 * (q_) + i_;                                                                                    \
      const uint32_t w_ = cvt2(xx[m_]);                                                                                \
      gwv[1][m_ >> 2][m_ & 3] = w_;                                                                                    \
      ff[m_] = unpack2(w_);                                                                                            \
      PIN2(ff[m_]);                                                                                                    \
    }                                                                                                                  \
  } while (0)
#define QD(q_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                 \
      const int m_ = 2 * (q_) + i_;                                                                                    \
      gwv[2][m_ >> 2][m_ & 3] = cvt2(xx[m_] - ff[m_]);                                                                 \
      PIN2(gwv[2][m_ >> 2][m_ & 3]);                                                                                   \
    }                                                                                                                  \
  } while (0)
#define TW(g_)                                                                                                         \
  do {                                                                                                                 \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_)                                                                   \
      *reinterpret_cast<u32x2*>(lds + tt_w + p_ * TT_PLANE + (tw0 ^ (8 * (g_)))) =                                      \
          u32x2{gwv[p_][(g_) >> 1][2 * ((g_) & 1)], gwv[p_][(g_) >> 1][2 * ((g_) & 1) + 1]};                            \
  } while (0)
    // V phase: tile t + 2 -> LDS (the buffer tile t - 1 was read from), loads of tile t + 3
#define STQ(i_) B4_STOREP(i_, qd_fre)
#define STS() do { B4_STORES(st_fre); qt_next = min(t + 3, nq - 1); goff = (unsigned)(qt_next * 32 * D + (tid >> 3) * D + (tid & 7) * 8); } while (0)
#define LDG(i_) B4_LOADP(i_)
#define LDS_() B4_LOADS(qt_next, t + 3 >= nq)
    // K phase: this wave's partial dQ^T(t - 1) [d 32 x q 32] -> X[khalf][q][d]
#define XW(g_) do { *reinterpret_cast<f32x4*>(xbuf + x_b + (xw0 ^ (8 * (g_)))) = f32x4{dq[4 * (g_)], dq[4 * (g_) + 1], dq[4 * (g_) + 2], dq[4 * (g_) + 3]}; } while (0)
#ifndef BWD4_ABL
#define BWD4_ABL 0
#endif
#if BWD4_ABL & 1            /* ablation (timing only, wrong results): no softmax / dropout / split arithmetic */
#undef HA
#undef HB
#undef HC
#undef LQ
#undef DL
#undef PA
#undef PB
#undef PC
#undef PD
#undef PE
#undef QA
#undef QB
#undef QC
#undef QD
#define HA(q_) ((void)0)
#define HB(q_) ((void)0)
#define HC(q_) ((void)0)
#define LQ(g_) ((void)0)
#define DL(g_) ((void)0)
#define PA(q_) ((void)0)
#define PB(q_) ((void)0)
#define PC(q_) ((void)0)
#define PD(q_) ((void)0)
#define PE(q_) ((void)0)
#define QA(q_) ((void)0)
#define QB(q_) ((void)0)
#define QC(q_) ((void)0)
#define QD(q_) ((void)0)
#endif
#if BWD4_ABL & 2            /* no staging (the prologue's tiles are re-read) */
#undef STQ
#undef STS
#undef LDG
#undef LDS_
#define STQ(i_) ((void)0)
#define STS() ((void)0)
#define LDG(i_) ((void)0)
#define LDS_() ((void)0)
#endif
#if BWD4_ABL & 4            /* no barrier */
#define B4_SYNC() ((void)0)
#else
#define B4_SYNC() __syncthreads()
#endif
#if BWD4_ABL & 8            /* no dS^T / dQ exchange through LDS, no dQ output */
#undef TW
#undef XW
#undef XOL
#undef XOS
#undef XOP
#undef XOW
#undef XSIG
#define TW(g_) ((void)0)
#define XW(g_) ((void)0)
#define XOL(i_) ((void)0)
#define XOS(i_) ((void)0)
#define XOP() ((void)0)
#define XOW() ((void)0)
#define XSIG() ((void)0)
#endif
#if BWD4_ABL & 16           /* no fragment reads: MFMA skeleton */
#undef FRQ
#undef FRD
#undef FRQN
#undef FRA
#undef FRT
#define FRQ(p_, j_) kf[j_][p_]
#define FRD(p_, j_) vf[j_][p_]
#define FRQN(p_, j_) kf[j_][p_]
#define FRA(pl_, jj_, mt_) kf[jj_][(pl_) % 3]
#define FRT(p_, ks_) vf[ks_][p_]
#endif
#if BWD4_ABL & 32           /* dQ partial sums are not stored */
#undef XOS
#define XOS(i_) ((void)0)
#endif
#if BWD4_ABL & 64           /* no dropout hashes */
#undef HA
#undef HB
#undef HC
#define HA(q_) ((void)0)
#define HB(q_) ((void)0)
#define HC(q_) ((void)0)
#endif
#if BWD4_ABL & 128          /* no P units (planes of Pd stay zero) */
#undef PA
#undef PB
#undef PC
#undef PD
#undef PE
#define PA(q_) ((void)0)
#define PB(q_) ((void)0)
#define PC(q_) ((void)0)
#define PD(q_) ((void)0)
#define PE(q_) ((void)0)
#endif
#if BWD4_ABL & 256          /* no Q units */
#undef QA
#undef QB
#undef QC
#undef QD
#define QA(q_) ((void)0)
#define QB(q_) ((void)0)
#define QC(q_) ((void)0)
#define QD(q_) ((void)0)
#endif
#if BWD4_ABL & 512          /* no scheduling fences: hipcc places the units */
#undef SB
#define SB() ((void)0)
#endif
#if BWD4_ABL & 1024         /* no MFMAs (the other work alone) */
#undef MFMA_SP
#undef MFMA_SP0
#undef MFMA_Q
#undef MFMA_Q0
#undef MFMA_VK
#define MFMA_SP(acc_, a_, b_) asm volatile("" : "+v"(acc_) : "v"(a_), "a"(b_))
#define MFMA_SP0(acc_, a_, b_) asm volatile("" : "=&v"(acc_) : "v"(a_), "a"(b_))
#define MFMA_Q(acc_, a_, b_) asm volatile("" : "+v"(acc_) : "a"(a_), "v"(b_))
#define MFMA_Q0(acc_, a_, b_) asm volatile("" : "=&v"(acc_) : "a"(a_), "v"(b_))
#define MFMA_VK(acc_, a_, b_) asm volatile("" : "+a"(acc_) : "v"(a_), "v"(b_))
#endif
#include "attn_bwd4_phase.inc"

    // dropout decisions of tile 0 (the loop computes tile t + 1's behind the dK products of tile t)
    uint32_t hbn = hb;
    HA(0); HA(1); HA(2); HA(3); HB(0); HB(1); HB(2); HB(3); HC(0); HC(1); HC(2); HC(3);
#pragma unroll
    for (int p = 0; p < 3; ++p) fr[0][p] = FRQ(p, 0);
    for (int t = 0; t <= nq; ++t) {
      hbn += 32u * 0x85EBCA77U;
      SB();
      BWD4_ITER();
      { const int t_ = qd_cur; qd_cur = qd_nxt; qd_nxt = qd_fre; qd_fre = t_; }
      { const int t_ = st_cur; st_cur = st_nxt; st_nxt = st_fre; st_fre = t_; }
      { const int t_ = tt_w; tt_w = tt_r; tt_r = t_; }
      { const int t_ = x_a; x_a = x_b; x_b = t_; }
      B4_SYNC();
    }
    {                                             // the last tile's dQ (written in iteration nq)
      const int t = nq + 1;
      XOL(0); XOL(1);
      XOP(); XOW();
      XOS(0); XOS(1);
      XSIG();
    }
  }
  uint32_t gmax = 0u;
  if (key < a.Lk) {
    float* pk = a.dk + ((size_t)b * a.Lk + key) * a.ldk + head * D;
    float* pv = a.dv + ((size_t)b * a.Lk + key) * a.ldv + head * D;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // Q was pre-scaled by log2(e)/8: dK = dS^T.Q / 8 = (dS^T.Qs) * ln 2
        const float zk = kvalid ? LN2 : 0.f, zv = kvalid ? 1.f : 0.f;          // (masked keys: finite garbage x 0)
        const float4 gk = make_float4(dk[mt][4 * g] * zk, dk[mt][4 * g + 1] * zk, dk[mt][4 * g + 2] * zk, dk[mt][4 * g + 3] * zk);
        const float4 gv = make_float4(dv[mt][4 * g] * zv, dv[mt][4 * g + 1] * zv, dv[mt][4 * g + 2] * zv, dv[mt][4 * g + 3] * zv);
        *reinterpret_cast<float4*>(pk + 32 * mt + 8 * g + 4 * h) = gk;
        *reinterpret_cast<float4*>(pv + 32 * mt + 8 * g + 4 * h) = gv;
        gmax = max(gmax, max(mag_bits4(gk), mag_bits4(gv)));
      }
  }
  if (a.mag) {                                // dk / dv's share of the row magnitudes of [dq | dk | dv] (common.h): lanes c, c + 32 hold a key's row
    gmax = max(gmax, (uint32_t)__shfl_xor((int)gmax, 32, 64));
    if (h == 0 && key < a.Lk) atomicMax(a.mag + (size_t)b * a.Lk + key, gmax);
  }
}

int attention_bwd4_emu_launch(const EmuAttn& a, bool chain, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    const void* ks[4] = {reinterpret_cast<const void*>(emu_attn_bwd4_kernel<true, true>), reinterpret_cast<const void*>(emu_attn_bwd4_kernel<false, true>),
                         reinterpret_cast<const void*>(emu_attn_bwd4_kernel<true, false>), reinterpret_cast<const void*>(emu_attn_bwd4_kernel<false, false>)};
    for (const void* k : ks)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4_LDS_BYTES) != hipSuccess) {
        set_error("attention_bwd_emu: cannot raise the dynamic LDS limit to %u bytes", B4_LDS_BYTES);
        return HOISDF_ERR_LAUNCH;
      }
    attr_set = true;
  }
  const dim3 grid(cdiv(a.Lk, 128) * 8 * cdiv(a.B * a.H, 8));
  const bool drop = a.drop_p > 0.f;
  if (chain) {
    if (drop) hipLaunchKernelGGL((emu_attn_bwd4_kernel<true, true>), grid, dim3(256), B4_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((emu_attn_bwd4_kernel<false, true>), grid, dim3(256), B4_LDS_BYTES, st, a);
  } else {
    if (drop) hipLaunchKernelGGL((emu_attn_bwd4_kernel<true, false>), grid, dim3(256), B4_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((emu_attn_bwd4_kernel<false, false>), grid, dim3(256), B4_LDS_BYTES, st, a);
  }
  return check_launch("attention_bwd_emu (bwd4)");
}

}  // namespace hoisdf
