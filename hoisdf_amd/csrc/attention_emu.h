// Shared declarations of the fp32-emulating attention kernels (attention_emu.hip: conversion pre-pass, forward, the round-3 backward;
// attention_emu_bwd4.hip: the round-5 backward).
#pragma once
#include "common.h"

namespace hoisdf {
namespace emu_attn {

struct EmuAttn {
  // planes (any may be null when a kernel does not use it): [p] = piece 0, 1, 2
  const __bf16 *q[3];              // Q rows (pre-scaled)
  const __bf16 *k[3];              // K rows
  const __bf16 *v[3], *vt[3];      // V rows (backward), V^T (forward)
  const __bf16 *d[3];              // dO rows
  const float* lse_in; const float* delta;
  float* out; float* lse; float* dq_part; float* dk; float* dv;
  // the chained dQ accumulation of emu_attn_bwd4_kernel<*, true>: dq_part = ONE running sum [bh][Lq][64] that the key blocks of a
  // (b, head) add to in key-block order, dq_flags [bh][key block][4 waves] = query tiles a wave has added, dq / ldq = the final output
  float* dq; int* dq_flags; int ldq;
  int chain_group;                 // key blocks per chain (consecutive blocks kb / chain_group share one running sum; >= 1)
  int ldo, ldk, ldv;
  int B, H, Lq, Lk, Lqp, Lkp, kv_len;
  float drop_p, inv_keep;
  uint32_t thresh;
  uint64_t seed;
  // f16x2 form: head magnitudes (common.h) the planes of Q, K, V and dO were made with - ONE scale per (sample, head) and operand:
  // word of (b, head) at x_hm[head * B + b] (each pointer is positioned at its operand's first head inside its matrix' array)
  const uint32_t *q_hm, *k_hm, *v_hm, *d_hm;
  float* dq_scale;                 // f16x2 backward: one float per (b, head) the kernel leaves for the dQ reduce pass (0.125 / (sK sS))
  uint32_t* mag;                   // row magnitudes (common.h) of what the kernel writes: out (forward), [dq | dk | dv] rows (backward); null = none
};

__device__ __forceinline__ bool emu_block(int nx, int nbh, int& tile, int& bh) {
  const int L = blockIdx.x, slot = L >> 3;        // every tile of a (b, head) on one XCD, as in attention.hip
  bh = (slot / nx) * 8 + (L & 7);
  tile = slot % nx;
  return bh < nbh;
}

}  // namespace emu_attn

// attention_emu_bwd4.hip: launches emu_attn_bwd4_kernel over planes that are already converted (the argument block of
// hoisdf_attention_bwd_emu); returns a HOISDF status
// (chain: dQ through the ordered in-L2 running sum - a.dq_part one [bh][Lq][64] buffer, a.dq_flags zeroed - instead of partials)
int attention_bwd4_emu_launch(const emu_attn::EmuAttn& a, bool chain, hipStream_t st);
// attention_emu_bwd4h.hip: the f16x2 form (two f16 planes per operand in a.q / a.k / a.v / a.d, per-(sample, head) scales from a.q_hm / k_hm / v_hm / d_hm)
int attention_bwd4h_emu_launch(const emu_attn::EmuAttn& a, hipStream_t st);
}  // namespace hoisdf
