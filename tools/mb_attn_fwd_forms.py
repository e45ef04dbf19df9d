#!/usr/bin/env python3
"""attention forward at the step's shape (B = 32, S = 2048, E = 256, H = 4, dropout 0.1): bf16x3 (six products) against the f16x2 form
(three products; conversion passes included in both)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
from hoisdf_amd import ops as O
from hoisdf_amd._lib import call, lib

B, S, E, H = 32, 2048, 256, 4
qkv = torch.randn(B, S, 3 * E, device="cuda")
q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
_hm = O._head_measure(qkv, 3 * E, B * S, 3 * H, S)              # head magnitudes of [q | k | v]: one scale per (sample, head) and operand
heads = (_hm, _hm[H * B:], _hm[2 * H * B:])


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


fl = 4.0 * B * H * S * S * 64
for p in (0.0, 0.1):
    a = t(lambda: O._attn_fwd_emu(q, k, v, H, S, p, 7))
    b = t(lambda: O._attn_fwd_emu(q, k, v, H, S, p, 7, heads=heads))
    print(f"dropout {p}: bf16x3 {a:8.1f} us = {fl / a / 1e6:6.1f} TF   f16x2 {b:8.1f} us = {fl / b / 1e6:6.1f} TF")

# backward (conversion of dO, delta pass, kernel, dQ reduce; q / k / v converted inside: no kept planes in this harness)
go = torch.randn(B, S, E, device="cuda") * 1e-3
dqkv = torch.empty(B, S, 3 * E, device="cuda"); dq, dkv = dqkv[..., :E], dqkv[..., E:]           # (the layouts of q / k / v)
flb = 10.0 * B * H * S * S * 64
for p in (0.0, 0.1):
    o1, l1 = O._attn_fwd_emu(q, k, v, H, S, p, 7)
    o2, l2 = O._attn_fwd_emu(q, k, v, H, S, p, 7, heads=heads)
    a = t(lambda: O._attn_bwd_emu(q, k, v, o1, l1, go, dq, dkv[..., :E], dkv[..., E:], H, S, p, 7))
    b = t(lambda: O._attn_bwd_emu(q, k, v, o2, l2, go, dq, dkv[..., :E], dkv[..., E:], H, S, p, 7, heads=heads))
    print(f"backward, dropout {p}: bf16x3 {a:8.1f} us = {flb / a / 1e6:6.1f} TF   f16x2 {b:8.1f} us = {flb / b / 1e6:6.1f} TF")
