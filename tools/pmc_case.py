#!/usr/bin/env python3
"""PMC driver: ONE launch family at ONE shape per process (run under `rocprofv3 --pmc ...`), so that a counter row can be keyed
by (C entry, shape) without guessing.  usage: pmc_case.py <case>;  `pmc_case.py --list` prints the case names."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CASES = {
    # name: (C entry, kernel substring, shape)
    "attn_bwd_f32_32x2048x2048": ("hoisdf_attention_bwd", "attn_bwd_fused_kernel", (32, 2048, 2048)),
    "attn_fwd_emu_32x2048x2048": ("hoisdf_attention_fwd_emu", "emu_attn_fwd2_kernel", (32, 2048, 2048)),
    "attn_bwd_emu_32x2048x2048": ("hoisdf_attention_bwd_emu", "emu_attn_bwd4_kernel", (32, 2048, 2048)),
    # the f16x2 form of the encoder attention (default in the layers)
    "attn_fwd_h2_32x2048x2048": ("hoisdf_attention_fwd_emu_mag", "emu_attn_fwd2_kernel<true, 2, true>", (32, 2048, 2048)),
    "attn_bwd_h2_32x2048x2048": ("hoisdf_attention_bwd_emu_mag", "emu_attn_bwd4h_kernel", (32, 2048, 2048)),
    "attn_fwd_bf16x2_4x8192x8192": ("hoisdf_attention_fwd_bf16x2", "emu_attn_fwd2_kernel<false, 2>", (4, 8192, 8192)),
    # the f16x2 form of the linear layers (default; what the Python path calls: the *_mag entries after one hoisdf_mag_measure per operand)
    "linear_fwd_emu_65536x1024x256": ("hoisdf_linear_fwd_emu_mag", "emu_h2_kernel<false, false, 2>", (65536, 1024, 256)),
    "linear_fwd_emu_65536x256x1024": ("hoisdf_linear_fwd_emu_mag", "emu_h2_kernel<false, false, 2>", (65536, 256, 1024)),
    "linear_fwd_emu_65536x256x256": ("hoisdf_linear_fwd_emu_mag", "emu_h2_kernel<false, false, 2>", (65536, 256, 256)),
    "linear_bwd_input_emu_65536x1024x256": ("hoisdf_linear_bwd_input_emu_mag", "emu_h2_kernel<true, false, 4>", (65536, 1024, 256)),
    "linear_bwd_weight_emu_65536x1024x256": ("hoisdf_linear_bwd_weight_emu_mag", "emu_dw2h_kernel<true, true>", (65536, 1024, 256)),
    # the bf16x3 grad-weight (no magnitudes needed: what runs where no words are at hand)
    "linear_bwd_weight_b3_65536x1024x256": ("hoisdf_linear_bwd_weight_emu", "emu_dw2_kernel<true, true>", (65536, 1024, 256)),
    "linear_fwd_f32_65536x1024x256": ("hoisdf_linear_fwd", "gemm_f32_kernel<true, true, false, false>", (65536, 1024, 256)),
}
if __name__ == "__main__":
    if sys.argv[1] == "--list":
        print(" ".join(CASES)); sys.exit(0)
    import torch
    from hoisdf_amd import ops as O
    case = sys.argv[1]
    entry, _, shape = CASES[case]
    dev = "cuda"
    if "attn" in case:
        B, Lq, Lk = shape
        E, H, p = 256, 4, 0.1
        q = torch.randn(B, Lq, E, device=dev); kv = torch.randn(B, Lk, 2 * E, device=dev); do = torch.randn(B, Lq, E, device=dev)
        k, v = kv[:, :, :E], kv[:, :, E:]
        dq = torch.empty_like(q); dkv = torch.empty_like(kv)
        if entry == "hoisdf_attention_fwd_bf16x2":              # configs[4]: evaluation, no dropout
            O.set_attention_f16_eval(True)
            with torch.no_grad():
                for _ in range(3):
                    O._attn_fwd_f16(q, k, v, H, Lk)
            torch.cuda.synchronize()
            sys.exit(0)
        if "_h2_" in case:
            import ctypes as C
            from hoisdf_amd._lib import call, lib
            hq_, hkv_ = O._head_measure(q, E, B * Lq, H, Lq), O._head_measure(kv, 2 * E, B * Lk, 2 * H, Lk)
            heads = (hq_, hkv_, hkv_[H * B:])
            o, lse = O._attn_fwd_emu(q, k, v, H, Lk, p, 1234, heads=heads)
            for _ in range(3):
                if entry == "hoisdf_attention_fwd_emu_mag":
                    O._attn_fwd_emu(q, k, v, H, Lk, p, 1234, heads=heads)
                else:
                    O._attn_bwd_emu(q, k, v, o, lse, do, dq, dkv[:, :, :E], dkv[:, :, E:], H, Lk, p, 1234, heads=heads)
            torch.cuda.synchronize()
            sys.exit(0)
        o, lse = O._attn_fwd(q, k, v, H, Lk, p, 1234)
        for _ in range(3):
            if entry == "hoisdf_attention_bwd":
                O._attn_bwd(q, k, v, o, lse, do, dq, dkv[:, :, :E], dkv[:, :, E:], H, Lk, p, 1234)
            elif entry == "hoisdf_attention_fwd_emu":
                O._attn_fwd_emu(q, k, v, H, Lk, p, 1234, keep=False)
            else:
                O._attn_bwd_emu(q, k, v, o, lse, do, dq, dkv[:, :, :E], dkv[:, :, E:], H, Lk, p, 1234)
    else:
        M, N, K = shape
        x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev)
        y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        bits = torch.empty(M, (N + 31) // 32, dtype=torch.int32, device=dev)
        O.set_gemm_emu("_emu" in case or "_b3_" in case)
        O._gemm_fwd(x, K, W, b, y, N, M, N, K, 1, 0.0, 0, bits)
        ymag = torch.zeros(M, dtype=torch.int32, device=dev) if "_emu" in case else None      # (as in the step: the epilogue leaves y's row magnitudes)
        xmag = O._mag_measure(x, K, M, K) if "_emu" in case else None
        for _ in range(3):
            if "linear_fwd" in case:
                O._gemm_fwd(x, K, W, b, y, N, M, N, K, 1, 0.1 if "_emu" in case else 0.0, 7, bits, x_mag=xmag, y_mag=ymag)
            elif "bwd_input" in case:
                O._gemm_bwd_input(dy, N, bits, 0.0, W, dx, K, M, N, K, 0)
            else:
                O._gemm_bwd_weight(dy, N, bits, 0.0, x, K, dW, db, M, N, K, form="b3" if "_b3_" in case else "h2")
    torch.cuda.synchronize()
