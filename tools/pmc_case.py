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
    # round 6, HBM-bound passes: (f4) BatchNorm + residual + ReLU of a 32 x 256 x 64 x 64 channels_last map (ResNet-50 layer1's block tail:
    # 131072 rows x 256 channels = 134 MB per map); shape = (rows, channels, 1 = with residual)
    "bn_stats_131072x256": ("hoisdf_bn_stats", "bn_stats_kernel", (131072, 256, 1)),
    "bn_apply_fwd_131072x256": ("hoisdf_bn_apply_fwd", "bn_apply_fwd_kernel<true, true>", (131072, 256, 1)),
    "bn_bwd_reduce_131072x256": ("hoisdf_bn_bwd", "bn_bwd_reduce_kernel<true>", (131072, 256, 1)),
    "bn_bwd_dx_131072x256": ("hoisdf_bn_bwd", "bn_bwd_dx_kernel<true, true>", (131072, 256, 1)),
    # add + LayerNorm of one encoder layer's 65536 x 256 rows (dropout on the residual branch, dx_add and dr as in the layer's backward)
    "add_ln_fwd_65536x256": ("hoisdf_add_layernorm_fwd", "add_ln_fwd256_kernel", (65536, 256, 0)),
    "add_ln_bwd_65536x256": ("hoisdf_add_layernorm_bwd", "add_ln_bwd256_kernel", (65536, 256, 0)),
    # K1 at the lattice survivors of one configs[3] sdf_infer call: 320 000 points x 992 channels (shape = rows, channels, 0)
    "gather_fwd_320000x992": ("hoisdf_project_gather_fwd", "gather_fwd4_kernel", (320000, 992, 0)),
}
if __name__ == "__main__":
    if sys.argv[1] == "--list":
        print(" ".join(CASES)); sys.exit(0)
    import torch
    from hoisdf_amd import ops as O
    case = sys.argv[1]
    entry, _, shape = CASES[case]
    dev = "cuda"
    if case.startswith("bn_"):
        M, Cc, _ = shape
        x = torch.randn(32, Cc, 64, M // (32 * 64), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        res = torch.randn_like(x).requires_grad_(True)
        w, b = torch.rand(Cc, device=dev).requires_grad_(True), torch.randn(Cc, device=dev).requires_grad_(True)
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        gy = torch.randn_like(x)
        for _ in range(3):
            x.grad = res.grad = w.grad = b.grad = None
            y = O.bn_act(x, w, b, rm, rv, True, 0.1, 1e-5, True, res)
            y.backward(gy)
        torch.cuda.synchronize()
        sys.exit(0)
    if case.startswith("add_ln_"):
        M, D, _ = shape
        x = torch.randn(M, D, device=dev, requires_grad=True); r = torch.randn(M, D, device=dev, requires_grad=True)
        g = torch.rand(D, device=dev, requires_grad=True); b = torch.randn(D, device=dev, requires_grad=True)
        gy = torch.randn(M, D, device=dev)
        for _ in range(3):
            x.grad = r.grad = g.grad = b.grad = None
            y = O.add_layernorm(x, r, g, b, 1e-5, 0.1)
            y.backward(gy)
        torch.cuda.synchronize()
        sys.exit(0)
    if case.startswith("gather_"):
        from hoisdf_amd import testing as T
        n, Cc, _ = shape
        B = 16
        g = torch.Generator(device=dev).manual_seed(0)
        levels = [torch.randn(B, h, h, c, device=dev, generator=g) for h, c in ((128, 32), (64, 64), (32, 128), (16, 256), (8, 512))]
        pyr = O.PyramidNHWC(levels)
        per = n // B
        # a 64^3-like lattice slab in front of the camera: consecutive points are depth neighbours, as sdf_infer's survivors are
        zz = torch.linspace(-1, 1, 64, device=dev)
        idx = torch.arange(per, device=dev)
        pts = torch.stack([((idx // 4096) % 64).float() / 31.5 - 1, ((idx // 64) % 64).float() / 31.5 - 1, zz[idx % 64]], 1)
        pts = pts.unsqueeze(0).repeat(B, 1, 1).contiguous()
        center = torch.tensor([0.0, 0.0, 0.7], device=dev).repeat(B, 1)
        K = torch.tensor([[250.0, 0, 128], [0, 250.0, 128], [0, 0, 1]], device=dev).repeat(B, 1, 1)
        with torch.no_grad():
            for _ in range(3):
                O.project_gather(pyr, pts, center, K, 3.1)
        torch.cuda.synchronize()
        sys.exit(0)
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
