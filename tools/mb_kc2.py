#!/usr/bin/env python3
"""A/B of the two main-loop forms of the emulated forward / grad-input kernel (HOISDF_EMU_KC=1: first form, 2: rotated,
hand-interleaved): per-shape time, and bit-equality of the outputs (same product order per accumulator)."""
import sys, os, math, subprocess, json, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SHAPES = [(65536, 1024, 256), (65536, 256, 1024), (65536, 768, 256), (65536, 256, 256), (294912, 256, 256), (49152, 1024, 992),
          (49152, 512, 512), (49152, 512, 256), (49152, 256, 256), (16384, 1024, 992), (16384, 512, 512), (16384, 256, 512), (49152, 512, 292),
          (294912, 60, 256), (1000, 200, 60)]


def child():
    import torch
    from hoisdf_amd import ops as O
    dev = "cuda"
    def timeit(fn, iters=20, warm=3):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e-3
    out = []
    g = torch.Generator(device=dev); g.manual_seed(1)
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)
        b = torch.randn(N, device=dev, generator=g)
        dy = torch.randn(M, N, device=dev, generator=g)
        y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev)
        bits = torch.empty(M, (N + 31) // 32, dtype=torch.int32, device=dev)
        fl = 2.0 * M * N * K
        t1 = timeit(lambda: O._gemm_fwd(x, K, W, b, y, N, M, N, K, 1, 0.1, 1234, bits))
        h1 = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12] + hashlib.sha1(bits.cpu().numpy().tobytes()).hexdigest()[:6]
        t0 = timeit(lambda: O._gemm_fwd(x, K, W, None, y, N, M, N, K, 0, 0.0, 0, None))
        h0 = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
        t2 = timeit(lambda: O._gemm_bwd_input(dy, N, bits, 0.1, W, dx, K, M, N, K, 0))
        h2 = hashlib.sha1(dx.cpu().numpy().tobytes()).hexdigest()[:12]
        t3 = timeit(lambda: O._gemm_bwd_input(dy, N, None, 0.0, W, dx, K, M, N, K, 0))
        h3 = hashlib.sha1(dx.cpu().numpy().tobytes()).hexdigest()[:12]
        # grad-weight (+ bias gradient): plain and with the sign bitmap
        dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        if N % 4 == 0 and K % 4 == 0 and M >= 8192:
            t4 = timeit(lambda: O._gemm_bwd_weight(dy, N, None, 0.0, x, K, dW, db, M, N, K))
            h4 = hashlib.sha1(dW.cpu().numpy().tobytes()).hexdigest()[:12] + hashlib.sha1(db.cpu().numpy().tobytes()).hexdigest()[:6]
            t5 = timeit(lambda: O._gemm_bwd_weight(dy, N, bits, 0.1, x, K, dW, db, M, N, K))
            keep = ((bits[:, torch.arange(N, device=dev) // 32] >> (torch.arange(N, device=dev) % 32)) & 1).double()
            ge = dy.double() * keep / 0.9
            edw = ((dW.double() - ge.t() @ x.double()).abs().max() / (ge.t() @ x.double()).abs().max()).item()
            edb = ((db.double() - ge.sum(0)).abs().max() / ge.sum(0).abs().max()).item()
        else:
            t4 = t5 = float("inf"); h4 = ""; edw = edb = 0.0
        # masked grad-input vs fp64
        O._gemm_bwd_input(dy, N, bits, 0.1, W, dx, K, M, N, K, 0)
        if M <= 70000:
            keep2 = ((bits[:, torch.arange(N, device=dev) // 32] >> (torch.arange(N, device=dev) % 32)) & 1).double()
            rdx = (dy.double() * keep2 / 0.9) @ W.double()
            edx = ((dx.double() - rdx).abs().max() / rdx.abs().max()).item()
            del keep2, rdx
        else:
            edx = 0.0
        # vs fp64 on a row sample
        idx = torch.arange(0, M, max(1, M // 64), device=dev)[:64]
        ref = (x[idx].double() @ W.double().t())
        O._gemm_fwd(x, K, W, None, y, N, M, N, K, 0, 0.0, 0, None)
        err = ((y[idx].double() - ref).abs().max() / ref.abs().max()).item()
        out.append(dict(shape=[M, N, K], tf=[fl / t / 1e12 for t in (t0, t1, t3, t2, t4, t5)], us=[t * 1e6 for t in (t0, t1, t3, t2, t4, t5)],
                        h=[h0, h1, h3, h2, h4], err=err, edx=edx, edw=edw, edb=edb))
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    # configurations: label=ENV1=v,ENV2=v ... (default: the two forms of the built library)
    cfgs = sys.argv[1:] or ["form1=HOISDF_EMU_KC=1", "form2=HOISDF_EMU_KC=2"]
    res = {}
    for rep in range(2):
        for c in cfgs:
            label, _, envs = c.partition("=")
            env = dict(os.environ)
            for kv in envs.split(","):
                if kv:
                    k, _, v = kv.partition("=")
                    env[k] = v
            p = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(label, p.stdout[-3000:], p.stderr[-3000:]); sys.exit(1)
            res.setdefault(label, []).append(json.loads(line[0][7:]))
    labels = list(res)
    print("TF-eq (best of 2 runs): fwd / fwd+bias+relu+dropout / dx / dx+mask / dW+db / dW+db+mask; '=' outputs bit-equal to the first configuration (fwd, fwd+act, dx, dx+mask, dW+db); err vs fp64: plain fwd, masked dx, masked dW, masked db")
    for i, (M, N, K) in enumerate(SHAPES):
        print(f"{M:7d} {N:5d} {K:5d}")
        for l in labels:
            b = [max(r[i]["tf"][j] for r in res[l]) for j in range(6)]
            eq = "".join("=" if res[l][0][i]["h"][j] == res[labels[0]][0][i]["h"][j] else "x" for j in range(5))
            r0 = res[l][0][i]
            print(f"      {l:>10}: " + " ".join(f"{v:6.1f}" for v in b) + f"  {eq}  err {r0['err']:.1e} dx {r0['edx']:.1e} dW {r0['edw']:.1e} db {r0['edb']:.1e}")
