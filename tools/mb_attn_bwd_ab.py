#!/usr/bin/env python3
"""A/B of two builds of the emulated attention backward (label=ENV ... like tools/mb_kc2.py): time per call at the step's shapes and
hashes of dq / dk / dv (a re-scheduling must leave them bit-identical)."""
import sys, os, subprocess, json, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SHAPES = [(32, 2048, 2048, 0.1), (32, 2048, 2048, 0.0), (32, 1536, 2048, 0.1), (32, 512, 2048, 0.1)]


def child():
    import torch
    from hoisdf_amd import ops
    dev = "cuda"
    def timeit(fn, iters=20, warm=3):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3
    out = []
    g = torch.Generator(device=dev); g.manual_seed(3)
    for B, Lq, Lk, p in SHAPES:
        E, H = 256, 4
        q = torch.randn(B, Lq, E, device=dev, generator=g); kv = torch.randn(B, Lk, 2 * E, device=dev, generator=g)
        do = torch.randn(B, Lq, E, device=dev, generator=g)
        k, v = kv[:, :, :E], kv[:, :, E:]
        dq = torch.empty_like(q); dkv = torch.empty_like(kv)
        oe, lsee = ops._attn_fwd_emu(q, k, v, H, Lk, p, 1234, keep=False)
        fn = lambda: ops._attn_bwd_emu(q, k, v, oe, lsee, do, dq, dkv[:, :, :E], dkv[:, :, E:], H, Lk, p, 1234)
        t = timeit(fn)
        h = hashlib.sha1(dq.cpu().numpy().tobytes() + dkv.cpu().numpy().tobytes()).hexdigest()[:12]
        out.append(dict(us=t, h=h))
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    cfgs = sys.argv[1:]
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
                print(label, p.stdout[-2000:], p.stderr[-2000:]); sys.exit(1)
            res.setdefault(label, []).append(json.loads(line[0][7:]))
    for i, sh in enumerate(SHAPES):
        print(sh, "  ".join(f"{l}: {min(r[i]['us'] for r in res[l]):8.1f} us {res[l][0][i]['h']}" for l in res))
