"""(run on the GPU box) how long does the host block in dist.all_reduce(async_op=True) for a 64 MB bucket issued while the stream is busy?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
flat = torch.zeros(16 * 1024 * 1024, device=dev)
a = torch.randn(8192, 8192, device=dev)
for _ in range(3):
    dist.all_reduce(flat); torch.cuda.synchronize()
for busy in (0, 1):
    ts = []
    for _ in range(5):
        if busy:
            for _ in range(20): b = a @ a          # ~ tens of ms of queued work on the current stream
        t0 = time.perf_counter()
        h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        t1 = time.perf_counter()
        h.wait()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3)))
    print("stream busy" if busy else "stream idle", "host ms in all_reduce(async) / in wait():", ts)
# GPU-side: does the collective delay following work on the main stream?
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for with_ar in (0, 1, 0, 1):
    torch.cuda.synchronize(); s.record()
    for i in range(40):
        b = a @ a
        if with_ar and i % 10 == 5:
            h = dist.all_reduce(flat, async_op=True)
    if with_ar: h.wait()
    e.record(); torch.cuda.synchronize()
    print("40 GEMMs", "with 4 async all-reduces" if with_ar else "alone", round(s.elapsed_time(e), 2), "ms")
