"""(f2) throughput of the device-side SDF sample selection: 64 frames x ~20k rows resident, batches of 32 frames,
1536 + 512 draws (+ the same again for the training pre-points)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from hoisdf_amd.sdf_data import SdfStore
rng = np.random.default_rng(0)
frames, index = [], []
for f in range(64):
    nh, no = 12000 + int(rng.integers(0, 2000)), 8000 + int(rng.integers(0, 2000))
    a = rng.standard_normal((nh + no, 6)).astype(np.float32); a[:, 3:5] = rng.uniform(-0.2, 0.2, (nh + no, 2))
    frames.append(a); index.append([nh, no])
st = SdfStore(frames, np.array(index))
ids = rng.integers(0, 64, 32)
for train in (False, True):
    for validate in (True, False):
        for _ in range(3): st.sample(ids, 1536, 512, 0.05, train, seed=1, validate=validate)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): st.sample(ids, 1536, 512, 0.05, train, seed=i, validate=validate)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"train={train} validate={validate}: {dt*1e3:.2f} ms per batch of 32 = {32/dt:.0f} samples/s")
