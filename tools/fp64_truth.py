#!/usr/bin/env python3
"""fp64 ground truth for the ill-conditioned element-wise pyramid gradient of the N=2048 training fixture: the CPU oracle run
in float64 and in float32 next to the REFERENCE's own fp32 value (tests/golden/g8_train_dexycb_n2048.npz).  Writes
tests/golden/g8_train_dexycb_n2048_fp64.npz (grad.pyr.stride32 in fp64): what the GPU paths are compared with when the question
is 'which fp32 implementation is closer to the exact value'."""
import sys, os, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from hoisdf_amd import testing as T
from hoisdf_amd.nets import mano as MANO
from oracle import hoisdf_oracle as O
torch.set_num_threads(8)
g = dict(np.load("tests/golden/g8_train_dexycb_n2048.npz"))
nh, no, b = 1536, 512, 2
def run(dt):
    Pm = {k: v.to(dt).requires_grad_(True) for k, v in T.det_params(T.hot_path_param_shapes(992, ik=False)).items()}
    cfg = O.OracleCfg(num_samp_hand=nh, num_samp_obj=no, bins_n=16, use_inverse_kinematics=False, dataset="dexycb", dropout=0.0, sdf_dropout=0.0)
    pyr = {k: v.to(dt).requires_grad_(True) for k, v in T.synthetic_pyramid(b, big=False, seed=3).items()}
    inputs, targets, meta = T.synthetic_batch(b, nh, no, seed=31)
    cast = lambda d: {k: (v.to(dt) if v.is_floating_point() else v) for k, v in d.items()}
    inputs, targets, meta = cast(inputs), cast(targets), cast(meta)
    layer = MANO.ManoLayer(MANO.synthetic_assets(0)).to(dt)
    random.seed(0); torch.manual_seed(1234)
    # the jitter is drawn in fp32 by the reference: draw it in fp32 and cast, so all runs see the same points
    _e = torch.empty_like
    def empty_like32(t, *a, **k):
        return _e(t.float(), *a, **k)
    keep = torch.Tensor.uniform_
    def uni(self, lo, hi):
        r = keep(torch.empty(self.shape, dtype=torch.float32), lo, hi)
        return self.copy_(r.to(self.dtype))
    torch.Tensor.uniform_ = uni
    try:
        out = O.hot_path_forward(Pm, cfg, pyr, inputs, targets, meta, "train", 0, 0.5, mano_layer=layer, hands_mean=layer.th_hands_mean)
    finally:
        torch.Tensor.uniform_ = keep
    total = sum(v.mean() for k, v in out.items() if "_out" not in k)
    total.backward()
    return pyr["stride32"].grad[:, ::16].double(), float(total)
ref = torch.from_numpy(g["grad.pyr.stride32"]).double()
g64, t64 = run(torch.float64)
g32, t32 = run(torch.float32)
mx = float(ref.abs().max())
print("total: fp64 %.9f  fp32 %.9f  golden %.9f" % (t64, t32, float(g["total"])))
for name, a in (("oracle fp32", g32), ("reference golden (fp32)", ref)):
    d = (a - g64).abs()
    idx = np.unravel_index(int(d.argmax()), d.shape)
    print(f"{name:24s} vs oracle fp64: max {float(d.max()) / mx:.3e} of max at {tuple(int(i) for i in idx)}  rms {float((d**2).mean().sqrt()) / mx:.3e}")
print("element (1,6,5,5): fp64 %.7f  oracle fp32 %.7f  golden %.7f" % (float(g64[1, 6, 5, 5]), float(g32[1, 6, 5, 5]), float(ref[1, 6, 5, 5])))
print("element (0,17,6,5): fp64 %.7f  oracle fp32 %.7f  golden %.7f" % (float(g64[0, 17, 6, 5]), float(g32[0, 17, 6, 5]), float(ref[0, 17, 6, 5])))
np.savez_compressed("tests/golden/g8_train_dexycb_n2048_fp64.npz", **{"grad.pyr.stride32": g64.numpy(), "total": np.float64(t64)})
