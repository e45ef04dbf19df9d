import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import test_gpu_model as M
from hoisdf_amd import ops, testing as T
from oracle import hoisdf_oracle as R
nh, no, bins, b = 384, 128, 64, 2
model, c = M.build("dexycb", nh, no, bins, train=True)
P = T.det_params(T.hot_path_param_shapes(992))
pyr_cpu = T.synthetic_pyramid(b, seed=8)
pyr, _ = M.nhwc_pyramid(pyr_cpu)
_, _, meta = T.synthetic_batch(b, nh, no, seed=81)
oc = M.oracle_cfg(c)
key = lambda r: tuple(round(float(x), 6) for x in r)
def oracle_run(training, seed):
    torch.manual_seed(seed)
    return R.sdf_infer(P, oc, pyr_cpu, meta["mano_root"], meta["cam_intr"], meta["bbox_hand"], 3.1, nh, "hand", return_debug=True, training=training)
pts_c, sdf_c, _, dbg_c = oracle_run(False, 0)
lattice = R.dense_lattice(bins)
clean_of = [{key(p): float(v) for p, v in zip(lattice[d["keep"]].tolist(), d["sdf"].abs().tolist())} for d in dbg_c]
noisy = [oracle_run(True, seed) for seed in (1, 2, 3, 4, 5, 6)]
m = T.to_device(meta, M.DEV)
for i in range(b):
    o_sets = [{key(r) for r in n[0][i].tolist()} for n in noisy]
    om = [float(n[1][i].abs().mean()) for n in noisy]; ok = [float(n[1][i].abs().max()) for n in noisy]
    oc_ = [sum(clean_of[i][k] for k in st) / nh for st in o_sets]
    print("oracle sample", i, "mean", [round(x,4) for x in om], "kth", [round(x,4) for x in ok], "clean", [round(x,4) for x in oc_])
for seed in range(12):
    ops.manual_seed(1000 + seed)
    r_ = model.sdf_infer(pyr, m["mano_root"], m["cam_intr"], m["bbox_hand"], 3.1, nh, "hand")
    out = []
    for i in range(b):
        sdf = r_[1][i, :, 0].abs().cpu()
        st = {key(r) for r in r_[0][i].cpu().tolist()}
        out.append((round(float(sdf.mean()),4), round(float(sdf.max()),4), round(sum(clean_of[i][k] for k in st) / nh,4)))
    print("device seed", seed, out)
