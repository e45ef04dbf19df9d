"""which outputs of sample 0 change (a) between two identical runs, (b) when the batch companions change"""
import sys, torch
sys.path.insert(0, ".")
from hoisdf_amd import ops, testing as T
from hoisdf_amd.config import Config
from hoisdf_amd.model import get_model
from hoisdf_amd.nets import mano as MANO
DEV = "cuda"; B, NH, NO = 32, 1536, 512
c = Config(); c.resnet_type = 50; c.apply_setting("dexycb"); c.num_samp_hand, c.num_samp_obj, c.bins_n = NH, NO, 64
torch.manual_seed(0)
model = get_model("train", cfg=c, mano_layer=MANO.ManoLayer(MANO.synthetic_assets(0)), with_encoder=False).to(DEV).eval()
model._jitter = lambda like, d: torch.zeros_like(like)
lv = [v.to(DEV).permute(0, 2, 3, 1).contiguous() for v in T.synthetic_pyramid(B, seed=3).values()]
batch = tuple(T.to_device(x, DEV) for x in T.synthetic_batch(B, NH, NO, seed=31))
other = tuple(T.to_device(x, DEV) for x in T.synthetic_batch(B, NH, NO, seed=977))
lo = [v.to(DEV).permute(0, 2, 3, 1).contiguous() for v in T.synthetic_pyramid(B, seed=55).values()]
def run(levels, bt, mode):
    with torch.no_grad():
        loss, out = model.hot_path(ops.PyramidNHWC(levels), *bt, mode, 0, 0.1)
    return {**loss, **out}
mixed = tuple({k: (torch.cat([v[:1], other[i][k][1:]]) if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in d.items()} for i, d in enumerate(batch))
for mode in sys.argv[1:] or ["train", "eval"]:
    a, b = run(lv, batch, mode), run(lv, batch, mode)
    m = run([torch.cat([l[:1], o[1:]]) for l, o in zip(lv, lo)], mixed, mode)
    for k, v in a.items():
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B:
            print(f"{mode:6s} {k:28s} rerun {'same' if torch.equal(v[0], b[k][0]) else 'DIFF %.2e' % float((v[0]-b[k][0]).abs().max()):14s}"
                  f" companions {'same' if torch.equal(v[0], m[k][0]) else 'DIFF %.2e' % float((v[0]-m[k][0]).abs().max())}")
