"""-m gpu: the hot path at BASELINE.json's full size (B = 32, 1536 + 512 points, C = 992), where the CPU oracle is too
slow to be the checker: size-independent properties of the domain instead.
  * samples are independent        -> permuting the batch permutes the outputs (catches cross-sample indexing);
  * the vote aggregation, the encoders and the MANO queries see the points as a SET
                                   -> permuting the hand / object points of a sample leaves joints, MANO outputs and the
                                      mean object pose unchanged;
  * sdf_infer (dense 64^3 lattice) -> every selected point projects strictly inside the bbox, |sdf| is ascending, no
                                      lattice node is selected twice, exactly K points per sample;
  * attention is linear in V and the GEMM is linear in x at the bench shapes."""
import pytest
import torch

from hoisdf_amd import testing as T
from hoisdf_amd.config import Config

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, NH, NO = 32, 1536, 512


@pytest.fixture(scope="module")
def setup():
    from hoisdf_amd import ops
    from hoisdf_amd.model import get_model
    from hoisdf_amd.nets import mano as MANO
    c = Config()
    c.resnet_type = 50
    c.apply_setting("dexycb")
    c.num_samp_hand, c.num_samp_obj, c.bins_n = NH, NO, 64
    torch.manual_seed(0)
    model = get_model("train", cfg=c, mano_layer=MANO.ManoLayer(MANO.synthetic_assets(0)), with_encoder=False).to(DEV)
    model.eval()                                                     # dropout off; mode="train" still selects branch A
    model._jitter = lambda like, d: torch.zeros_like(like)
    pyr = ops.PyramidNHWC([v.to(DEV).permute(0, 2, 3, 1).contiguous() for v in T.synthetic_pyramid(B, seed=3).values()])
    batch = tuple(T.to_device(x, DEV) for x in T.synthetic_batch(B, NH, NO, seed=31))
    return model, c, pyr, batch


def run(model, pyr, batch, mode="train"):
    with torch.no_grad():
        loss, out = model.hot_path(pyr, *batch, mode, 0, 0.1)
    return {**loss, **out}


def index_batch(d, perm):
    return {k: (v[perm] if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in d.items()}


KEYS = ("hand_joints_out", "mano_joints_out", "mano_mesh_out")


def test_batch_permutation_equivariance(setup):
    from hoisdf_amd import ops
    model, c, pyr, batch = setup
    ref = run(model, pyr, batch)
    for k in KEYS:                                                   # a degenerate output would pass trivially
        assert bool(torch.isfinite(ref[k]).all()) and float(ref[k].std()) > 1e-4
        assert float((ref[k][0] - ref[k][1]).abs().max()) > 1e-5    # samples really differ
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    pyr_p = ops.PyramidNHWC([l[perm].contiguous() for l in pyr.levels])
    got = run(model, pyr_p, tuple(index_batch(d, perm) for d in batch))
    for k in KEYS:
        # (round 6: bit for bit - a sample's arithmetic no longer depends on where in the batch it sits or who its neighbours are)
        assert torch.equal(got[k], ref[k][perm]), f"{k}: {(got[k] - ref[k][perm]).abs().max().item():.3e}"


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_a_samples_outputs_do_not_depend_on_its_batch_companions(setup, mode):
    """True of the reference (every op is per sample); true here since round 6 gave every row of a contraction's row operand and every
    (sample, head) of the attention operands its OWN f16x2 scale: sample 0's outputs are BIT-IDENTICAL whatever samples 1 .. 31
    are - other pyramids, other points, other cameras, and a x 300 louder batch (a sigma-gate / activation outlier next door used
    to coarsen everybody's rounding: one scale per matrix in round 5).  "train" = branch A on the given points (dropout off),
    "eval" = the dense-lattice sdf_infer path (ragged survivor counts per sample: rows of different samples share GEMM tiles)."""
    from hoisdf_amd import ops
    model, c, pyr, batch = setup
    ref = run(model, pyr, batch, mode)
    other = tuple(T.to_device(x, DEV) for x in T.synthetic_batch(B, NH, NO, seed=977))
    pyr_o = [v.to(DEV).permute(0, 2, 3, 1).contiguous() for v in T.synthetic_pyramid(B, seed=55).values()]
    for loud in (1.0, 300.0):
        levels = [torch.cat([l[:1], lo[1:] * loud]) for l, lo in zip(pyr.levels, pyr_o)]
        mixed = tuple({k: (torch.cat([v[:1], other[i][k][1:]]) if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in d.items()}
                      for i, d in enumerate(batch))
        got = run(model, ops.PyramidNHWC(levels), mixed, mode)
        n = 0
        for k, v in ref.items():
            if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B and k.endswith("_out"):
                assert float((got[k][1] - v[1]).abs().max()) > 0, k                        # the companions really changed
                assert torch.equal(got[k][0], v[0]), (mode, loud, k, float((got[k][0] - v[0]).abs().max()))
                n += 1
        assert n >= 3


def test_point_permutation_invariance(setup):
    model, c, pyr, batch = setup
    inputs, targets, meta = batch
    ref = run(model, pyr, batch)
    g = torch.Generator().manual_seed(2)
    ph, po = torch.randperm(NH, generator=g).to(DEV), torch.randperm(NO, generator=g).to(DEV)
    inp = dict(inputs)
    inp["hand_pre_points"], inp["obj_pre_points"] = inputs["hand_pre_points"][:, ph], inputs["obj_pre_points"][:, po]
    got = run(model, pyr, (inp, targets, meta))
    for k in KEYS:
        err = (got[k] - ref[k]).abs().max().item()
        assert err <= 1e-5, f"{k}: {err:.3e}"                        # metres; only the summation order over points changes
    for k in ("loss_joint_3d", "loss_joint_cls", "loss_all_joint_3d"):
        a, b = float(got[k].mean()), float(ref[k].mean())
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (k, a, b)


def test_sdf_infer_full_lattice_properties(setup):
    model, c, pyr, batch = setup
    _, _, meta = batch
    pts, sdf, pe, _ = model.sdf_infer(pyr, meta["mano_root"], meta["cam_intr"], meta["bbox_hand"], c.hand_sdf_scale, NH, "hand")
    assert pts.shape == (B, NH, 3) and sdf.shape == (B, NH, 1) and pe.shape == (B, NH, 30)
    a = sdf[..., 0].abs()
    assert bool((a[:, 1:] >= a[:, :-1]).all())                                        # ascending |sdf| (clamped values)
    assert bool((a <= c.ClampingDistance + 1e-7).all())
    cam = pts / c.hand_sdf_scale + meta["mano_root"][:, None]
    q = torch.einsum("bij,bpj->bpi", meta["cam_intr"], cam)
    uv = q[..., :2] / q[..., 2:]
    bb = meta["bbox_hand"][:, None]
    inside = (uv[..., 0] > bb[..., 0]) & (uv[..., 0] < bb[..., 2]) & (uv[..., 1] > bb[..., 1]) & (uv[..., 1] < bb[..., 3])
    assert bool(inside.all())                                                        # strict bbox filter (main/model.py:292-299)
    for b in range(0, B, 8):                                                         # no lattice node twice
        assert len({tuple(r) for r in pts[b].cpu().numpy().round(5).tolist()}) == NH


def test_linearity_at_bench_shapes():
    from hoisdf_amd import ops
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B, 2048, 768, generator=g).to(DEV)
    v2 = torch.randn(B, 2048, 256, generator=g).to(DEV)
    with torch.no_grad():
        o1 = ops.attention_self(qkv, 4)
        q2 = qkv.clone(); q2[..., 512:] = v2
        o2 = ops.attention_self(q2, 4)
        q3 = qkv.clone(); q3[..., 512:] = qkv[..., 512:] + v2
        o3 = ops.attention_self(q3, 4)
        assert (o3 - (o1 + o2)).abs().max().item() <= 2e-5 * max(1.0, o3.abs().max().item())   # linear in V
        x, y = torch.randn(65536, 992, generator=g).to(DEV), torch.randn(65536, 992, generator=g).to(DEV)
        W = (torch.randn(512, 992, generator=g) / 31.5).to(DEV)
        lhs = ops.linear(x + y, W, None)
        rhs = ops.linear(x, W, None) + ops.linear(y, W, None)
        assert (lhs - rhs).abs().max().item() <= 2e-5 * lhs.abs().max().item()
