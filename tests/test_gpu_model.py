"""-m gpu: the whole HIP hot path (hoisdf_amd.model.Model) against
  (a) the committed golden vectors captured from the real reference (tests/golden/g7_*, g8_*), and
  (b) the CPU oracle on the same seeded inputs (eval forward, train forward+backward with p = 0).
north_star tolerance: joint / vertex coordinates within 1e-4 abs (fp32, metres)."""
import math
import random

import pytest
import torch

from conftest import load_golden, pyramid_gradient_close
from hoisdf_amd import testing as T
from hoisdf_amd.config import Config

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(setting, nh, no, bins, train=False):
    from hoisdf_amd.model import get_model
    from hoisdf_amd.nets import mano as MANO
    c = Config()
    c.resnet_type = 18
    c.apply_setting(setting)
    c.num_samp_hand, c.num_samp_obj, c.bins_n = nh, no, bins
    model = get_model("test", cfg=c, mano_layer=MANO.ManoLayer(MANO.synthetic_assets(0)), with_encoder=False)
    sd = model.state_dict()
    for k in sd:
        if not k.startswith("mano_head"):
            sd[k] = T.det_param(k, sd[k].shape)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV)
    model.train(train)
    return model, c


def _needs_the_emulated_attention():
    """The saturated-attention fixtures (_smallbeta: first-layer scores of ~2^21 in the log2 domain, where one f32 ulp of a score is a
    factor 1.19 of P) pin the DEFAULT arithmetic, whose backward re-accumulates the scores bit-identically to its forward.  The exact-f32
    MFMA attention (HOISDF_ATTENTION=f32 / HOISDF_ATTN_BWD=f32, the round-2 kernels kept for A/B runs) sums them in another order
    forward and backward: joints 1.5e-4 off and first-layer gradients without meaning on this fixture (measured, round 6)."""
    import os
    from hoisdf_amd import ops
    if not (ops.gemm_emu() and ops.attention_emu()) or os.environ.get("HOISDF_ATTN_BWD") == "f32":
        pytest.skip("saturated-attention fixture: holds for the emulated (default) attention, not for the exact-f32 MFMA kernels")


def nhwc_pyramid(pyr, requires_grad=False):
    from hoisdf_amd import ops
    lv = [v.to(DEV).permute(0, 2, 3, 1).contiguous().requires_grad_(requires_grad) for v in pyr.values()]
    return ops.PyramidNHWC(lv), lv


def oracle_cfg(c, **kw):
    from oracle import hoisdf_oracle as R
    return R.OracleCfg(num_samp_hand=c.num_samp_hand, num_samp_obj=c.num_samp_obj, bins_n=c.bins_n,
                       use_inverse_kinematics=c.use_inverse_kinematics, dataset=c.dataset, **kw)


E2E = [("dexycb", False, 48, 16, 16, 2), ("ho3d", True, 48, 16, 16, 2), ("ho3d_render", False, 48, 16, 16, 2),
       ("dexycb", False, 384, 128, 64, 1), ("ho3d_render", False, 384, 128, 64, 1), ("ho3d", True, 384, 128, 64, 1),
       # BASELINE.json sizes: configs[1] points, configs[3] (IK variant, 4096 points), configs[4] (8192 points)
       ("dexycb", False, 1536, 512, 64, 2), ("ho3d_render", False, 3072, 1024, 64, 1), ("dexycb", False, 6144, 2048, 64, 1),
       # round 6: trained-like statistics (betas 2e-3 / 1e-2: sigma gates up to 500; x 100 outlier channels in the pyramid)
       ("dexycb", False, 1536, 512, 64, 2, "_smallbeta")]


@pytest.mark.parametrize("case", E2E, ids=lambda c: "-".join(str(x) for x in c))
def test_eval_forward_matches_reference_goldens(case):
    setting, big, nh, no, bins, b = case[:6]
    sfx = case[6] if len(case) > 6 else ""
    g = load_golden(f"g7_e2e_{setting}_n{nh + no}{sfx}")
    model, c = build(setting, nh, no, bins)
    if sfx == "_smallbeta":
        _needs_the_emulated_attention()
        with torch.no_grad():
            for k_, v_ in T.SMALL_BETA.items():
                getattr(model, k_).fill_(v_)
    pyr, _ = nhwc_pyramid(T.synthetic_pyramid(b, big=big, seed=2, outliers=100.0 if sfx == "_smallbeta" else 1.0))
    inputs, targets, meta = T.synthetic_batch(b, nh, no, seed=21)
    if bins == 16:
        meta["bbox_hand"] = torch.tensor([0.0, 0, 256, 256]).repeat(b, 1)
        meta["bbox_obj"] = torch.tensor([0.0, 0, 256, 256]).repeat(b, 1)
    inputs, targets, meta = (T.to_device(x, DEV) for x in (inputs, targets, meta))
    with torch.no_grad():
        loss, out = model.hot_path(pyr, inputs, targets, meta, "eval")
    res = {**loss, **out}
    n_checked = 0
    for k, ref in g.items():
        if k.endswith("_mean"):                              # big fixtures hold the per-point outputs as means over points
            k, got = k[:-5], res[k[:-5]].float().cpu().mean(1)
            assert (got - ref).abs().max().item() <= 1e-4, k
            n_checked += 1
            continue
        assert k in res, f"missing output {k}"
        got = res[k].float().cpu()
        if k in ("obj_rot_out", "obj_trans_out"):           # per-point rows follow the |sdf| order: compare means
            got, ref = got.mean(1), ref.mean(1)
        if k in ("hand_joints_out", "mano_joints_out", "mano_mesh_out", "mano_joints_gt_out", "mano_mesh_gt_out"):
            tol = 1e-4                                       # the north-star bar (metres)
        elif "loss" in k or k in ("obj_rot", "obj_trans"):
            tol = 2e-4 * max(1.0, float(ref.abs().max()))
        else:
            tol = 1e-4
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        err = (got - ref).abs().nan_to_num(0.0).max().item()
        assert torch.isnan(got).equal(torch.isnan(ref)), k
        assert err <= tol, f"{k}: max abs err {err:.3e} > {tol:.1e}"
        n_checked += 1
    assert n_checked >= 6


def test_sdf_infer_selects_the_oracle_set():
    from oracle import hoisdf_oracle as R
    nh, no, bins, b = 384, 128, 64, 2
    model, c = build("dexycb", nh, no, bins)
    P = T.det_params(T.hot_path_param_shapes(992))
    pyr_cpu = T.synthetic_pyramid(b, seed=8)
    pyr, _ = nhwc_pyramid(pyr_cpu)
    _, _, meta = T.synthetic_batch(b, nh, no, seed=81)
    pts_r, sdf_r, pe_r, dbg = R.sdf_infer(P, oracle_cfg(c), pyr_cpu, meta["mano_root"], meta["cam_intr"],
                                          meta["bbox_hand"], 3.1, nh, "hand", return_debug=True)
    m = T.to_device(meta, DEV)
    pts, sdf, pe, _ = model.sdf_infer(pyr, m["mano_root"], m["cam_intr"], m["bbox_hand"], 3.1, nh, "hand")
    for i in range(b):
        sa = {tuple(r) for r in pts[i].cpu().numpy().round(6).tolist()}
        sb = {tuple(r) for r in pts_r[i].numpy().round(6).tolist()}
        # candidates whose |sdf| sits within fp32 noise of the k-th value may swap
        assert len(sa ^ sb) <= 4, len(sa ^ sb)
        assert abs(float(sdf[i].abs().sum().cpu() - sdf_r[i].abs().sum())) < 1e-3
        assert bool((sdf[i, 1:, 0].abs() >= sdf[i, :-1, 0].abs() - 1e-7).all())      # ascending |sdf|
    with pytest.raises(ValueError):
        tiny = torch.tensor([100.0, 100, 101, 101], device=DEV).repeat(b, 1)
        model.sdf_infer(pyr, m["mano_root"], m["cam_intr"], tiny, 3.1, nh, "hand")


def test_sdf_infer_in_train_mode_ranks_under_dropout_like_the_reference():
    """Branch B of a TRAINING step (main/model.py:462-481): the reference runs sdf_infer under no_grad but with the module in train
    mode, so the SDF decoder's dropout (p = 0.2, common/nets/sdf_net.py:112-113) is live while the lattice points are ranked - the
    selected set is random (two oracle runs with independent masks share ~2 % of their 384 points on these weights).  The
    device's mask stream is not torch's, so the check is statistical, against three oracle runs: the order statistics of the
    NOISY |sdf| that ranked the points (mean and K-th smallest over ~29 000 candidates: tightly concentrated), the CLEAN |sdf| of
    the selected points (how far from the surface the noise lets the selection drift: 0.11 against 0.0017 without dropout), the
    overlaps between selections; eval mode returns the deterministic set again."""
    from oracle import hoisdf_oracle as R
    nh, no, bins, b = 384, 128, 64, 2
    model, c = build("dexycb", nh, no, bins, train=True)
    P = T.det_params(T.hot_path_param_shapes(992))
    pyr_cpu = T.synthetic_pyramid(b, seed=8)
    pyr, _ = nhwc_pyramid(pyr_cpu)
    _, _, meta = T.synthetic_batch(b, nh, no, seed=81)
    oc = oracle_cfg(c)
    key = lambda r: tuple(round(float(x), 6) for x in r)

    def oracle_run(training, seed):
        torch.manual_seed(seed)
        return R.sdf_infer(P, oc, pyr_cpu, meta["mano_root"], meta["cam_intr"], meta["bbox_hand"], 3.1, nh, "hand",
                           return_debug=True, training=training)

    pts_c, sdf_c, _, dbg_c = oracle_run(False, 0)
    lattice = R.dense_lattice(bins)
    clean_of = [{key(p): float(v) for p, v in zip(lattice[d["keep"]].tolist(), d["sdf"].abs().tolist())} for d in dbg_c]
    noisy = [oracle_run(True, seed) for seed in (1, 2, 3)]
    m = T.to_device(meta, DEV)
    assert model.hand_sdf_decoder.training
    # (the three statistics of ONE draw scatter by +-10 % around their expectation, the oracle's as much as the device's - measured over
    # 12 device seeds and 6 oracle seeds, tools/dbg_branchb_spread.py: single draws are held to a gross-error band, the AVERAGE of six
    # draws to the oracle's average; the device's mask stream is seeded here so that the test does not depend on what ran before it)
    from hoisdf_amd import ops as _ops
    _ops.manual_seed(20260930)
    runs = [model.sdf_infer(pyr, m["mano_root"], m["cam_intr"], m["bbox_hand"], 3.1, nh, "hand") for _ in range(6)]
    for i in range(b):
        o_sets = [{key(r) for r in n[0][i].tolist()} for n in noisy]
        d_sets = [{key(r) for r in r_[0][i].cpu().tolist()} for r_ in runs]
        o_mean = sum(float(n[1][i].abs().mean()) for n in noisy) / 3
        o_kth = sum(float(n[1][i].abs().max()) for n in noisy) / 3
        o_clean = sum(sum(clean_of[i][k] for k in st) / nh for st in o_sets) / 3
        assert o_clean > 20 * float(sdf_c[i].abs().mean())                 # (the oracle itself: dropout moves the selection off the surface)
        d_mean, d_kth, d_clean = [], [], []
        for r_, st in zip(runs, d_sets):
            sdf = r_[1][i, :, 0].abs().cpu()
            assert bool((sdf[1:] >= sdf[:-1] - 1e-7).all())                  # ranked by the noisy values it returns
            d_mean.append(float(sdf.mean())); d_kth.append(float(sdf.max()))
            d_clean.append(sum(clean_of[i][k] for k in st) / nh)             # (every selected point is a lattice survivor: KeyError otherwise)
            assert abs(d_mean[-1] - o_mean) <= 0.35 * o_mean and abs(d_kth[-1] - o_kth) <= 0.35 * o_kth, (d_mean[-1], o_mean, d_kth[-1], o_kth)
            assert abs(d_clean[-1] - o_clean) <= 0.35 * o_clean, (d_clean[-1], o_clean)
        avg = lambda xs: sum(xs) / len(xs)
        assert abs(avg(d_mean) - o_mean) <= 0.10 * o_mean, (avg(d_mean), o_mean)
        assert abs(avg(d_kth) - o_kth) <= 0.10 * o_kth, (avg(d_kth), o_kth)
        assert abs(avg(d_clean) - o_clean) <= 0.10 * o_clean, (avg(d_clean), o_clean)
        ov = lambda x, y: len(x & y) / nh
        oo = max(ov(o_sets[0], o_sets[1]), ov(o_sets[0], o_sets[2]), ov(o_sets[1], o_sets[2]))
        assert ov(d_sets[0], d_sets[1]) <= oo + 0.05 and ov(d_sets[0], o_sets[0]) <= oo + 0.05, "selections as random as the oracle's"
        assert d_sets[0] != d_sets[1], "two calls draw two masks"
    model.eval()
    pts_e, _, _, _ = model.sdf_infer(pyr, m["mano_root"], m["cam_intr"], m["bbox_hand"], 3.1, nh, "hand")
    for i in range(b):
        assert len({key(r) for r in pts_e[i].cpu().tolist()} ^ {key(r) for r in pts_c[i].tolist()}) <= 4


@pytest.mark.parametrize("setting,nh,no,suffix", [("dexycb", 48, 16, ""), ("ho3d_render", 48, 16, ""), ("ho3d", 48, 16, ""),
                                                  ("dexycb", 1536, 512, "_n2048"), ("dexycb", 48, 16, "_branchB"),
                                                  ("dexycb", 1536, 512, "_n2048_smallbeta"), ("dexycb", 1536, 512, "_n2048_trainedlike")])
def test_train_fwd_bwd_matches_reference_goldens_and_oracle(setting, nh, no, suffix):
    """branch A (pre-points + jitter), every dropout p = 0: losses and gradients vs g8 goldens (the _n2048 fixture is the
    reference's own fwd+bwd at BASELINE configs[1]'s 1536+512 points).  _branchB: the training step after
    cfg.point_sampling_epoch with the draw p >= 0.4 - query points from the dense-lattice sdf_infer (main/model.py:470-481).
    _n2048_smallbeta (round 6): trained-like statistics - betas 2e-3 / 1e-2 (sigma up to 500: token rows spanning ~30 decades inside
    one matrix) and x 100 outlier channels in the pyramid.  There fp32 ITSELF is ill-conditioned: the reference's own fp32 gradient
    norms sit up to 1.3e-3 (hand_sigmoid_beta: 11.7 %) away from the fp64 values of the pinned oracle (tools/fp64_truth_smallbeta.py),
    so gradient norms are held two-sided: within 1e-3 of the fp64 truth, or no further from it than 1.5 x the reference's own fp32.
    With det_param's unit-gain q / k weights those tokens give FIRST-LAYER attention scores of 7e6 (hand) / 2.4e8 (object) in the log2
    domain (tools/dbg_smallbeta.py): one fp32 ulp of such a score is 0.5 / 16, the softmax is one-hot by rounding, and a flash-style
    backward - P recomputed from S and the saved LSE, delta = rowsum(dO o O) - cannot reproduce the EXACT cancellation PyTorch gets from
    its materialised P (dS = P (dP - sum P dP) = 0 bit for bit on a one-hot row); the exact-f32 kernels are off by 1e9 x on the beta
    gradient there, the f16x2 form (scores re-accumulated in the forward's product order since round 6) by 4-6 % on what lies upstream
    of that attention: the token MLP, the betas, the first in-projections and the pyramid gradient.  _n2048_smallbeta holds those at
    0.1 and everything else at the bar; _n2048_trainedlike = the same betas and outlier channels with the first-layer q / k projections at
    the scale a trained network has them (testing.TRAINED_LIKE_QK: scores of O(10-100)) holds EVERY gradient norm at 1e-3."""
    g = load_golden(f"g8_train_{setting}{suffix}")
    small = suffix.endswith(("_smallbeta", "_trainedlike"))
    saturated = suffix.endswith("_smallbeta")
    if saturated:
        _needs_the_emulated_attention()
    upstream = ("linear_transformerin.", "hand_sigmoid_beta", "obj_sigmoid_beta", "hand_transformer.encoder.layers.0.self_attn.in_proj",
                "obj_transformer.encoder.layers.0.self_attn.in_proj")
    g64 = load_golden(f"g8_train_{setting}{suffix}_fp64") if small else None
    epoch_cnt = 10 ** 8 if suffix == "_branchB" else 0
    b = 2
    model, c = build(setting, nh, no, 16, train=True)
    if small:
        with torch.no_grad():
            for k_, v_ in T.SMALL_BETA.items():
                getattr(model, k_).fill_(v_)
        if suffix.endswith("_trainedlike"):
            sd_ = dict(model.named_parameters())
            T.apply_trained_like(lambda n_: sd_[n_])
    c.dropout = 0.0
    for m in model.modules():
        if hasattr(m, "p"):
            m.p = 0.0
        if hasattr(m, "dropout_prob"):
            m.dropout_prob = 0.0
    pyr, levels = nhwc_pyramid(T.synthetic_pyramid(b, big=setting == "ho3d", seed=3, outliers=100.0 if small else 1.0), requires_grad=True)   # "ho3d": C = 3968
    inputs, targets, meta = T.synthetic_batch(b, nh, no, seed=31)
    # reproduce the reference's CPU jitter stream (torch.manual_seed(1234): hand first, then obj)
    torch.manual_seed(1234)
    jit = [torch.empty_like(inputs["hand_pre_points"]).uniform_(-0.05, 0.05),
           torch.empty_like(inputs["obj_pre_points"]).uniform_(-0.05, 0.05)]
    model._jitter = lambda like, d: jit.pop(0).to(DEV)
    model._py_random = random.Random(0)
    inputs, targets, meta = (T.to_device(x, DEV) for x in (inputs, targets, meta))
    loss, out = model.hot_path(pyr, inputs, targets, meta, "train", epoch_cnt, 0.5)
    losses = {k: v.mean() for k, v in loss.items()}
    for k, v in losses.items():
        ref = g["loss." + k]
        assert abs(float(v) - float(ref)) <= 1e-4 * max(1.0, abs(float(ref))), (k, float(v), float(ref))
    total = sum(losses.values())
    assert abs(float(total) - float(g["total"])) <= 1e-4 * abs(float(g["total"]))
    total.backward()
    n = 0
    for name, p in model.named_parameters():
        key = "gradnorm." + name
        if key in g:
            assert p.grad is not None, name
            gn = p.grad.double().norm().item()
            # branch B: the scalar beta gradients are heavily cancelling sums over near-surface points (two fp32 summation
            # orders on the CPU already differ by 1.3e-3 there, tests/test_oracle_golden.py); since round 4 the device sums
            # them in a fixed block order (hoisdf_token_build_bwd_ordered) - the bar is the CPU's own order noise, 3e-3
            rt = 3e-3 if (suffix == "_branchB" and name.endswith("sigmoid_beta")) else 1e-3
            if small:
                t64, ref = float(g64[key]), float(g[key])
                if saturated and name == "hand_sigmoid_beta":
                    # a cancelling scalar sum over the very tokens whose attention is decided by rounding: 3e3 ... 1.3e5 from run to run
                    # (the gather backward's atomics reorder the noise) around fp64's 1.3e4 - only finiteness can be asked of it
                    assert math.isfinite(gn), gn
                    n += 1
                    continue
                if saturated and name.startswith(upstream):
                    rt = 0.15
                assert abs(gn - t64) <= max(rt * abs(t64), 1.5 * abs(ref - t64)) + 1e-6, (name, gn, t64, ref)
            else:
                assert abs(gn - float(g[key])) <= rt * float(g[key]) + 1e-6, (name, gn, float(g[key]))
            n += 1
        elif not name.startswith(("backbone", "decoder_net")):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    assert n > 100

    def gclose(a, ref, rel=1e-3):
        err = (a.float().cpu() - ref).abs().max().item()
        assert err <= rel * float(ref.abs().max()) + 1e-9, err

    brel = 3e-3 if suffix == "_branchB" else 1e-3
    if not small:                   # (small betas: the two scalars are covered by the two-sided gradient-norm rule above)
        gclose(model.hand_sigmoid_beta.grad, g["grad.hand_sigmoid_beta"], brel)
        gclose(model.obj_sigmoid_beta.grad, g["grad.obj_sigmoid_beta"], brel)
    gclose(model.linear_handcls.layers[2].weight.grad, g["grad.linear_handcls.layers.2.weight"])
    gclose(model.hand_sdf_decoder.linh0.weight_g.grad, g["grad.hand_sdf_decoder.linh0.weight_g"])
    gn2 = levels[0].grad.double().norm().item()
    t2 = float(g64["grad.pyr.stride2_norm"]) if small else float(g["grad.pyr.stride2_norm"])
    if saturated:                   # (the pyramid gradient is upstream of the saturated first-layer attention: see the docstring)
        assert abs(gn2 - t2) <= 0.15 * t2, (gn2, t2)
        return
    pyramid_gradient_close(levels[4].grad.permute(0, 3, 1, 2)[:, ::16], g["grad.pyr.stride32"], suffix)
    assert abs(gn2 - t2) <= max(1e-3 * t2, 1.5 * abs(float(g["grad.pyr.stride2_norm"]) - t2)), (gn2, t2)


def test_transformer_seq_first_surface_matches_golden():
    """the reference-signature entry points (seq-first Transformer.forward / VoteTransformer.forward)."""
    from hoisdf_amd.model import get_mano_memory_mask, get_mano_tgt_mask
    g = load_golden("g5_transformer")
    model, c = build("dexycb", 48, 16, 16)
    src = g["src"].to(DEV)
    with torch.no_grad():
        hs, mem, inter, _ = model.hand_transformer(src=src, mask=None, pos_embed=torch.zeros_like(src), src_mask=None,
                                                   query_embed=model.mano_query_embed.weight,
                                                   tgt_mask=get_mano_tgt_mask(c), memory_mask=get_mano_memory_mask(c))
        omem, ointer = model.obj_transformer(src=src, mask=None, pos_embed=torch.zeros_like(src), src_mask=None)
    for a, k in ((hs, "hs"), (mem, "memory"), (inter, "inter"), (omem, "obj_memory"), (ointer, "obj_inter")):
        err = (a.cpu() - g[k]).abs().max().item()
        assert err <= 1e-4, (k, err)


def build_opts(nh, no, bins, train=False, **opts):
    """build() with reference config switches set (cfg.pre_norm / cfg.ClassifierBranch, main/config.py:122,91)"""
    from hoisdf_amd.model import get_model
    from hoisdf_amd.nets import mano as MANO
    c = Config()
    c.resnet_type = 18
    c.apply_setting("dexycb")
    c.num_samp_hand, c.num_samp_obj, c.bins_n = nh, no, bins
    for k, v in opts.items():
        assert hasattr(c, k), k
        setattr(c, k, v)
    model = get_model("test", cfg=c, mano_layer=MANO.ManoLayer(MANO.synthetic_assets(0)), with_encoder=False)
    sd = model.state_dict()
    for k in sd:
        if not k.startswith("mano_head"):
            sd[k] = T.det_param(k, sd[k].shape)
    model.load_state_dict(sd, strict=True)
    return model.to(DEV).train(train), c


def test_pre_norm_transformers_match_the_reference_fixture():
    """cfg.pre_norm = True (common/nets/transformer.py:304-331,397-437 + encoder.norm): the reference's own outputs (g5p), and
    the state dict carries encoder.norm like the reference's"""
    from hoisdf_amd.model import get_mano_memory_mask, get_mano_tgt_mask
    import json, os
    g = load_golden("g5p_transformer_prenorm")
    model, c = build_opts(48, 16, 16, pre_norm=True)
    ref_keys = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g10_state_dict_options.json")))["pre_norm"]
    ours = {k for k in model.state_dict() if not k.startswith(("mano_head", "backbone_net", "decoder_net"))}
    assert ours == {k for k in ref_keys if not k.startswith("mano_head")}, sorted(ours ^ set(ref_keys))[:6]
    src = g["src"].to(DEV)
    with torch.no_grad():
        hs, mem, inter, _ = model.hand_transformer(src=src, mask=None, pos_embed=torch.zeros_like(src), src_mask=None,
                                                   query_embed=model.mano_query_embed.weight,
                                                   tgt_mask=get_mano_tgt_mask(c), memory_mask=get_mano_memory_mask(c))
        omem, ointer = model.obj_transformer(src=src, mask=None, pos_embed=torch.zeros_like(src), src_mask=None)
    for a, k in ((hs, "hs"), (mem, "memory"), (inter, "inter"), (omem, "obj_memory"), (ointer, "obj_inter")):
        err = (a.cpu() - g[k]).abs().max().item()
        assert err <= 2e-4, (k, err)


def test_pre_norm_backward_matches_the_oracle():
    """gradients of a pre-norm encoder stack + decoder (dropout off) against autograd of the CPU oracle on the same weights"""
    from oracle import hoisdf_oracle as R
    from hoisdf_amd.model import get_mano_memory_mask, get_mano_tgt_mask
    g = load_golden("g5p_transformer_prenorm")
    model, c = build_opts(48, 16, 16, train=True, pre_norm=True)
    for m in model.modules():
        if hasattr(m, "p"):
            m.p = 0.0
    src = g["src"].to(DEV).requires_grad_(True)
    hs, mem, inter, _ = model.hand_transformer(src=src, mask=None, pos_embed=torch.zeros_like(src), src_mask=None,
                                               query_embed=model.mano_query_embed.weight,
                                               tgt_mask=get_mano_tgt_mask(c), memory_mask=get_mano_memory_mask(c))
    w = torch.linspace(-1, 1, hs.numel(), device=DEV).view_as(hs)
    ((hs * w).sum() + inter.pow(2).mean()).backward()
    P = {k: v.clone().requires_grad_(True) for k, v in T.det_params(T.hot_path_param_shapes(992, pre_norm=True)).items()}
    ocfg = R.OracleCfg(num_samp_hand=48, num_samp_obj=16, pre_norm=True)
    s2 = g["src"].clone().requires_grad_(True)
    m2, i2 = R.encoder(s2, P, "hand_transformer.encoder", 6, ocfg, False)
    h2 = R.decoder(m2, P["mano_query_embed.weight"], P, "hand_transformer.decoder", 4, ocfg, R.mano_tgt_mask(), R.memory_mask(17, 48, 16), False)
    ((h2 * w.cpu()).sum() + i2.pow(2).mean()).backward()
    def rel(a, b):
        return float((a.cpu() - b).abs().max()) / max(float(b.abs().max()), 1e-30)
    assert rel(src.grad, s2.grad) <= 2e-4
    sd = dict(model.named_parameters())
    for k in ("hand_transformer.encoder.norm.weight", "hand_transformer.encoder.layers.0.self_attn.in_proj_weight",
              "hand_transformer.encoder.layers.5.linear2.weight", "hand_transformer.decoder.layers.0.norm1.weight",
              "hand_transformer.decoder.layers.3.multihead_attn.in_proj_weight", "hand_transformer.encoder.inter_norm.bias"):
        assert rel(sd[k].grad, P[k].grad) <= 5e-4, (k, rel(sd[k].grad, P[k].grad))


def test_classifier_branch_matches_the_reference_fixture():
    """cfg.ClassifierBranch = True (common/nets/sdf_net.py:73-75,93-94,119-122; main/model.py:236-240,351-352): the decoder's
    class logits against the reference's own, the SDF unchanged, sdf_infer hands the logits of its selected points back"""
    import json, os
    g = load_golden("g2c_sdf_decoder_cls")
    model, c = build_opts(48, 16, 16, ClassifierBranch=True)
    ref_keys = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g10_state_dict_options.json")))["classifier"]
    ours = {k for k in model.state_dict() if not k.startswith(("mano_head", "backbone_net", "decoder_net"))}
    assert ours == {k for k in ref_keys if not k.startswith("mano_head")}
    with torch.no_grad():
        y, cls = model.hand_sdf_decoder(g["x"].to(DEV))
    assert float((y.cpu() - g["y"]).abs().max()) <= 2e-5 and float((cls.cpu() - g["cls"]).abs().max()) <= 1e-4
    pyr = T.synthetic_pyramid(2, big=False, seed=1)
    inputs, targets, meta = T.synthetic_batch(2, 48, 16, seed=11)
    P, _ = nhwc_pyramid(pyr)
    di, dm = T.to_device(inputs, DEV), T.to_device(meta, DEV)
    with torch.no_grad():
        sh, ch, peh = model.sdf_forward(P, di["hand_sdf_points"], dm["mano_root"], dm["cam_intr"], c.hand_sdf_scale, "hand")
    assert float((sh.cpu() - g["sdf_hand"]).abs().max()) <= 2e-5
    assert float((ch.cpu() - g["cls_hand"]).abs().max()) <= 1e-4 and ch.shape == (2, 48, 6)
    dm["bbox_hand"] = torch.tensor([20.0, 20, 236, 236], device=DEV).repeat(2, 1)
    pts, sdf, pe, pcls = model.sdf_infer(P, dm["mano_root"], dm["cam_intr"], dm["bbox_hand"], c.hand_sdf_scale, 24, "hand")
    assert pcls.shape == (2, 24, 6)
    with torch.no_grad():
        _, again, _ = model.sdf_forward(P, pts, dm["mano_root"], dm["cam_intr"], c.hand_sdf_scale, "hand")
    assert float((pcls - again).abs().max()) <= 1e-5          # (few-tile GEMMs: summation order is not fixed)


@pytest.mark.parametrize("opts", [dict(pre_norm=True), dict(ClassifierBranch=True)])
def test_training_step_with_the_option_switches_matches_the_oracle(opts):
    """a whole hot-path train step (dropout off) with either switch on: losses and a few gradients against the CPU oracle"""
    from oracle import hoisdf_oracle as R
    nh, no, b = 48, 16, 2
    model, c = build_opts(nh, no, 16, train=True, **opts)
    for m in model.modules():
        if hasattr(m, "p"):
            m.p = 0.0
        if hasattr(m, "dropout_prob"):
            m.dropout_prob = 0.0
    pyr = T.synthetic_pyramid(b, big=False, seed=1)
    inputs, targets, meta = T.synthetic_batch(b, nh, no, seed=11)
    P_, _ = nhwc_pyramid(pyr)
    di, dt, dm = (T.to_device(x, DEV) for x in (inputs, targets, meta))
    # the oracle's CPU jitter stream (torch.manual_seed(1234): hand first, then obj), handed to the model through its test hook
    torch.manual_seed(1234)
    jit = [torch.empty_like(inputs["hand_pre_points"]).uniform_(-0.05, 0.05), torch.empty_like(inputs["obj_pre_points"]).uniform_(-0.05, 0.05)]
    model._jitter = lambda like, d: jit.pop(0).to(DEV)
    model._py_random = random.Random(0)
    loss, out = model.hot_path(P_, di, dt, dm, "train", 0, 0.5)
    total = sum(v.mean() for v in loss.values())
    total.backward()
    Pm = {k: v.clone().requires_grad_(True)
          for k, v in T.det_params(T.hot_path_param_shapes(992, pre_norm=bool(opts.get("pre_norm")), classifier=bool(opts.get("ClassifierBranch")))).items()}
    ocfg = oracle_cfg(c, dropout=0.0, sdf_dropout=0.0, pre_norm=bool(opts.get("pre_norm")), ClassifierBranch=bool(opts.get("ClassifierBranch")))
    from hoisdf_amd.nets import mano as MANO
    layer_cpu = MANO.ManoLayer(MANO.synthetic_assets(0))
    torch.manual_seed(1234)
    ref = R.hot_path_forward(Pm, ocfg, pyr, inputs, targets, meta, "train", mano_layer=layer_cpu, hands_mean=layer_cpu.th_hands_mean,
                             epoch_cnt=0, batch_ratio=0.5, rng=random.Random(0))
    rl = {k: v for k, v in ref.items() if not k.endswith("_out")}
    for k, v in loss.items():
        a, r_ = float(v.mean()), float(rl[k].mean())
        assert abs(a - r_) <= 2e-4 * max(1.0, abs(r_)), (k, a, r_)
    sum(v.mean() for v in rl.values()).backward()
    sd = dict(model.named_parameters())
    for k in ("linear_transformerin.layers.0.weight", "hand_transformer.encoder.layers.0.linear1.weight", "obj_transformer.encoder.layers.2.norm2.weight",
              "hand_sdf_decoder.linh4.weight"):
        ga, gr = sd[k].grad.cpu(), Pm[k].grad
        assert float((ga - gr).norm()) <= 2e-3 * float(gr.norm()) + 1e-9, (k, float((ga - gr).norm()), float(gr.norm()))
    if opts.get("ClassifierBranch"):
        assert sd["hand_sdf_decoder.classifier_head.weight"].grad is None          # nothing reads the logits (as in the reference)


def test_full_model_with_encoder_runs_and_is_finite():
    """ResNet-18 encoder (PyTorch/MIOpen) + HIP hot path, one train step with dropout ON."""
    from hoisdf_amd.model import get_model
    c = Config()
    c.resnet_type = 18
    c.apply_setting("dexycb")
    c.num_samp_hand, c.num_samp_obj = 96, 32
    model = get_model("train", cfg=c).to(DEV).train()
    inputs, targets, meta = (T.to_device(x, DEV) for x in T.synthetic_batch(2, 96, 32, seed=5))
    out = model(inputs, targets, meta, "train", 0, 0.1)
    loss = sum(v.mean() for k, v in out.items() if "_out" not in k)
    loss.backward()
    assert torch.isfinite(loss)
    assert out["hand_joints_out"].shape == (2, 20, 3) and out["mano_mesh_out"].shape == (2, 778, 3)
    g = model.linear_sdfin.layers[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
    assert model.backbone_net.resnet.conv1.weight.grad is not None


def test_bench_distributed_path_single_rank_rccl():
    """The N>1 code path of bench.py (RCCL init, bucketed async all-reduce from autograd hooks, barrier + MAX
    reduction of the timing) on one GPU: world_size 1, launched through torch.distributed.run like the driver does."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29641", os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--batch", "4", "--n-hand", "192", "--n-obj", "64", "--resnet", "18", "--no-cpu-baseline",
           "--force-dist", "--miopen-find", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=repo)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["config"]["parallelism"] == "dp1"


def test_bench_two_ranks_control_flow_on_one_gpu():
    """bench.py launched exactly like the driver launches it for N = 2 (torch.distributed.run, --gpus 2): rendezvous,
    identical-weights / distinct-seeds asserts, per-rank synthetic shards, bucketed all-reduce from the hooks, barrier +
    MAX-over-ranks timing, one JSON line from rank 0 with the whole-job rate.  One GPU here, so both ranks sit on cuda:0
    with gloo as the transport (HOISDF_BENCH_ONE_GPU_GLOO=1): a functional check of the control flow, not a measurement."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HOISDF_BENCH_ONE_GPU_GLOO="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29643", os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "4", "--n-hand", "192", "--n-obj", "64", "--resnet", "18", "--no-cpu-baseline",
           "--miopen-find", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 8
    assert res["scaling"] == "weak" and res["value"] > 0
    assert abs(res["value"] - 8 / (res["ms_per_step"] * 1e-3)) <= 1e-2 * res["value"]     # whole-job samples / max-rank time


@pytest.mark.parametrize("nh,no", [(384, 128), (6144, 2048)])
def test_eval_with_f16_mfma_attention_meets_the_joint_bar(nh, no):
    """BASELINE configs[4] ("fp16 MFMA attention"): with the f16-operand attention kernel switched on, the eval
    forward still matches the REFERENCE golden joints / vertices within the north-star 1e-4 m - at 512 points and at
    configs[4]'s own 6144 + 2048 = 8192 points (g7_e2e_dexycb_n8192, the reference's forward at that size)."""
    from hoisdf_amd import ops
    setting, bins, b = "dexycb", 64, 1
    g = load_golden(f"g7_e2e_{setting}_n{nh + no}")
    model, c = build(setting, nh, no, bins)
    pyr, _ = nhwc_pyramid(T.synthetic_pyramid(b, big=False, seed=2))
    inputs, targets, meta = (T.to_device(x, DEV) for x in T.synthetic_batch(b, nh, no, seed=21))
    ops.set_attention_f16_eval(True)
    try:
        with torch.no_grad():
            loss, out = model.hot_path(pyr, inputs, targets, meta, "eval")
    finally:
        ops.set_attention_f16_eval(False)
    for k in ("hand_joints_out", "mano_joints_out", "mano_mesh_out"):
        err = (out[k].float().cpu() - g[k]).abs().max().item()
        assert err <= 1e-4, f"{k}: {err:.3e}"
    # the switches really select other kernels (the fused encoder-layer node once bypassed the f16 attention silently)
    with torch.no_grad():
        _, out32 = model.hot_path(pyr, inputs, targets, meta, "eval")
    assert not torch.equal(out32["hand_joints_out"], out["hand_joints_out"])


def test_c_host_allreduce():
    """A plain C host (no Python, no torch) drives libhoisdf_rccl.so: RCCL communicator with one rank + all-reduce."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = "/tmp/hoisdf_test_allreduce"
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", os.path.join(repo, "tests", "c", "test_allreduce.c"), "-I", os.path.join(repo, "include"),
                    "-L", os.path.join(repo, "hoisdf_amd"), "-lhoisdf_rccl", "-Wl,-rpath," + os.path.join(repo, "hoisdf_amd"),
                    "-o", exe], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "allreduce ok" in out.stdout, out.stdout + out.stderr


def test_c_host_allreduce_two_ranks_file_rendezvous():
    """libhoisdf_rccl.so with TWO ranks and a file rendezvous of the 256-byte token (the id-exchange path a launcher without
    torch uses).  Both ranks must end up with the same token.  With two GPUs the all-reduce itself is checked; on a one-GPU box
    RCCL refuses (or stalls on) two ranks on one device - then the token plumbing is what this test pins, and the processes are
    ended after a short wait."""
    import os
    import subprocess
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = "/tmp/hoisdf_test_allreduce_w2"
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", os.path.join(repo, "tests", "c", "test_allreduce_world2.c"), "-I",
                    os.path.join(repo, "include"), "-L", os.path.join(repo, "hoisdf_amd"), "-lhoisdf_rccl",
                    "-Wl,-rpath," + os.path.join(repo, "hoisdf_amd"), "-o", exe], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "id.bin")
        procs = [subprocess.Popen([exe, str(r), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
                 for r in (0, 1)]
        outs = []
        for p in procs:
            try:
                o, e = p.communicate(timeout=90)
            except subprocess.TimeoutExpired:           # two ranks on one device: RCCL waits for a peer it cannot place
                p.kill()
                o, e = p.communicate()
                o += "init refused: timeout\n"
            outs.append((p.returncode, o, e))
    ids = [[l for l in o.splitlines() if l.startswith("id ")] for _, o, _ in outs]
    assert all(len(i) == 1 for i in ids) and ids[0] == ids[1], outs              # the token reached rank 1 intact
    if torch.cuda.device_count() >= 2:
        assert all(rc == 0 and "allreduce ok" in o for rc, o, _ in outs), outs
    else:
        assert all("allreduce ok" in o or "init refused" in o for _, o, _ in outs), outs


def test_c_host_linear():
    """A plain C program links libhoisdf_hip.so through include/hoisdf.h only (no Python, no torch types)."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = "/tmp/hoisdf_test_linear_host"
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", os.path.join(repo, "tests", "c", "test_linear_host.c"), "-I",
                    os.path.join(repo, "include"), "-L", os.path.join(repo, "hoisdf_amd"), "-lhoisdf_hip",
                    "-Wl,-rpath," + os.path.join(repo, "hoisdf_amd"), "-o", exe], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "c host ok" in out.stdout, out.stdout + out.stderr


def test_c_host_encoder_layer(tmp_path):
    """A plain C host runs one transformer encoder layer through the coarse entries hoisdf_encoder_layer_fwd / _bwd on values
    dumped from fixture g5 (the reference's TransformerEncoderLayer output + the stack's inter_norm), then checks the
    backward against a finite difference - tests/c/test_encoder_layer_host.c."""
    import os
    import struct
    import subprocess
    import numpy as np
    g = load_golden("g5_transformer")
    pre = "hand_transformer.encoder."
    names = ["layers.0.self_attn.in_proj_weight", "layers.0.self_attn.in_proj_bias", "layers.0.self_attn.out_proj.weight",
             "layers.0.self_attn.out_proj.bias", "layers.0.norm1.weight", "layers.0.norm1.bias", "layers.0.linear1.weight",
             "layers.0.linear1.bias", "layers.0.linear2.weight", "layers.0.linear2.bias", "layers.0.norm2.weight",
             "layers.0.norm2.bias", "inter_norm.weight", "inter_norm.bias"]
    E, F = 256, 1024
    shapes = [(3 * E, E), (3 * E,), (E, E), (E,), (E,), (E,), (F, E), (F,), (E, F), (E,), (E,), (E,), (E,), (E,)]
    src = g["src"].permute(1, 0, 2).contiguous()                       # fixture is (S, B, E); the entry takes (B, S, E)
    Bn, S, _ = src.shape
    path = str(tmp_path / "enc_layer.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<5i", Bn, S, E, F, 4))
        f.write(src.numpy().astype("<f4").tobytes())
        for n, sh in zip(names, shapes):
            f.write(T.det_param(pre + n, sh).numpy().astype("<f4").tobytes())
        f.write(g["enc_layer0"].permute(1, 0, 2).contiguous().numpy().astype("<f4").tobytes())
        f.write(g["inter"][0].permute(1, 0, 2).contiguous().numpy().astype("<f4").tobytes())
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = "/tmp/hoisdf_test_encoder_layer_host"
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", os.path.join(repo, "tests", "c", "test_encoder_layer_host.c"), "-I",
                    os.path.join(repo, "include"), "-L", os.path.join(repo, "hoisdf_amd"), "-lhoisdf_hip",
                    "-Wl,-rpath," + os.path.join(repo, "hoisdf_amd"), "-o", exe], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "c host encoder layer ok" in out.stdout, out.stdout + out.stderr


def test_graft_entry_smoke():
    """the driver's smoke(): tiny eval forward vs the oracle + one train forward/backward"""
    import __graft_entry__ as g
    g.smoke()


def test_trainer_checkpoint_roundtrip_with_fused_adamw(tmp_path):
    """reference-format snapshot (module.-prefixed network, optimizer, lr_scheduler) written by one Trainer resumes in
    another: same parameters, same AdamW moments / step, and the next step gives the same loss."""
    from hoisdf_amd.engine import Trainer, latest_snapshot
    c = Config()
    c.resnet_type = 18
    c.apply_setting("dexycb")
    c.num_samp_hand, c.num_samp_obj = 96, 32
    c.model_dir = str(tmp_path)
    dev = torch.device("cuda", 0)
    tr = Trainer(c, dev, batch_size=2, tune_encoder=False)      # no MIOpen search for these one-off shapes
    it = iter(tr.batch_generator)
    for _ in range(2):
        tr.model._py_random = random.Random(0)
        tr.train_step(*next(it), 0, 0.0)
    tr.save_model(0, 1)
    assert latest_snapshot(c.model_dir)[1:] == (0, 1)
    tr2 = Trainer(c, dev, batch_size=2, tune_encoder=False)
    assert tr2.load_model() == 1
    for (k, a), (_, b) in zip(tr.model.state_dict().items(), tr2.model.state_dict().items()):
        assert torch.equal(a, b), k
    p1 = [p for p in tr.model.parameters() if p.requires_grad and p in tr.optimizer.state][0]
    p2 = [p for p in tr2.model.parameters() if p.requires_grad][[id(q) for q in tr.model.parameters() if q.requires_grad].index(id(p1))]
    assert float(tr2.optimizer.state[p2]["step"]) == 2.0
    assert torch.equal(tr.optimizer.state[p1]["exp_avg"], tr2.optimizer.state[p2]["exp_avg"])
    batch = next(it)
    from hoisdf_amd import ops
    losses = []
    for t in (tr, tr2):
        t.model._py_random = random.Random(1)
        ops.manual_seed(123)
        torch.manual_seed(5)
        losses.append(float(t.train_step(*batch, 0, 0.0)[0]))
    assert abs(losses[0] - losses[1]) <= 1e-4 * abs(losses[0]), losses


def test_two_stream_step_with_reducer_matches_single_stream_autograd():
    """cfg.overlap_streams (object stack on a second HIP stream) + the bucketed GradReducer (packs on the main stream,
    waits on side-stream events) hand over the same gradients as a single-stream plain backward: global relative L2
    difference at float-atomic noise level (measured 5-7e-6; bound 1e-4)."""
    from hoisdf_amd.ddp import GradReducer, reducible_parameters
    from hoisdf_amd.model import get_model
    from hoisdf_amd import ops
    c = Config()
    c.resnet_type = 18
    c.apply_setting("dexycb")
    c.num_samp_hand, c.num_samp_obj = 384, 128
    torch.manual_seed(0)
    model = get_model("train", cfg=c).to(DEV).eval()                      # dropout off (the two modes draw seeds in a different order)
    batch = tuple(T.to_device(x, DEV) for x in T.synthetic_batch(4, 384, 128, seed=5))

    def run(two, reducer=None):
        c.overlap_streams = two
        if reducer is not None:
            reducer.zero_grad()
        else:
            model.zero_grad(set_to_none=True)
        model._py_random = random.Random(0)
        torch.manual_seed(3)                                              # the pre-point jitter draws from torch's RNG
        out = model(*batch, "train", 0, 0.1)
        total = sum(v.mean() for k, v in out.items() if "_out" not in k)
        total.backward()
        if reducer is not None:
            reducer.finish()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters()
                if p.grad is not None and not n.startswith(("backbone_net", "decoder_net"))}

    ref = run(False)
    red = GradReducer(reducible_parameters(model))
    for _ in range(2):
        got = run(True, red)
        num = sum(float((got[n] - ref[n]).double().pow(2).sum()) for n in ref)
        den = sum(float(ref[n].double().pow(2).sum()) for n in ref)
        assert set(got) == set(ref) and (num / den) ** 0.5 < 1e-4, (num / den) ** 0.5


def test_sdf_infer_with_counts_queued_ahead_equals_the_blocking_form():
    """hoisdf_sdf_infer_count_begin (queued before other work, waited for through an event) + hoisdf_sdf_infer (device scan of
    the offsets, no synchronisation) return what the one-shot form returns, and Model.forward's early request is consumed by
    hot_path (main/model.py:246-355)."""
    from hoisdf_amd import ops as O
    nh, no, bins, b = 384, 128, 64, 3
    model, c = build("dexycb", nh, no, bins)
    pyr, _ = nhwc_pyramid(T.synthetic_pyramid(b, seed=8))
    inputs, targets, meta = T.synthetic_batch(b, nh, no, seed=81)
    meta["bbox_hand"][1] = torch.tensor([60.0, 50, 200, 210])            # different survivor counts per sample
    m = T.to_device(meta, DEV)
    ref = model.sdf_infer(pyr, m["mano_root"], m["cam_intr"], m["bbox_hand"], 3.1, nh, "hand")
    h = model.infer_counts_begin(m)
    junk = torch.randn(2048, 2048, device=DEV) @ torch.randn(2048, 2048, device=DEV)       # work queued behind the request
    got = model.sdf_infer(pyr, m["mano_root"], m["cam_intr"], m["bbox_hand"], 3.1, nh, "hand", h["hand"])
    for a, r in zip(got[:3], ref[:3]):
        assert torch.equal(a, r)
    cl = h["hand"].wait()
    assert len(cl) == b and cl[1] != cl[0]
    # a handle made for other inputs is not used blindly
    other = O.sdf_infer_count_begin(m["obj_center_cam"], m["cam_intr"], m["bbox_obj"], 3.1, bins)
    got2 = model.sdf_infer(pyr, m["mano_root"], m["cam_intr"], m["bbox_hand"], 3.1, nh, "hand", other)
    assert torch.equal(got2[0], ref[0])
    # the whole eval step: early request in hot_path == no request
    di, dt = T.to_device(inputs, DEV), T.to_device(targets, DEV)
    with torch.no_grad():
        _, o1 = model.hot_path(pyr, di, dt, m, "eval")
        _, o2 = model.hot_path(pyr, di, dt, m, "eval", infer_counts=model.infer_counts_begin(m))
    # (not bit-wise: the few-tile decoder GEMMs of the forward use split-K float atomics)
    assert float((o1["hand_joints_out"] - o2["hand_joints_out"]).abs().max()) <= 1e-6 * float(o1["hand_joints_out"].abs().max())
    del junk
