"""Pins the CPU oracle (oracle/hoisdf_oracle.py) to golden vectors captured from the real
reference (tests/golden/make_golden.py).  CPU only.  Tolerances: the reference's own fp32
noise floor is ~1e-7; we allow 2e-5 abs on O(1) activations."""
import random

import numpy as np
import pytest
import torch

from conftest import load_golden
from hoisdf_amd import testing as T
from hoisdf_amd.nets import mano as MANO
from oracle import hoisdf_oracle as O

B, NH, NO = 2, 48, 16


def close(a, b, atol=2e-5, rtol=1e-5):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    d = (a - b).abs().nan_to_num(0.0).max().item()
    # NaN == NaN: the reference itself yields NaN for loss_joint_3d when no point is near a joint
    assert torch.allclose(a, b, atol=atol, rtol=rtol, equal_nan=True), f"max abs diff {d}"


@pytest.fixture(scope="module")
def P():
    return T.det_params(T.hot_path_param_shapes(992))


@pytest.fixture(scope="module")
def stage_inputs():
    pyr = T.synthetic_pyramid(B, big=False, seed=1)
    inputs, targets, meta = T.synthetic_batch(B, NH, NO, seed=11)
    return pyr, inputs, targets, meta


def test_g1_sdf_forward(P, stage_inputs):
    g = load_golden("g1_sdf_forward")
    pyr, inputs, _, meta = stage_inputs
    cfg = O.OracleCfg(num_samp_hand=NH, num_samp_obj=NO)
    sh, peh = O.sdf_forward(P, cfg, pyr, inputs["hand_sdf_points"], meta["mano_root"], meta["cam_intr"], 3.1, "hand")
    so, peo = O.sdf_forward(P, cfg, pyr, inputs["obj_sdf_points"], meta["obj_center_cam"], meta["cam_intr"], 3.1, "obj")
    sf, _ = O.sdf_forward(P, cfg, pyr, inputs["hand_sdf_points"] * 6.0, meta["mano_root"], meta["cam_intr"], 3.1, "hand")
    close(sh, g["sdf_hand"]); close(so, g["sdf_obj"]); close(peh, g["pe_hand"]); close(peo, g["pe_obj"])
    close(sf, g["sdf_far"])
    # fixture sanity: the clamp is exercised but not everywhere
    frac = (g["sdf_hand"].abs() >= 0.15).float().mean().item()
    assert 0.0 < frac < 0.9


def test_bilinear_explicit_matches_grid_sample(stage_inputs):
    pyr, inputs, _, meta = stage_inputs
    _, grid = O.project_points(inputs["hand_sdf_points"] * 4.0, meta["mano_root"], meta["cam_intr"], 3.1)
    for name, fmap in pyr.items():
        a = O.sample_pyramid({name: fmap}, grid, [name])
        b = O.bilinear_gather_explicit(fmap, grid)
        close(a, b, atol=1e-5)


def test_g2_sdf_decoder(P):
    g = load_golden("g2_sdf_decoder")
    y = O.sdf_decoder(g["x"], P, "hand_sdf_decoder")
    close(y, g["y"])
    close(O.weightnorm_weight(P, "hand_sdf_decoder.linh0")[0], g["w0_row0"], atol=1e-6)
    close(O.weightnorm_weight(P, "hand_sdf_decoder.linh1")[5], g["w1_row5"], atol=1e-6)


def test_g3_lattice():
    g = load_golden("g3_lattice16")
    assert torch.equal(O.dense_lattice(16), g["lattice"])          # bit exact
    p = load_golden("g3_lattice64_probe")
    L = O.dense_lattice(64)
    assert torch.equal(L[::4099], p["rows"])
    assert np.allclose(L.double().sum(0).numpy(), p["colsum"], rtol=0, atol=1e-6)
    assert [len(torch.unique(L[:, i])) for i in range(3)] == list(p["nuniq"])


@pytest.mark.parametrize("bins,kh,ko", [(16, 24, 8), (64, NH, NO)])
def test_g3_sdf_infer(P, stage_inputs, bins, kh, ko):
    g = load_golden(f"g3_sdf_infer_bins{bins}")
    pyr, _, _, meta = stage_inputs
    cfg = O.OracleCfg(num_samp_hand=kh, num_samp_obj=ko, bins_n=bins)
    ph, sh, peh = O.sdf_infer(P, cfg, pyr, meta["mano_root"], meta["cam_intr"], meta["bbox_hand"], 3.1, kh, "hand")
    po, so, peo = O.sdf_infer(P, cfg, pyr, meta["obj_center_cam"], meta["cam_intr"], meta["bbox_obj"], 3.1, ko, "obj")
    # same sort on the same machine -> identical order expected; compare as sets to be safe
    for a, b in ((ph, g["pts_hand"]), (po, g["pts_obj"])):
        for i in range(a.shape[0]):
            sa = {tuple(r) for r in a[i].numpy().round(6).tolist()}
            sb = {tuple(r) for r in b[i].numpy().round(6).tolist()}
            assert len(sa ^ sb) <= 2, len(sa ^ sb)
    close(sh.abs().sum(1), g["sdf_hand"].abs().sum(1), atol=1e-4)
    close(so.abs().sum(1), g["sdf_obj"].abs().sum(1), atol=1e-4)
    close(peh.sum(1), g["pe_hand"].sum(1), atol=2e-3)


def test_sdf_infer_too_few_survivors_raises(P, stage_inputs):
    pyr, _, _, meta = stage_inputs
    cfg = O.OracleCfg(bins_n=16)
    tiny = torch.tensor([100.0, 100, 101, 101]).repeat(B, 1)
    with pytest.raises(ValueError):
        O.sdf_infer(P, cfg, pyr, meta["mano_root"], meta["cam_intr"], tiny, 3.1, 8, "hand")


def test_g4_token_mlp(P, stage_inputs):
    g = load_golden("g4_token_mlp")
    pyr, inputs, _, meta = stage_inputs
    cfg = O.OracleCfg()
    fea, cam = O.token_mlp(P, cfg, pyr, inputs["hand_pre_points"], meta["mano_root"], meta["cam_intr"], 3.1)
    close(fea, g["fea"]); close(cam, g["cam"], atol=1e-6)


def test_g5_transformer(P):
    g = load_golden("g5_transformer")
    cfg = O.OracleCfg(num_samp_hand=NH, num_samp_obj=NO)
    src = g["src"]
    close(O.encoder_layer(src, P, "hand_transformer.encoder.layers.0", cfg, False), g["enc_layer0"])
    mem, inter = O.encoder(src, P, "hand_transformer.encoder", 6, cfg, False)
    close(mem, g["memory"], atol=5e-5); close(inter, g["inter"], atol=5e-5)
    tm, mm = O.mano_tgt_mask(), O.memory_mask(17, NH, NO)
    assert np.array_equal(tm.numpy(), g["tgt_mask"]) and np.array_equal(mm.numpy(), g["memory_mask"])
    hs = O.decoder(mem, P["mano_query_embed.weight"], P, "hand_transformer.decoder", 4, cfg, tm, mm, False)
    close(hs, g["hs"], atol=5e-5)
    omem, ointer = O.encoder(src, P, "obj_transformer.encoder", 3, cfg, False)
    close(omem, g["obj_memory"], atol=5e-5); close(ointer, g["obj_inter"], atol=5e-5)


def test_g5p_transformer_prenorm():
    """cfg.pre_norm = True (main/config.py:122): forward_pre of both layer types + encoder.norm, against the reference's own output"""
    g = load_golden("g5p_transformer_prenorm")
    P = T.det_params(T.hot_path_param_shapes(992, pre_norm=True))
    cfg = O.OracleCfg(num_samp_hand=NH, num_samp_obj=NO, pre_norm=True)
    src = g["src"]
    close(O.encoder_layer(src, P, "hand_transformer.encoder.layers.0", cfg, False), g["enc_layer0"], atol=5e-5)
    mem, inter = O.encoder(src, P, "hand_transformer.encoder", 6, cfg, False)
    close(mem, g["memory"], atol=1e-4); close(inter, g["inter"], atol=1e-4)
    tm, mm = O.mano_tgt_mask(), O.memory_mask(17, NH, NO)
    hs = O.decoder(mem, P["mano_query_embed.weight"], P, "hand_transformer.decoder", 4, cfg, tm, mm, False)
    close(hs, g["hs"], atol=1e-4)
    omem, ointer = O.encoder(src, P, "obj_transformer.encoder", 3, cfg, False)
    close(omem, g["obj_memory"], atol=1e-4); close(ointer, g["obj_inter"], atol=1e-4)
    # the switch changes the result (the fixture is not the post-norm one)
    assert float((g["memory"] - load_golden("g5_transformer")["memory"]).abs().max()) > 1e-2


def test_g2c_sdf_decoder_with_the_classifier_branch(stage_inputs):
    """cfg.ClassifierBranch = True (main/config.py:91): class logits from the last hidden layer, against the reference's own output"""
    g = load_golden("g2c_sdf_decoder_cls")
    P = T.det_params(T.hot_path_param_shapes(992, classifier=True))
    y, c = O.sdf_decoder(g["x"], P, "hand_sdf_decoder", classifier=True)
    close(y, g["y"]); close(c, g["cls"], atol=5e-5)
    pyr, inputs, _, meta = stage_inputs
    cfg = O.OracleCfg(num_samp_hand=NH, num_samp_obj=NO, ClassifierBranch=True)
    sh, peh, ch = O.sdf_forward(P, cfg, pyr, inputs["hand_sdf_points"], meta["mano_root"], meta["cam_intr"], 3.1, "hand")
    close(sh, g["sdf_hand"]); close(peh, g["pe_hand"]); close(ch, g["cls_hand"], atol=5e-5)
    # the SDF itself does not depend on the switch
    close(sh, load_golden("g1_sdf_forward")["sdf_hand"])


def test_option_state_dict_schemas_match_the_reference():
    import json, os
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g10_state_dict_options.json")))
    skip = ("mano_head.", "mano_layer.")
    for key, kw in (("pre_norm", dict(pre_norm=True)), ("classifier", dict(classifier=True))):
        ours = set(T.hot_path_param_shapes(992, **kw))
        theirs = {k for k in ref[key] if not k.startswith(skip)}
        assert ours == theirs, (key, sorted(ours ^ theirs)[:8])
    # the product modules register their parameters in the reference's ORDER (the fixture lists are unsorted since round 6:
    # optimizer state in a checkpoint is indexed by parameters() order - ClassifierBranch puts classifier_head behind linh4)
    from hoisdf_amd.config import Config
    from hoisdf_amd.model import get_model
    for key, attr in (("pre_norm", "pre_norm"), ("classifier", "ClassifierBranch")):
        c = Config()
        c.resnet_type = 18
        c.apply_setting("dexycb")
        setattr(c, attr, True)
        ours = [k for k in get_model("test", cfg=c).state_dict() if not k.startswith(("backbone_net", "decoder_net"))]
        assert ours == ref[key], next((i, a, b) for i, (a, b) in enumerate(zip(ours, ref[key])) if a != b)


def test_g6_vote(P):
    g = load_golden("g6_vote")
    inter = load_golden("g5_transformer")["inter"]
    off = O.mlp(inter[:, :NH], P, "linear_handvote", 4, False)
    cls = O.mlp(inter[:, :NH], P, "linear_handcls", 3, False)
    close(off, g["hand_off"], atol=5e-5); close(cls, g["hand_cls"], atol=5e-5)
    l1, l2, l3, joints = O.joint_vote(g["pts"], g["hand_off"], g["hand_cls"], g["joint_gt"], 0.04)
    close(joints, g["joints"], atol=1e-6)
    close(l1, g["loss_joint_3d"], rtol=1e-5, atol=1e-4)
    close(l2, g["loss_joint_cls"], rtol=1e-5)
    close(l3, g["loss_all_joint_3d"], rtol=1e-5, atol=1e-4)


def test_g9_mano_and_rotations():
    g = load_golden("g9_mano")
    layer = MANO.ManoLayer(MANO.synthetic_assets(0))
    v, j = layer(g["pose"], g["betas"])
    close(v, g["verts"], atol=2e-3)     # millimetres
    close(j, g["joints"], atol=2e-3)
    R = O.rot6d_to_mat(g["x6"])
    close(R, g["R"], atol=1e-6)
    close(O.mat_to_aa(R), g["aa"], atol=1e-5)
    close(O.rodrigues_via_quat(g["aa"]), g["R_back"], atol=1e-5)


E2E = [("dexycb", False, 48, 16, 16, 2), ("ho3d", True, 48, 16, 16, 2), ("ho3d_render", False, 48, 16, 16, 2),
       ("dexycb", False, 384, 128, 64, 1), ("ho3d_render", False, 384, 128, 64, 1), ("ho3d", True, 384, 128, 64, 1),
       # the sizes BASELINE.json's configs[1] / configs[3] name (configs[4]'s 6144+2048 fixture is checked on the GPU only)
       ("dexycb", False, 1536, 512, 64, 2), ("ho3d_render", False, 3072, 1024, 64, 1),
       # round 6: trained-like statistics (betas 2e-3 / 1e-2, x 100 outlier channels in the pyramid: make_golden.py smallbeta_goldens)
       ("dexycb", False, 1536, 512, 64, 2, "_smallbeta")]


def _smallbeta(Pm):
    for k, v in T.SMALL_BETA.items():
        Pm[k] = torch.full_like(Pm[k], v)


@pytest.mark.parametrize("case", E2E, ids=lambda c: "-".join(str(x) for x in c))
def test_g7_e2e(case):
    setting, big, nh, no, bins, b = case[:6]
    sfx = case[6] if len(case) > 6 else ""
    g = load_golden(f"g7_e2e_{setting}_n{nh + no}{sfx}")
    ik = setting == "ho3d_render"
    Pm = T.det_params(T.hot_path_param_shapes(3968 if big else 992, ik=ik))
    if sfx == "_smallbeta":
        _smallbeta(Pm)
    cfg = O.OracleCfg(num_samp_hand=nh, num_samp_obj=no, bins_n=bins, use_inverse_kinematics=ik,
                      dataset="ho3d" if "ho3d" in setting else "dexycb")
    pyr = T.synthetic_pyramid(b, big=big, seed=2, outliers=100.0 if sfx == "_smallbeta" else 1.0)
    inputs, targets, meta = T.synthetic_batch(b, nh, no, seed=21)
    if bins == 16:
        meta["bbox_hand"] = torch.tensor([0.0, 0, 256, 256]).repeat(b, 1)
        meta["bbox_obj"] = torch.tensor([0.0, 0, 256, 256]).repeat(b, 1)
    layer = MANO.ManoLayer(MANO.synthetic_assets(0))
    with torch.no_grad():
        out = O.hot_path_forward(Pm, cfg, pyr, inputs, targets, meta, "eval", mano_layer=layer,
                                 hands_mean=layer.th_hands_mean)
    checked = 0
    for k, ref in g.items():
        if k.endswith("_mean"):                        # big fixtures store the per-point outputs as means
            close(out[k[:-5]].mean(1), ref, atol=2e-5)
            checked += 1
            continue
        if k not in out:
            raise AssertionError(f"oracle misses key {k}")
        if k in ("obj_rot_out", "obj_trans_out"):      # per-point rows follow sort order: compare means
            close(out[k].mean(1), ref.mean(1), atol=2e-5)
        else:
            tol = 1e-4 if ("loss" in k or k in ("obj_rot", "obj_trans")) else 2e-5
            close(out[k], ref, atol=tol, rtol=2e-5)
        checked += 1
    assert checked >= 6


@pytest.mark.parametrize("setting,nh,no,suffix", [("dexycb", 48, 16, ""), ("ho3d_render", 48, 16, ""), ("ho3d", 48, 16, ""),
                                                  ("dexycb", 1536, 512, "_n2048"), ("dexycb", 48, 16, "_branchB"),
                                                  ("dexycb", 1536, 512, "_n2048_smallbeta"), ("dexycb", 1536, 512, "_n2048_trainedlike")])
def test_g8_train_fwd_bwd(setting, nh, no, suffix):
    """_branchB: epoch >= cfg.point_sampling_epoch and the draw p = 0.844 >= 0.4 -> the query points come from the
    dense-lattice sdf_infer (main/model.py:470-481), the rest of the step trains on them."""
    g = load_golden(f"g8_train_{setting}{suffix}")
    epoch_cnt = 10 ** 8 if suffix == "_branchB" else 0
    ik = setting == "ho3d_render"
    big = setting == "ho3d"                              # the big decoder: C = 3968 (main/config.py:96,101-108)
    b = 2
    Pm = T.det_params(T.hot_path_param_shapes(3968 if big else 992, ik=ik))
    small = suffix.endswith(("_smallbeta", "_trainedlike"))        # trained-like statistics (make_golden.py smallbeta_goldens)
    g64 = load_golden(f"g8_train_{setting}{suffix}_fp64") if small else None
    if small:
        _smallbeta(Pm)
    if suffix.endswith("_trainedlike"):
        T.apply_trained_like(lambda n_: Pm[n_])
    for v in Pm.values():
        v.requires_grad_(True)
    cfg = O.OracleCfg(num_samp_hand=nh, num_samp_obj=no, bins_n=16, use_inverse_kinematics=ik,
                      dataset="ho3d" if "ho3d" in setting else "dexycb", dropout=0.0, sdf_dropout=0.0)
    pyr = {k: v.requires_grad_(True) for k, v in T.synthetic_pyramid(b, big=big, seed=3, outliers=100.0 if small else 1.0).items()}
    inputs, targets, meta = T.synthetic_batch(b, nh, no, seed=31)
    layer = MANO.ManoLayer(MANO.synthetic_assets(0))
    random.seed(0)
    torch.manual_seed(1234)
    out = O.hot_path_forward(Pm, cfg, pyr, inputs, targets, meta, "train", epoch_cnt, 0.5, mano_layer=layer,
                             hands_mean=layer.th_hands_mean)
    losses = {k: v.mean() for k, v in out.items() if "_out" not in k}
    for k, v in losses.items():
        close(v, g["loss." + k], rtol=2e-5, atol=1e-5)
    total = sum(losses.values())
    close(total, g["total"], rtol=2e-5)
    total.backward()
    n = 0
    for name, p in Pm.items():
        key = "gradnorm." + name
        if key in g:
            assert p.grad is not None, name
            # branch B samples the points closest to the surface: d sigma / d beta is large there and the scalar beta
            # gradient is a heavily cancelling sum (observed 1.3e-3 between two fp32 summation orders)
            rt = 3e-3 if (suffix == "_branchB" and name.endswith("sigmoid_beta")) else 2e-4
            if small:
                # fp32 itself is ill-conditioned at trained-like statistics: the REFERENCE's fp32 norms sit up to 1.3e-3 (hand beta:
                # 11.7 %) from the fp64 values (tools/fp64_truth_smallbeta.py) - the restatement is held to the same distance
                t64, ref, got = float(g64[key]), float(g[key]), float(p.grad.double().norm())
                assert abs(got - t64) <= max(1e-3 * abs(t64), 1.5 * abs(ref - t64)) + 1e-6, (name, got, t64, ref)
            else:
                close(p.grad.double().norm().float(), g[key], rtol=rt, atol=1e-6)
            n += 1
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name   # unused params
    assert n > 100
    # element-wise, tolerance relative to the tensor's max; at 2048 points each pyramid-gradient element is an fp32 sum of
    # ~25 k terms whose order differs between the restatement's gather and ATen's (observed 3.2e-4 of the max)
    def gclose(a, b, rel=3e-4 if nh < 1000 else 1e-3):
        close(a, b, rtol=0, atol=rel * float(b.abs().max()) + 1e-9)

    if not small:
        gclose(Pm["hand_sigmoid_beta"].grad, g["grad.hand_sigmoid_beta"], **({"rel": 3e-3} if suffix == "_branchB" else {}))
    gclose(Pm["linear_handcls.layers.2.weight"].grad, g["grad.linear_handcls.layers.2.weight"])
    if small:                         # two-sided against the fp64 run, like the gradient norms above
        t64 = g64["grad.pyr.stride32"].double()
        mx = float(t64.abs().max())
        d_ref = float((g["grad.pyr.stride32"].double() - t64).abs().max())
        d_got = float((pyr["stride32"].grad[:, ::16].double() - t64).abs().max())
        assert d_got <= max(1.2e-3 * mx, 1.5 * d_ref), (d_got / mx, d_ref / mx)
        t2, r2 = float(g64["grad.pyr.stride2_norm"]), float(g["grad.pyr.stride2_norm"])
        assert abs(float(pyr["stride2"].grad.double().norm()) - t2) <= max(1e-3 * t2, 1.5 * abs(r2 - t2))
        return
    gclose(pyr["stride32"].grad[:, ::16], g["grad.pyr.stride32"])
    close(pyr["stride2"].grad.double().norm().float(), g["grad.pyr.stride2_norm"], rtol=1e-4)


def test_fp64_truth_fixture_documents_the_references_own_fp32_distance():
    """tests/golden/g8_train_dexycb_n2048_fp64.npz = the pinned oracle run in float64 (tools/fp64_truth.py).  The element-wise
    stride-32 pyramid gradient is ill-conditioned: the REFERENCE's fp32 value sits 1.0e-3 of the tensor's max away from fp64
    at its worst element, while its total loss agrees to 1e-8 - the yardstick the GPU tests use (conftest.pyramid_gradient_close)."""
    g = load_golden("g8_train_dexycb_n2048")
    g64 = load_golden("g8_train_dexycb_n2048_fp64")
    ref, tru = g["grad.pyr.stride32"].double(), g64["grad.pyr.stride32"].double()
    mx = float(ref.abs().max())
    d = float((ref - tru).abs().max()) / mx
    assert 5e-4 < d < 1.2e-3, d
    assert abs(float(g64["total"]) - float(g["total"])) <= 1e-7 * abs(float(g["total"]))


def test_g13_aux_image_losses():
    """(f4) the restated auxiliary image losses against the reference's own forward (main/model.py:128-143,404-422) on the
    seeded decoder output; every second pixel of the maps + exact means + the gradient w.r.t. the decoder output."""
    g = load_golden("g13_aux_losses")
    dec = T.synthetic_decoder_out(2, seed=13).requires_grad_(True)
    _, targets, _ = T.synthetic_batch(2, 48, 16, seed=31)
    from hoisdf_amd.config import Config
    out = O.aux_image_losses(dec, targets, Config().sigma)
    sub = lambda t: t[..., ::2, ::2]
    close(sub(out["heatmap"]), g["heatmap"], atol=1e-4, rtol=1e-6)          # peaks of 255 per joint
    close(sub(out["joint_heatmap"]), g["joint_heatmap"], atol=1e-2, rtol=1e-5)   # squared differences up to ~6e4
    close(sub(out["obj_seg"]), g["obj_seg"], atol=1e-6, rtol=1e-6)
    close(sub(out["hand_seg"]), g["hand_seg"], atol=1e-6, rtol=1e-6)
    for k in ("joint_heatmap", "obj_seg", "hand_seg"):
        assert abs(float(out[k].detach().double().mean()) - float(g["mean_" + k])) <= 1e-6 * abs(float(g["mean_" + k]))
    (out["joint_heatmap"].mean() + out["obj_seg"].mean() + out["hand_seg"].mean()).backward()
    close(sub(dec.grad), g["grad_decoder_out"], atol=1e-9, rtol=1e-5)
    assert abs(float(dec.grad.double().norm()) - float(g["grad_norm"])) <= 1e-6 * float(g["grad_norm"])


def test_g14_sampler_fixture_is_the_references_selection():
    """(f2) tests/golden/g14_sampler.npz was produced by executing data/dexycb.py:514-549,:288,:596-617 themselves; here: its
    row sets obey the reference's contract on the regenerated frames (regions, no repeats, the |sdf| < dist pre-filter) and
    the hand-off it stores equals a direct numpy evaluation of the same rows - so the GPU test can trust it as the target."""
    g = load_golden("g14_sampler")
    frames, index = T.synthetic_sdf_frames(4, seed=14)
    nh, no, dist, sc = 64, 48, 0.05, 3.1
    for mode in ("train", "test"):
        for i, (a, (n_h, n_o)) in enumerate(zip(frames, index)):
            k = f"{mode}{i}."
            idx = np.asarray(g[k + "all_idx"])
            h, o = idx[:nh], idx[nh:nh + no]
            assert len(set(h)) == nh and h.max() < n_h and len(set(o)) == no and o.min() >= n_h and o.max() < n_h + n_o
            d = a[idx].copy()
            if i % 2 == 1:
                d[:, 0] *= -1
            if mode == "train":
                eh, eo = np.asarray(g[k + "elig_hand"]), np.asarray(g[k + "elig_obj"])
                assert np.array_equal(eh, np.where(np.abs(a[:n_h, 3]) < dist)[0])
                assert np.array_equal(eo, np.where(np.abs(a[n_h:, 4]) < dist)[0] + n_h)
                assert set(idx[nh + no:2 * nh + no]) <= set(eh) and set(idx[2 * nh + no:]) <= set(eo)
                d[:, :3] = d[:, :3].dot(g[k + "rot_mat"].double().numpy().T)
            root, oc = g[k + "hand_root"].numpy(), g[k + "obj_center_cam"].numpy()
            hand = d[:nh, :5].copy(); hand[:, :3] -= root; hand *= sc
            np.testing.assert_allclose(g[k + "hand_sdf_points"].numpy(), hand, atol=1e-6)
            if mode == "train":
                np.testing.assert_allclose(g[k + "obj_pre_points"].numpy(), (d[2 * nh + no:, :3] - oc) * sc, atol=1e-6)
