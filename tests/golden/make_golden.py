#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by IMPORTING AND RUNNING the real
reference (/root/reference, read-only) on CPU.  Runs only in the build container; the
reference never travels - only the arrays written here do.

Usage:  python tests/golden/make_golden.py            (from the repo root)

Harness shims (none of them edits the reference):
  1. sys.path[0] = /root/reference, CWD = a scratch dir (config import mkdirs outputs/log);
  2. a stub ``torchvision.models.resnet`` exposing BasicBlock / Bottleneck / model_urls
     (torchvision is not installed; encoder-only dependency);
  3. ``Tensor.cuda`` / ``Module.cuda`` -> identity (the reference hard-codes .cuda());
  4. ``ManoLayer`` -> hoisdf_amd.nets.mano.ManoLayer over a synthetic MANO-shaped asset
     (the licensed MANO_RIGHT.pkl is absent); the LBS maths itself is pinned separately
     against manopth's own ``ManoLayer.forward`` (fixture g9).
Weights are a pure function of the parameter name (hoisdf_amd.testing.det_param) and are not
stored; inputs come from the seeded generators in hoisdf_amd.testing.
"""
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
from hoisdf_amd import testing as T                      # noqa: E402
from hoisdf_amd.nets import encoder as ENC               # noqa: E402
from hoisdf_amd.nets import mano as MANO                 # noqa: E402

REF = "/root/reference"


def install_shims():
    os.chdir(tempfile.mkdtemp(prefix="hoisdf_golden_"))
    sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")
    tvr.BasicBlock, tvr.Bottleneck, tvr.model_urls = ENC.BasicBlock, ENC.Bottleneck, {}
    tv.models, tvm.resnet = tvm, tvr
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm,
                        "torchvision.models.resnet": tvr})
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32) if torch.is_tensor(t) else np.asarray(t)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np32(v) if torch.is_tensor(v) else np.asarray(v)
                                 for k, v in arrs.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)")


class _Backbone(torch.nn.Module):
    def forward(self, img):
        return None, None


class _Decoder(torch.nn.Module):
    def __init__(self, pyr, decoder_out=None):
        super().__init__()
        self.pyr = pyr
        self.decoder_out = decoder_out

    def forward(self, a, b):
        B = next(iter(self.pyr.values())).shape[0]
        return self.pyr, (self.decoder_out if self.decoder_out is not None else torch.full((B, 3, 128, 128), 0.5))


def build_reference(setting: str, n_hand: int, n_obj: int, bins_n: int, resnet_type=18):
    """setting in {dexycb, ho3d, ho3d_render} -> reference Model with deterministic weights."""
    from main.config import cfg
    import main.model as M
    cfg.setting = setting
    cfg.dataset = "ho3d" if "ho3d" in setting else "dexycb"
    cfg.use_big_decoder = setting == "ho3d"
    cfg.use_inverse_kinematics = setting == "ho3d_render"
    cfg.resnet_type = 50 if cfg.use_big_decoder else resnet_type
    cfg.num_samp_hand, cfg.num_samp_obj, cfg.bins_n = n_hand, n_obj, bins_n
    cfg.calc_mutliscale_dim(cfg.use_big_decoder, cfg.resnet_type)
    M.ManoLayer = lambda **k: MANO.ManoLayer(MANO.synthetic_assets(0))
    torch.manual_seed(0)
    model = M.get_model("test")
    with torch.no_grad():
        for name, p in model.named_parameters():
            if not name.startswith(("backbone_net", "decoder_net")):
                p.copy_(T.det_param(name, p.shape))
    return model, cfg


def ocfg_dict(cfg):
    return dict(num_samp_hand=cfg.num_samp_hand, num_samp_obj=cfg.num_samp_obj,
                bins_n=cfg.bins_n, use_inverse_kinematics=bool(cfg.use_inverse_kinematics),
                dataset=cfg.dataset)


def stage_goldens():
    """g1..g6: stage-wise fixtures (small decoder, C=992)."""
    B, nh, no = 2, 48, 16
    model, cfg = build_reference("dexycb", nh, no, 16)
    model.eval()
    pyr = T.synthetic_pyramid(B, big=False, seed=1)
    inputs, targets, meta = T.synthetic_batch(B, nh, no, seed=11)
    root, oc, K = meta["mano_root"], meta["obj_center_cam"], meta["cam_intr"]
    with torch.no_grad():
        # g1 sdf_forward, both decoders (main/model.py:181-244)
        sh, _, peh = model.sdf_forward(pyr, inputs["hand_sdf_points"], root, K, cfg.hand_sdf_scale, "hand")
        so, _, peo = model.sdf_forward(pyr, inputs["obj_sdf_points"], oc, K, cfg.obj_sdf_scale, "obj")
        # a point that projects outside the image (exercises the border clamp)
        far = inputs["hand_sdf_points"].clone() * 6.0
        sf, _, pef = model.sdf_forward(pyr, far, root, K, cfg.hand_sdf_scale, "hand")
        save("g1_sdf_forward", sdf_hand=sh, pe_hand=peh, sdf_obj=so, pe_obj=peo, sdf_far=sf)

        # g2 SDFDecoder alone + explicit effective weights (common/nets/sdf_net.py:87-122)
        x = torch.from_numpy(np.random.default_rng(5).standard_normal((96, 289)).astype(np.float32))
        y, _ = model.hand_sdf_decoder(x)
        save("g2_sdf_decoder", x=x, y=y,
             w0_row0=model.hand_sdf_decoder.linh0.weight[0], w1_row5=model.hand_sdf_decoder.linh1.weight[5])

        # g4 get_input_transformer (main/model.py:145-179)
        fea, cam = model.get_input_transformer(pyr, inputs["hand_pre_points"], root, K, cfg.hand_sdf_scale)
        save("g4_token_mlp", fea=fea, cam=cam)

        # g5 transformers (common/nets/transformer.py)
        r = np.random.default_rng(7)
        S = nh + no
        src = torch.from_numpy(r.standard_normal((S, B, 256)).astype(np.float32))
        from common.utils.misc import get_mano_tgt_mask, get_mano_memory_mask
        l0 = model.hand_transformer.encoder.layers[0](src)
        hs, mem, inter, _ = model.hand_transformer(
            src=src, mask=None, pos_embed=torch.zeros_like(src), src_mask=None,
            query_embed=model.mano_query_embed.weight, tgt_mask=get_mano_tgt_mask(),
            memory_mask=get_mano_memory_mask())
        omem, ointer = model.obj_transformer(src=src, mask=None, pos_embed=torch.zeros_like(src),
                                             src_mask=None)
        save("g5_transformer", src=src, enc_layer0=l0, hs=hs, memory=mem, inter=inter,
             obj_memory=omem, obj_inter=ointer, tgt_mask=get_mano_tgt_mask().numpy(),
             memory_mask=get_mano_memory_mask().numpy())

        # g6 heads + JointvoteLoss (main/model.py:587-593, common/nets/loss.py:23-61)
        enc = inter[:, :nh]
        off = model.linear_handvote(enc)
        cls = model.linear_handcls(enc)
        pts = inputs["hand_pre_points"] / 3.1
        gt = pts[:, :20] * 1000 + torch.from_numpy(r.standard_normal((B, 20, 3)).astype(np.float32)) * 10
        l1, l2, l3, joints = model.joints_vote_loss(pts, off, cls, gt)
        save("g6_vote", hand_off=off, hand_cls=cls, pts=pts, joint_gt=gt, loss_joint_3d=l1,
             loss_joint_cls=l2, loss_all_joint_3d=l3, joints=joints)

    # g3 lattice + sdf_infer (main/model.py:246-355), bins 16 and 64
    with torch.no_grad():
        for bins in (16, 64):
            cfg.bins_n = bins
            k_h, k_o = (24, 8) if bins == 16 else (nh, no)
            ph, sh, peh, _ = model.sdf_infer(pyr, root, K, meta["bbox_hand"], cfg.hand_sdf_scale, k_h, "hand")
            po, so, peo, _ = model.sdf_infer(pyr, oc, K, meta["bbox_obj"], cfg.obj_sdf_scale, k_o, "obj")
            save(f"g3_sdf_infer_bins{bins}", pts_hand=ph, sdf_hand=sh, pe_hand=peh, pts_obj=po,
                 sdf_obj=so, pe_obj=peo)
        # the lattice itself: replicate the reference's own lines by running them (model.py:257-273)
        for bins in (16, 64):
            n = bins
            idx = torch.arange(0, n ** 3, 1, out=torch.LongTensor())
            s = torch.zeros(n ** 3, 3)
            s[:, 2] = idx % n
            s[:, 1] = (idx.long() / n) % n
            s[:, 0] = ((idx.long() / n) / n) % n
            v = 2.0 / (n - 1)
            s[:, 0] = (s[:, 0] * v) + -1
            s[:, 1] = (s[:, 1] * v) + -1
            s[:, 2] = (s[:, 2] * v) + -1
            if bins == 16:
                save("g3_lattice16", lattice=s)
            else:
                save("g3_lattice64_probe", rows=s[::4099], colsum=s.double().sum(0).numpy(),
                     nuniq=np.array([len(torch.unique(s[:, i])) for i in range(3)]))


def e2e_goldens():
    """g7: full Model.forward in eval mode, encoder bypassed by a synthetic pyramid."""
    for setting, big in (("dexycb", False), ("ho3d", True), ("ho3d_render", False)):
        for (nh, no, bins, B) in ((48, 16, 16, 2), (384, 128, 64, 1)):
            if big and nh > 48:
                nh, no = 384, 128       # round 4: the big decoder (C = 3968) through the 64^3 lattice as well
            model, cfg = build_reference(setting, nh, no, bins)
            model.eval()
            pyr = T.synthetic_pyramid(B, big=big, seed=2)
            model.backbone_net, model.decoder_net = _Backbone(), _Decoder(pyr)
            inputs, targets, meta = T.synthetic_batch(B, nh, no, seed=21)
            if bins == 16:      # the 16^3 lattice is coarse: widen the boxes so enough survive
                meta["bbox_hand"] = torch.tensor([0.0, 0, 256, 256]).repeat(B, 1)
                meta["bbox_obj"] = torch.tensor([0.0, 0, 256, 256]).repeat(B, 1)
            with torch.no_grad():
                out = model(inputs, targets, meta, "eval")
            keep = {k: v for k, v in out.items() if torch.is_tensor(v) and v.numel() < 200000
                    and k not in ("hand_seg_gt_out", "obj_seg_gt_out", "hand_seg_pred_out",
                                  "obj_seg_pred_out", "joint_heatmap_out", "joint_heatmap",
                                  "obj_seg", "hand_seg")}
            save(f"g7_e2e_{setting}_n{nh + no}", **keep)


def train_goldens(sizes=((2, 48, 16, ""),), settings=("dexycb", "ho3d_render", "ho3d")):
    """g8: train-mode forward + backward with every dropout p forced to 0 (branch A)."""
    for (B, nh, no, suffix) in sizes:
        _train_goldens(B, nh, no, suffix, settings)


def train_branch_b_golden():
    """g8 ..._branchB: the training step after cfg.point_sampling_epoch when the draw p >= 0.4 (random.seed(0): 0.844):
    query points come from the dense-lattice sdf_infer under no_grad (main/model.py:470-481), everything after it trains."""
    _train_goldens(2, 48, 16, "_branchB", ("dexycb",), epoch_cnt=10 ** 8)


def _train_goldens(B, nh, no, suffix, settings, epoch_cnt=0, prepare=None, pyramid_kw=None):
    for setting in settings:
        model, cfg = build_reference(setting, nh, no, 16)
        if prepare:
            prepare(model)
        model.train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, torch.nn.MultiheadAttention):
                m.dropout = 0.0
            if hasattr(m, "dropout_prob"):
                m.dropout_prob = 0.0
        pyr = T.synthetic_pyramid(B, big=setting == "ho3d", seed=3, **(pyramid_kw or {}))      # "ho3d" = the big decoder: C = 3968
        pyr = {k: v.clone().requires_grad_(True) for k, v in pyr.items()}
        model.backbone_net, model.decoder_net = _Backbone(), _Decoder(pyr)
        inputs, targets, meta = T.synthetic_batch(B, nh, no, seed=31)
        targets["joint_coord"] = targets["joint_coord"]
        random.seed(0)
        torch.manual_seed(1234)
        out = model(inputs, targets, meta, "train", epoch_cnt, 0.5)
        skip = ("joint_heatmap", "obj_seg", "hand_seg")
        losses = {k: v.mean() for k, v in out.items() if "_out" not in k and k not in skip}
        total = sum(losses.values())
        total.backward()
        gn = {}
        for name, p in model.named_parameters():
            if p.grad is not None:
                gn["gradnorm." + name] = p.grad.double().norm().float()
        sel = {
            "grad.hand_sigmoid_beta": model.hand_sigmoid_beta.grad,
            "grad.obj_sigmoid_beta": model.obj_sigmoid_beta.grad,
            "grad.linear_sdfin.layers.1.bias": model.linear_sdfin.layers[1].bias.grad,
            "grad.hand_sdf_decoder.linh0.weight_g": model.hand_sdf_decoder.linh0.weight_g.grad,
            "grad.hand_transformer.encoder.layers.0.self_attn.in_proj_bias":
                model.hand_transformer.encoder.layers[0].self_attn.in_proj_bias.grad,
            "grad.linear_handcls.layers.2.weight": model.linear_handcls.layers[2].weight.grad,
            "grad.pyr.stride32": pyr["stride32"].grad[:, ::16],
            "grad.pyr.stride2_norm": pyr["stride2"].grad.double().norm().float(),
        }
        save(f"g8_train_{setting}{suffix}", total=total, **{"loss." + k: v for k, v in losses.items()},
             **gn, **sel)


def big_goldens():
    """Fixtures at the sizes BASELINE.json's configs name: the reference itself runs these on CPU
    in seconds to minutes.  g7 eval: configs[1] points (1536+512, dexycb, B=2), configs[3] (3072+1024, ho3d_render =
    the IK variant, B=1), configs[4] (6144+2048, dexycb, B=1), all through sdf_infer on the 64^3 lattice;
    g8 train fwd+bwd at 1536+512, B=2 (losses + per-parameter gradient norms + a few gradient slices)."""
    for setting, nh, no, B in (("dexycb", 1536, 512, 2), ("ho3d_render", 3072, 1024, 1), ("dexycb", 6144, 2048, 1)):
        model, cfg = build_reference(setting, nh, no, 64)
        model.eval()
        pyr = T.synthetic_pyramid(B, big=False, seed=2)
        model.backbone_net, model.decoder_net = _Backbone(), _Decoder(pyr)
        inputs, targets, meta = T.synthetic_batch(B, nh, no, seed=21)
        with torch.no_grad():
            out = model(inputs, targets, meta, "eval")
        keep = {k: v for k, v in out.items() if torch.is_tensor(v) and v.numel() < 200000
                and k not in ("hand_seg_gt_out", "obj_seg_gt_out", "hand_seg_pred_out",
                              "obj_seg_pred_out", "joint_heatmap_out", "joint_heatmap",
                              "obj_seg", "hand_seg")}
        for k in ("obj_rot_out", "obj_trans_out"):      # per-point rows follow the |sdf| order: keep the means only
            if k in keep:
                keep[k + "_mean"] = keep.pop(k).mean(1)
        save(f"g7_e2e_{setting}_n{nh + no}", **keep)
    train_goldens(sizes=((2, 1536, 512, "_n2048"),), settings=("dexycb",))


def smallbeta_goldens():
    """Round 6: trained-like statistics at BASELINE configs[1]'s points.  Every other fixture runs det_param's
    beta = 0.08 / 0.12 (sigma <= 12.5); a trained model's betas sit near the 2e-3 floor (main/model.py:123-126): here hand 2e-3 (the
    floor itself: sigma = 500 outside the surface, ~1e-30 inside) and obj 1e-2, on a pyramid whose channels 3 / 17 / 40 are x 100
    louder than the rest (testing.synthetic_pyramid(outliers=100)).  Token rows then span ~30 decades inside one matrix and the
    pyramid-fed MLPs see outlier columns - what the f16x2 form's shared scales had never been shown.
    g7_e2e_dexycb_n2048_smallbeta: eval through the 64^3-lattice sdf_infer (B = 2); g8_train_dexycb_n2048_smallbeta: train fwd + bwd."""
    nh, no, B = 1536, 512, 2

    def set_beta(model):
        with torch.no_grad():
            for k, v in T.SMALL_BETA.items():
                getattr(model, k).fill_(v)

    model, cfg = build_reference("dexycb", nh, no, 64)
    set_beta(model)
    model.eval()
    pyr = T.synthetic_pyramid(B, big=False, seed=2, outliers=100.0)
    model.backbone_net, model.decoder_net = _Backbone(), _Decoder(pyr)
    inputs, targets, meta = T.synthetic_batch(B, nh, no, seed=21)
    with torch.no_grad():
        out = model(inputs, targets, meta, "eval")
    keep = {k: v for k, v in out.items() if torch.is_tensor(v) and v.numel() < 200000
            and k not in ("hand_seg_gt_out", "obj_seg_gt_out", "hand_seg_pred_out", "obj_seg_pred_out", "joint_heatmap_out",
                          "joint_heatmap", "obj_seg", "hand_seg")}
    for k in ("obj_rot_out", "obj_trans_out"):
        if k in keep:
            keep[k + "_mean"] = keep.pop(k).mean(1)
    save("g7_e2e_dexycb_n2048_smallbeta", **keep)
    _train_goldens(B, nh, no, "_n2048_smallbeta", ("dexycb",), prepare=set_beta, pyramid_kw=dict(outliers=100.0))

    # "_trainedlike": the same + the first encoder layers' q / k projections at the scale a trained network would have them for such
    # tokens (testing.TRAINED_LIKE_QK: scores of O(10-100) instead of 7e6 / 2.4e8, where fp32 softmax is decided by rounding)
    def trained_like(model):
        set_beta(model)
        sd = dict(model.named_parameters())
        T.apply_trained_like(lambda n: sd[n])
    _train_goldens(B, nh, no, "_n2048_trainedlike", ("dexycb",), prepare=trained_like, pyramid_kw=dict(outliers=100.0))


def aux_loss_golden():
    """g13 (f4): the encoder-side auxiliary image losses as the reference's own forward computes them
    (main/model.py:128-143 render_gaussian_heatmap, :404-422 MSE + 2 x BCELoss(reduction="none")) on a seeded decoder output
    (hoisdf_amd.testing.synthetic_decoder_out), plus the gradient of the sum of their means w.r.t. the decoder output.
    Stored: every second pixel of the maps (the test regenerates the inputs from the seeds) and the exact means."""
    B, nh, no = 2, 48, 16
    model, cfg = build_reference("dexycb", nh, no, 16)
    model.train()
    pyr = T.synthetic_pyramid(B, big=False, seed=3)
    dec = T.synthetic_decoder_out(B, seed=13).requires_grad_(True)
    model.backbone_net, model.decoder_net = _Backbone(), _Decoder(pyr, dec)
    inputs, targets, meta = T.synthetic_batch(B, nh, no, seed=31)
    random.seed(0)
    torch.manual_seed(1234)
    out = model(inputs, targets, meta, "train", 0, 0.5)
    hm = model.render_gaussian_heatmap(targets["joint_coord"])
    tot = out["joint_heatmap"].mean() + out["obj_seg"].mean() + out["hand_seg"].mean()
    g, = torch.autograd.grad(tot, dec)
    sub = lambda t: t.detach()[..., ::2, ::2]
    out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
    save("g13_aux_losses", heatmap=sub(hm), joint_heatmap=sub(out["joint_heatmap"]), obj_seg=sub(out["obj_seg"]),
         hand_seg=sub(out["hand_seg"]), mean_joint_heatmap=out["joint_heatmap"].double().mean().numpy(),
         mean_obj_seg=out["obj_seg"].double().mean().numpy(), mean_hand_seg=out["hand_seg"].double().mean().numpy(),
         mean_heatmap=hm.double().mean().numpy(), grad_decoder_out=sub(g), grad_norm=g.double().norm().numpy())


def sampler_golden():
    """g14 (f2): the SDF sample selection and hand-off of the reference's dataset object, produced by EXECUTING the reference's
    own source lines (data/dexycb.py:514-549 draw + flip, :288 in-plane rotation, :596-617 centre + scale) - read from
    /root/reference at generation time, dedented and exec'ed on a synthetic ``sdf_processed`` frame set
    (hoisdf_amd.testing.synthetic_sdf_frames) with ``np.random.seed`` fixed.  __getitem__ itself cannot run here (images,
    annotations, cv2); the lines between the blocks (image crop / augmentation) do not touch the points except for the
    rotation line, which is executed too.  Pinned: the eligibility sets of the |sdf| < dist pre-filter, numpy's own draws
    for this seed (the device sampler draws from another stream - it is held to the sets and counts), and the hand-off of
    those drawn rows."""
    import textwrap
    src = open(os.path.join(REF, "data", "dexycb.py")).read().split("\n")
    block = lambda a, b: textwrap.dedent("\n".join(src[a - 1:b]))
    draw, flip, rotate, handoff = block(514, 546), block(548, 549), block(288, 288), block(596, 617)
    assert "np.random.choice" in draw and "sdf_points[:, 0] *= -1" in flip and "rot_mat.T" in rotate and "hand_pre_points" in handoff
    frames, index = T.synthetic_sdf_frames(4, seed=14)
    tmp = tempfile.mkdtemp(prefix="hoisdf_sdf_")
    paths = []
    for i, a in enumerate(frames):
        paths.append(os.path.join(tmp, f"{i}.npy"))
        np.save(paths[-1], a)
    nh, no = 64, 48
    out = {}
    r = np.random.default_rng(141)
    for mode in ("train", "test"):
        for idx in range(len(frames)):
            do_flip = idx % 2 == 1
            self = types.SimpleNamespace(sdf_path_list=paths, sdf_index_list=[np.array(v) for v in index], num_samp_hand=nh,
                                         num_samp_obj=no, mode=mode, dist=0.05, hand_sdf_scale=3.1, obj_sdf_scale=3.1)
            ns = {"np": np, "self": self, "idx": idx, "do_flip": do_flip}
            np.random.seed(1000 + idx)
            exec(draw, ns)
            exec(flip, ns)
            th = 0.3 * (idx - 1.5)
            ns["rot_mat"] = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
            if mode == "train":
                exec(rotate, ns)
            ns["hand_root"] = (np.array([0.0, 0.0, 0.7]) + 0.01 * r.standard_normal(3)).astype(np.float32)
            ns["obj_center_cam"] = (np.array([0.03, 0.02, 0.72]) + 0.01 * r.standard_normal(3)).astype(np.float32)
            exec(handoff, ns)
            k = f"{mode}{idx}."
            sd, si = frames[idx], index[idx]
            out[k + "all_idx"] = ns["all_idx"].astype(np.int64)
            out[k + "hand_root"], out[k + "obj_center_cam"], out[k + "rot_mat"] = ns["hand_root"], ns["obj_center_cam"], ns["rot_mat"]
            out[k + "hand_sdf_points"], out[k + "obj_sdf_points"] = ns["hand_sdf_points"], ns["obj_sdf_points"]   # (n, 5) rows, scaled
            out[k + "sdf_raw_label"] = ns["sdf_raw_label"]
            if mode == "train":
                out[k + "hand_pre_points"], out[k + "obj_pre_points"] = ns["hand_pre_points"], ns["obj_pre_points"]
                # the eligibility sets exactly as the reference forms them (:530, :535)
                out[k + "elig_hand"] = np.where(np.abs(sd[: si[0], 3]) < self.dist)[0].astype(np.int64)
                out[k + "elig_obj"] = (np.where(np.abs(sd[si[0]:, 4]) < self.dist)[0] + si[0]).astype(np.int64)
    save("g14_sampler", **out)


def option_goldens():
    """The two reference switches no released configuration turns on (round-4 verdict item 6):
      g5p  cfg.pre_norm = True          common/nets/transformer.py:304-331 (encoder forward_pre), :397-437 (decoder forward_pre),
                                        encoder.norm on the stack output (:82-84, :199-200); main/config.py:122
      g2c  cfg.ClassifierBranch = True  common/nets/sdf_net.py:73-75,93-94,119-122 (classifier_head on the last hidden layer),
                                        main/model.py:236-240 (sdf_forward reshapes the logits); main/config.py:91
    Same inputs as g5 / g2 / g1; the state-dict keys of both variants are stored for the schema test."""
    from main.config import cfg
    B, nh, no = 2, 48, 16
    pyr = T.synthetic_pyramid(B, big=False, seed=1)
    inputs, targets, meta = T.synthetic_batch(B, nh, no, seed=11)
    root, K = meta["mano_root"], meta["cam_intr"]
    try:
        cfg.pre_norm = True
        model, _ = build_reference("dexycb", nh, no, 16)
        model.eval()
        keys_pre = [k for k in model.state_dict() if not k.startswith(("backbone_net", "decoder_net"))]      # registration order
        with torch.no_grad():
            r = np.random.default_rng(7)
            S = nh + no
            src = torch.from_numpy(r.standard_normal((S, B, 256)).astype(np.float32))
            from common.utils.misc import get_mano_tgt_mask, get_mano_memory_mask
            l0 = model.hand_transformer.encoder.layers[0](src)
            hs, mem, inter, _ = model.hand_transformer(
                src=src, mask=None, pos_embed=torch.zeros_like(src), src_mask=None,
                query_embed=model.mano_query_embed.weight, tgt_mask=get_mano_tgt_mask(),
                memory_mask=get_mano_memory_mask())
            omem, ointer = model.obj_transformer(src=src, mask=None, pos_embed=torch.zeros_like(src), src_mask=None)
            save("g5p_transformer_prenorm", src=src, enc_layer0=l0, hs=hs, memory=mem, inter=inter, obj_memory=omem, obj_inter=ointer)
    finally:
        cfg.pre_norm = False
    try:
        cfg.ClassifierBranch = True
        model, _ = build_reference("dexycb", nh, no, 16)
        model.eval()
        keys_cls = [k for k in model.state_dict() if not k.startswith(("backbone_net", "decoder_net"))]     # registration order
        with torch.no_grad():
            x = torch.from_numpy(np.random.default_rng(5).standard_normal((96, 289)).astype(np.float32))
            y, c = model.hand_sdf_decoder(x)
            sh, ch, peh = model.sdf_forward(pyr, inputs["hand_sdf_points"], root, K, cfg.hand_sdf_scale, "hand")
            save("g2c_sdf_decoder_cls", x=x, y=y, cls=c, sdf_hand=sh, cls_hand=ch, pe_hand=peh)
    finally:
        cfg.ClassifierBranch = False
    import json
    with open(os.path.join(OUT, "g10_state_dict_options.json"), "w") as f:
        json.dump({"pre_norm": keys_pre, "classifier": keys_cls}, f, indent=0)
    print("  wrote g10_state_dict_options.json")


def mano_golden():
    """g9: hoisdf_amd.nets.mano.ManoLayer vs manopth's own ManoLayer.forward
    (manopth/manopth/manolayer.py:111-276) on the same synthetic asset."""
    from manopth.manopth.manolayer import ManoLayer as RefMano
    assets = MANO.synthetic_assets(0)
    ref = RefMano.__new__(RefMano)
    torch.nn.Module.__init__(ref)
    ref.center_idx, ref.rot, ref.ncomps, ref.use_pca = 0, 3, 45, False
    ref.joint_rot_mode = ref.root_rot_mode = "axisang"
    ref.side, ref.robust_rot, ref.flat_hand_mean = "right", False, True
    for k, v in assets.items():
        ref.register_buffer(k, v.clone())
    r = np.random.default_rng(9)
    pose = torch.from_numpy((0.4 * r.standard_normal((5, 48))).astype(np.float32))
    betas = torch.from_numpy(r.standard_normal((5, 10)).astype(np.float32))
    with torch.no_grad():
        v, j = ref(th_pose_coeffs=pose, th_betas=betas)
    # also the 6D -> axis-angle chain of the reference mano head (common/nets/mano_head.py)
    from common.nets.mano_head import rot6d2mat, mat2aa, batch_rodrigues
    x6 = torch.from_numpy(r.standard_normal((64, 6)).astype(np.float32))
    Rm = rot6d2mat(x6)
    aa = mat2aa(Rm.clone())
    Rg = batch_rodrigues(aa)
    save("g9_mano", pose=pose, betas=betas, verts=v, joints=j, x6=x6, R=Rm, aa=aa, R_back=Rg)


def _stub_module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def metrics_golden():
    """g11: the reference's evaluation metrics (common/metrics.py:62-232, common/eval_util.py:11-103) on seeded inputs.
    Harness stubs: ``cv2`` (metrics.py imports it at module level; only the unused per-sample ``eval_batch_obj_direct``
    calls cv2.Rodrigues) and ``open3d`` (eval_util.py's calculate_fscore uses its nearest-neighbour distances; the
    fixture's F-scores are therefore computed here with brute-force numpy nearest neighbours over the SAME definition,
    eval_util.py:117-136, and labelled as such)."""
    _stub_module("cv2")
    _stub_module("open3d")
    from common.metrics import eval_batched_obj_direct, eval_hand_joint, rigid_align
    from common.eval_util import EvalUtil
    r = np.random.default_rng(11)
    f32 = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    B, V = 6, 300
    pred_j = f32(0.05 * r.standard_normal((B, 21, 3)))
    gt_j = pred_j * 1.1 + f32(0.004 * r.standard_normal((B, 21, 3))) + 0.01
    mje, pamje = eval_hand_joint(pred_j, gt_j)
    aligned = np.stack([rigid_align(pred_j[i].numpy(), gt_j[i].numpy()) for i in range(B)])
    templates = [{"verts": f32(0.05 * r.standard_normal((V, 3)))} for _ in range(4)]
    out = {"obj_rot": f32(0.3 * r.standard_normal((B, 40, 3))), "obj_trans": f32(0.05 * r.standard_normal((B, 40, 3)))}
    targets = {"obj_rot": f32(0.3 * r.standard_normal((B, 3))), "rel_obj_trans": f32(0.05 * r.standard_normal((B, 3)))}
    obj_cls = torch.tensor([1, 2, 3, 4, 1, 2])
    meta = {"cam_intr": torch.eye(3).repeat(B, 1, 1), "obj_cls": obj_cls}
    adds, mce, oce, _, n = eval_batched_obj_direct(out, targets, meta, templates, None, None)
    names = {0: "a", 1: "b", 2: "c", 3: "d"}
    meta_h = {"cam_intr": meta["cam_intr"], "obj_cls": [names[int(c) - 1] for c in obj_cls]}
    adds_h, _, _, mme_h, n_h = eval_batched_obj_direct(out, targets, meta_h, templates, None, names)
    ev = EvalUtil(num_kp=V)
    gt_v = f32(0.05 * r.standard_normal((B, V, 3)))
    pr_v = gt_v + f32(0.006 * r.standard_normal((B, V, 3)))
    for i in range(B):
        ev.feed(gt_v[i].numpy(), np.ones(V), pr_v[i].numpy())
    m3d, med, auc, pck, th = ev.get_measures(0.0, 0.05, 100)

    def fscore_np(gt, pr, t):           # eval_util.py:117-136 with brute-force nearest neighbours
        d = np.sqrt(((gt[:, None].astype(np.float64) - pr[None].astype(np.float64)) ** 2).sum(-1))
        d1, d2 = d.min(1), d.min(0)
        rec, prec = (d2 < t).mean(), (d1 < t).mean()
        return 2 * rec * prec / (rec + prec) if rec + prec > 0 else 0.0

    fs = np.array([[fscore_np(gt_v[i].numpy(), pr_v[i].numpy(), t) for t in (0.005, 0.015)] for i in range(B)])
    save("g11_metrics", pred_j=pred_j, gt_j=gt_j, mje=np.float64(mje), pamje=np.float64(pamje), aligned=aligned,
         templates=torch.stack([t["verts"] for t in templates]), obj_rot=out["obj_rot"], obj_trans=out["obj_trans"],
         obj_rot_gt=targets["obj_rot"], obj_trans_gt=targets["rel_obj_trans"], obj_cls=obj_cls.numpy(),
         adds=np.float64(adds), mce=np.float64(mce), oce=np.float64(oce), adds_ho3d=np.float64(adds_h),
         mme_ho3d=np.float64(mme_h), gt_v=gt_v, pr_v=pr_v, mesh_mean=np.float64(m3d), mesh_median=np.float64(med),
         mesh_auc=np.float64(auc), mesh_pck=pck, fscore_bruteforce=fs)


def ik_golden():
    """g12: the reference's closed-form IK post-process (common/utils/inverse_kinematics.py:15-150) run on seeded joints.
    The reference imports ``kornia.geometry.conversions.rotation_matrix_to_axis_angle`` (:9, used at :70 and in the finger
    fits).  kornia is a third-party dependency absent from /root/reference and from this image (requirements.txt:7 names it
    without a version); the harness supplies that ONE function as a restatement of kornia's published algorithm (0.6.9 - 0.7.x
    ``kornia/geometry/conversions.py``): rotation matrix -> quaternion (w, x, y, z) by the four-branch trace method with
    eps = 1e-8 under the square roots and divisions clamped at the dtype's tiny, then quaternion -> axis-angle through
    2 atan2(+-sin, +-cos) / sin with the k = 2 small-angle branch - so the fixture carries the reference's arithmetic
    (branch choices, the 1e-8 guard, float32 throughout), not scipy's SO(3) log map as in round 5.  Everything else in the
    fixture is the reference's own code, with ``ManoLayer`` = the synthetic MANO-shaped asset (shim 4 above)."""

    def rotation_matrix_to_quaternion(rotation_matrix, eps=1.0e-8):
        def safe_zero_division(numerator, denominator):
            return numerator / torch.clamp(denominator, min=torch.finfo(numerator.dtype).tiny)
        v = rotation_matrix.reshape(*rotation_matrix.shape[:-2], 9)
        m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.chunk(v, chunks=9, dim=-1)
        trace = m00 + m11 + m22

        def trace_positive_cond():
            sq = torch.sqrt(trace + 1.0 + eps) * 2.0          # 4 qw
            return torch.cat((0.25 * sq, safe_zero_division(m21 - m12, sq), safe_zero_division(m02 - m20, sq),
                              safe_zero_division(m10 - m01, sq)), dim=-1)

        def cond_1():
            sq = torch.sqrt(1.0 + m00 - m11 - m22 + eps) * 2.0  # 4 qx
            return torch.cat((safe_zero_division(m21 - m12, sq), 0.25 * sq, safe_zero_division(m01 + m10, sq),
                              safe_zero_division(m02 + m20, sq)), dim=-1)

        def cond_2():
            sq = torch.sqrt(1.0 + m11 - m00 - m22 + eps) * 2.0  # 4 qy
            return torch.cat((safe_zero_division(m02 - m20, sq), safe_zero_division(m01 + m10, sq), 0.25 * sq,
                              safe_zero_division(m12 + m21, sq)), dim=-1)

        def cond_3():
            sq = torch.sqrt(1.0 + m22 - m00 - m11 + eps) * 2.0  # 4 qz
            return torch.cat((safe_zero_division(m10 - m01, sq), safe_zero_division(m02 + m20, sq),
                              safe_zero_division(m12 + m21, sq), 0.25 * sq), dim=-1)

        where_2 = torch.where(m11 > m22, cond_2(), cond_3())
        where_1 = torch.where((m00 > m11) & (m00 > m22), cond_1(), where_2)
        return torch.where(trace > 0.0, trace_positive_cond(), where_1)

    def quaternion_to_axis_angle(quaternion):
        q1, q2, q3 = quaternion[..., 1], quaternion[..., 2], quaternion[..., 3]
        cos_theta = quaternion[..., 0]
        sin_squared_theta = q1 * q1 + q2 * q2 + q3 * q3
        sin_theta = torch.sqrt(sin_squared_theta)
        two_theta = 2.0 * torch.where(cos_theta < 0.0, torch.atan2(-sin_theta, -cos_theta), torch.atan2(sin_theta, cos_theta))
        k = torch.where(sin_squared_theta > 0.0, two_theta / sin_theta, 2.0 * torch.ones_like(sin_theta))
        return torch.stack((q1 * k, q2 * k, q3 * k), dim=-1)

    def rotation_matrix_to_axis_angle(R):
        # (a reflected palm fit goes through the same arithmetic - NaN or garbage, as in kornia; the reference keeps only the
        # proper rotations, `[batch_id]`, :66-71)
        return quaternion_to_axis_angle(rotation_matrix_to_quaternion(R))

    _stub_module("kornia")
    _stub_module("kornia.geometry")
    _stub_module("kornia.geometry.conversions", rotation_matrix_to_axis_angle=rotation_matrix_to_axis_angle)
    import manopth.manopth.manolayer as ML
    ML.ManoLayer = lambda **k: MANO.ManoLayer(MANO.synthetic_assets(0))
    sys.modules.pop("common.utils.inverse_kinematics", None)
    from common.utils.inverse_kinematics import ik_solver_mano
    layer = MANO.ManoLayer(MANO.synthetic_assets(0))
    r = np.random.default_rng(12)
    B = 8
    pose = torch.from_numpy((0.3 * r.standard_normal((B, 48))).astype(np.float32))
    betas = torch.from_numpy((0.5 * r.standard_normal((B, 10))).astype(np.float32))
    _, j = layer(pose, betas)
    joints = j / 1000.0 + torch.from_numpy((0.004 * r.standard_normal((B, 21, 3))).astype(np.float32))   # noisy "predictions"
    joints = joints + torch.from_numpy((0.3 * r.standard_normal((B, 1, 3))).astype(np.float32))
    joints[-1, :, 0] *= -1                       # a mirrored hand: the palm fit is a reflection -> pose stays zero (:66-71)
    with torch.no_grad():
        res = ik_solver_mano(betas, joints.clone())
        res0 = ik_solver_mano(None, joints.clone())
    save("g12_ik", joints_in=joints, betas=betas, verts=res["verts"], joints=res["joints"], pose=res["pose"],
         vis=res["vis"].numpy(), pose_noshape=res0["pose"], joints_noshape=res0["joints"])


def schema_golden():
    """g10: the reference Model's state-dict schema (names + shapes) for the three variants - what a released
    snapshot_*.pth.tar holds under ckpt["network"] (minus the DataParallel "module." prefix)."""
    import json
    out = {}
    for setting, rt in (("dexycb", 50), ("ho3d", 50), ("ho3d_render", 50), ("dexycb", 18)):
        model, cfg = build_reference(setting, 600, 200, 64, resnet_type=rt)
        out[f"{setting}_resnet{cfg.resnet_type}"] = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(OUT, "g10_state_dict_schema.json"), "w") as f:
        json.dump(out, f)
    print("  wrote g10_state_dict_schema.json", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    install_shims()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["stage", "e2e", "train", "trainB", "mano", "schema", "big", "metrics", "ik", "aux", "sampler", "options", "smallbeta"]
    if "schema" in which:
        schema_golden()
    if "mano" in which:
        mano_golden()
    if "stage" in which:
        stage_goldens()
    if "e2e" in which:
        e2e_goldens()
    if "train" in which:
        train_goldens()
    if "trainB" in which:
        train_branch_b_golden()
    if "big" in which:
        big_goldens()
    if "metrics" in which:
        metrics_golden()
    if "ik" in which:
        ik_golden()
    if "aux" in which:
        aux_loss_golden()
    if "sampler" in which:
        sampler_golden()
    if "options" in which:
        option_goldens()
    if "smallbeta" in which:
        smallbeta_goldens()
