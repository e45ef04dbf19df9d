"""``Model`` / ``get_model``: drop-in for the reference's main/model.py surface with the hot path
(everything after the CNN encoder/decoder) running on hand-written gfx950 kernels.

Same constructor-level module names (state-dict schema of SURVEY.md Appendix D), same
``forward(inputs, targets, meta_info, mode, epoch_cnt, batch_ratio)`` signature and the
``*_out`` key convention (main/train.py:111-112).  Differences, all internal:
  * tokens are batch-first (B,S,256) and the pyramid is consumed channels-last;
  * ``sdf_infer`` is batched on the device (lattice -> bbox compaction -> SDF -> exact top-K),
    one host read of B survivor counts per field instead of 2*B CPU<->GPU hops
    (reference main/model.py:285-352);
  * the dense-grid selection returns the same *set* in the same ascending-|sdf| order.
Reference line numbers are cited per method.
"""
from __future__ import annotations

import contextlib
import os
import random
import weakref
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .config import cfg as _global_cfg
from .nets.blocks import MLP, SDFDecoder, Transformer, VoteTransformer
from .nets.encoder import BackboneNet, DecoderNet
from .nets.heads import JointvoteLoss, ManoHead, ManoLoss, ManoShapeLoss, SepSDFLoss
from .nets.mano import ManoLayer


def get_mano_tgt_mask(cfg=_global_cfg):
    """common/utils/misc.py:11-31 - True = masked."""
    n = cfg.mano_num_queries
    m = torch.ones(n, n, dtype=torch.bool)
    m[0, 0] = False
    for i in range(5):
        m[3 * i + 1:3 * i + 4, 3 * i + 1:3 * i + 4] = False
    m[cfg.mano_shape_indx, cfg.mano_shape_indx] = False
    return m


def get_mano_memory_mask(cfg=_global_cfg):
    """common/utils/misc.py:42-47."""
    m = torch.zeros(cfg.mano_num_queries, cfg.num_samp_hand + cfg.num_samp_obj, dtype=torch.bool)
    m[:, cfg.num_samp_hand:] = True
    return m


def get_manoshape_memory_mask(cfg=_global_cfg):
    """common/utils/misc.py:34-39."""
    m = torch.zeros(1, cfg.num_samp_hand + cfg.num_samp_obj, dtype=torch.bool)
    m[:, cfg.num_samp_hand:] = True
    return m


# per-model cache of the hoisdf_sdf_query_fwd weight descriptors (ctypes structs of raw device pointers): kept OUT of the
# module's __dict__ so that copy.deepcopy(model) / torch.save(model) never see them
_SDFQ_CACHE = weakref.WeakKeyDictionary()


class Model(nn.Module):
    def __init__(self, backbone_net, decoder_net, hand_sdf_decoder, obj_sdf_decoder, hand_transformer,
                 obj_transformer, mano_layer, cfg=_global_cfg):
        super().__init__()
        self.cfg = cfg
        self.backbone_net = backbone_net
        self.decoder_net = decoder_net
        self.hand_sdf_decoder = hand_sdf_decoder
        self.obj_sdf_decoder = obj_sdf_decoder
        self.hand_transformer = hand_transformer
        self.obj_transformer = obj_transformer
        self.hand_sigmoid_beta = nn.Parameter(0.1 * torch.ones(1))
        self.obj_sigmoid_beta = nn.Parameter(0.1 * torch.ones(1))
        D, C = cfg.hidden_dim, cfg.mutliscale_dim
        self.norm1 = nn.LayerNorm(C)                       # defined, never applied (reference :55)
        self.linear_transformerin = MLP(C, [1024, 512, 256], D - cfg.PointFeatSize, 4, True)
        self.linear_sdfin = MLP(C, [512], D, 2, True)
        coord_change_mat = torch.tensor([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
        if cfg.use_inverse_kinematics:
            self.mano_query_embed = nn.Embedding(1, D)
        else:
            self.mano_query_embed = nn.Embedding(cfg.mano_num_queries, D)
            self.mano_head = ManoHead(mano_layer, coord_change_mat=coord_change_mat)
            self.linear_pose = MLP(D, D, 6, 3)
        self.linear_shape = MLP(D, D, 10, 3)
        self.linear_handvote = MLP(D, D, 20 * 3, 4)
        self.linear_handcls = MLP(D, D, 20, 3)
        self.linear_objvote = MLP(D, D, 8 * 3, 4)          # unused in forward (reference :86-87)
        self.linear_objcls = MLP(D, D, 8, 3)
        self.linear_obj_rel_trans = MLP(D, D, 3, 3)
        self.linear_obj_rot = MLP(D, D, 3, 3)
        self.joints_vote_loss = JointvoteLoss(cfg.hand_cls_dist)
        self.sdf_loss = SepSDFLoss()
        if cfg.use_inverse_kinematics:
            self.mano_shape_loss = ManoShapeLoss(cfg.lambda_manoshape, cfg.mano_lambda_regulshape)
        else:
            self.mano_loss = ManoLoss(cfg.lambda_verts3d, cfg.lambda_joints3d, cfg.lambda_manopose,
                                      cfg.lambda_manoshape)
        self.freeze_stages()
        self._py_random = random            # the p < 0.4 branch draw (reference :426); injectable for tests
        self._jitter = None                 # test hook: callable(like, d) -> jitter tensor

    def freeze_stages(self):
        if self.backbone_net is None:
            return
        for name, p in self.backbone_net.named_parameters():
            if "bn" in name:
                p.requires_grad = False

    # ---- pieces ---------------------------------------------------------------------------
    def _pyramid(self, feature_pyramid) -> ops.PyramidNHWC:
        if isinstance(feature_pyramid, ops.PyramidNHWC):
            return feature_pyramid
        return ops.PyramidNHWC.from_nchw([feature_pyramid[k] for k in self.cfg.mutliscale_layers])

    def sdf_activation(self, input, beta):
        """reference :123-126 (sigma = sigmoid(sdf/beta)/beta, beta floored in place)."""
        beta.data.clamp_(min=2e-3)
        return torch.sigmoid(input / beta) / beta

    def _sdf_rows(self, pyr, points, center, cam_intr, scale, kind, sample_idx=None, want_class=False):
        """K1-K4 on a flat list of points: returns (sdf clamped (n,), sdf_raw (n,), pe (n,30), cam (n,3)) and, with ``want_class``
        (cfg.ClassifierBranch: the public sdf_forward / sdf_infer hand the decoder's class logits back, main/model.py:236-240 -
        nothing in Model.forward reads them, so the hot path never asks), the logits (n,6) as a fifth element."""
        c = self.cfg
        dec = self.hand_sdf_decoder if kind == "hand" else self.obj_sdf_decoder
        if (not want_class and sample_idx is None and torch.is_grad_enabled() and ops.sdf_query_train_ok()
                and pyr.C == self.linear_sdfin.layers[0].weight.shape[1]):
            # one C-ABI call per direction (hoisdf_sdf_query_train_fwd / hoisdf_sdf_query_bwd)
            lin = self.linear_sdfin.layers
            routed = [lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias]
            for i in range(4):
                l = getattr(dec, f"linh{i}")
                routed += [l.effective_weight(), l.bias]
            routed += [dec.linh4.weight, dec.linh4.bias]
            sdf, pe, cam = ops.sdf_query_train(self._query_weights(kind), pyr, points, center, cam_intr, scale, c.ClampingDistance,
                                               c.input_img_shape, dec.dropout_prob if dec.training else 0.0, routed)
            return sdf, None, pe, cam
        feat, cam = ops.project_gather(pyr, points, center, cam_intr, scale, c.input_img_shape, sample_idx)
        fea = self.linear_sdfin(feat)
        pts = points.reshape(-1, 3)
        pe = ops.posenc(pts)
        # decoder input rows [feat256 | pe30 | xyz3] in a 292-wide (16-byte aligned) buffer
        n = pts.shape[0]
        x0 = torch.cat([fea, pe, pts, pts.new_zeros(n, 3)], dim=1)[:, :c.hidden_dim + c.PointFeatSize]
        if want_class:
            sdf, raw, cls = dec.forward_clamped(x0, c.ClampingDistance, want_class=True)
            return sdf, raw, pe, cam, cls
        sdf, raw = dec.forward_clamped(x0, c.ClampingDistance)
        return sdf, raw, pe, cam

    def _query_weights(self, kind):
        q = _SDFQ_CACHE.get(self)
        if q is None:
            q = {"hand": ops.SdfQueryWeights(self.linear_sdfin, self.hand_sdf_decoder),
                 "obj": ops.SdfQueryWeights(self.linear_sdfin, self.obj_sdf_decoder)}
            _SDFQ_CACHE[self] = q
        return q[kind]

    def invalidate_sdf_query_weights(self):
        """drop the cached weight-norm folds (call after changing SDF-MLP weights in a way torch's version counters and
        FusedAdamW cannot see, e.g. a foreign kernel writing the parameters in place)"""
        _SDFQ_CACHE.pop(self, None)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_sdf_query_weights()
        return r

    @torch.no_grad()
    def _sdf_query(self, pyr, points, center, cam_intr, scale, kind, sample_idx=None, feat=None, want_feat=False):
        """K1-K4 for the call sites whose results are only used detached (reference :483-484,:517-518,:540,:558) and for
        sdf_infer: one C-ABI call (hoisdf_sdf_query_fwd), optionally on rows already gathered for the same camera points.
        Dropout follows the decoder's train()/eval() mode like the reference's module calls do.
        -> (sdf clamped (n,), sdf_raw (n,), pe (n,30), feat (n,C) | None)"""
        c = self.cfg
        dec = self.hand_sdf_decoder if kind == "hand" else self.obj_sdf_decoder
        p = dec.dropout_prob if dec.training else 0.0
        sdf, raw, pe, _, f = ops.sdf_query(self._query_weights(kind), pyr, points, center, cam_intr, scale,
                                           c.ClampingDistance, c.input_img_shape, sample_idx, feat, want_feat, False, p)
        return sdf, raw, pe, f

    def sdf_forward(self, feature_pyramid, sdf_points, center_joint, cam_intr, sdf_scale, type="hand"):
        """reference :181-244 -> (pred_sdf (B,P,1), pred_class (B,P,6) with cfg.ClassifierBranch else None, pos_enc3d (B,P,30))."""
        B, P, _ = sdf_points.shape
        if self.cfg.ClassifierBranch:
            sdf, _, pe, _, cls = self._sdf_rows(self._pyramid(feature_pyramid), sdf_points, center_joint, cam_intr, sdf_scale, type,
                                                want_class=True)
            return sdf.view(B, P, 1), cls.view(B, P, -1), pe.view(B, P, -1)
        sdf, _, pe, _ = self._sdf_rows(self._pyramid(feature_pyramid), sdf_points, center_joint, cam_intr,
                                       sdf_scale, type)
        return sdf.view(B, P, 1), None, pe.view(B, P, -1)

    def get_input_transformer(self, feature_pyramid, sdf_points, center_joint, cam_intr, sdf_scale):
        """reference :145-179 -> (transformer_latent (B,P,223), cam_sdf_points (B,P,3))."""
        B, P, _ = sdf_points.shape
        feat, cam = ops.project_gather(self._pyramid(feature_pyramid), sdf_points, center_joint, cam_intr, sdf_scale,
                                       self.cfg.input_img_shape)
        return self.linear_transformerin(feat).view(B, P, -1), cam.view(B, P, 3)

    @torch.no_grad()
    def sdf_infer(self, feature_pyramid, center_joint, cam_intr, bbox, sdf_scale, num_points, type="hand", counts=None):
        """reference :246-355, batched on the device -> (points (B,K,3), sdf (B,K,1), posenc (B,K,30), class logits (B,K,6) | None).
        ``counts``: the lattice-survivor count queued earlier (``infer_counts_begin``), so that nothing drains here."""
        c = self.cfg
        pyr = self._pyramid(feature_pyramid)
        B = center_joint.shape[0]
        dec = self.hand_sdf_decoder if type == "hand" else self.obj_sdf_decoder
        try:
            pose_points, pose_sdf, pose_pe = ops.sdf_infer(self._query_weights(type), pyr, center_joint, cam_intr, bbox, sdf_scale,
                                                           c.bins_n, num_points, c.ClampingDistance, c.input_img_shape,
                                                           dec.dropout_prob if dec.training else 0.0, counts)
        except ValueError as e:
            raise ValueError(str(e).replace("sdf_infer:", f"sdf_infer({type}):")) from None
        pose_sdf = pose_sdf.unsqueeze(-1)
        pose_class = None
        if c.ClassifierBranch:          # :351-352: the logits of the selected points (the decoder re-evaluated on them)
            pose_class = self.sdf_forward(pyr, pose_points, center_joint, cam_intr, sdf_scale, type)[1]
        return pose_points, pose_sdf, pose_pe, pose_class

    def render_gaussian_heatmap(self, joint_coord):
        """reference :128-143 (encoder-side auxiliary target; plain torch)."""
        c = self.cfg
        x = torch.arange(c.output_hm_shape[2], device=joint_coord.device).float()
        y = torch.arange(c.output_hm_shape[1], device=joint_coord.device).float()
        yy, xx = torch.meshgrid(y, x, indexing="ij")
        jx, jy = joint_coord[:, :, 0, None, None], joint_coord[:, :, 1, None, None]
        hm = torch.exp(-(((xx[None, None] - jx) / c.sigma) ** 2) / 2 - (((yy[None, None] - jy) / c.sigma) ** 2) / 2)
        return hm.sum(1) * 255

    # ---- the point-branch draw and the early survivor counts -----------------------------------
    def draw_branch(self, mode, epoch_cnt=1e8) -> bool:
        """reference :426-427: the ONE python-random draw of a forward; True = pre-sampled points + jitter (branch A),
        False = query points from the dense-lattice sdf_infer (always in eval)."""
        p = self._py_random.uniform(0, 1)
        return (p < 0.4 or epoch_cnt < self.cfg.point_sampling_epoch) and mode == "train"

    def infer_counts_begin(self, meta_info):
        """queue the lattice-survivor counts of both fields (they depend on mano_root / obj_center_cam / cam_intr / bbox only,
        reference :286-302).  Model.forward calls this BEFORE the image encoder; sdf_infer then waits for an event that has
        long fired instead of draining the device twice per field (round 3: four pipeline drains per eval step)."""
        c = self.cfg
        K = meta_info["cam_intr"]
        return {"hand": ops.sdf_infer_count_begin(meta_info["mano_root"], K, meta_info["bbox_hand"], c.hand_sdf_scale, c.bins_n),
                "obj": ops.sdf_infer_count_begin(meta_info["obj_center_cam"], K, meta_info["bbox_obj"], c.obj_sdf_scale, c.bins_n)}

    # ---- the hot path ----------------------------------------------------------------------
    def hot_path(self, pyr: ops.PyramidNHWC, inputs, targets, meta_info, mode, epoch_cnt=1e8, batch_ratio=0, branch_a=None,
                 infer_counts=None):
        """reference :370-402 and :424-662: everything after decoder_net except the aux image losses.  ``branch_a`` /
        ``infer_counts``: the draw and the queued survivor counts when the caller (Model.forward) made them ahead of the
        encoder; drawn / queued here otherwise."""
        c = self.cfg
        training = mode == "train"
        if training:
            pyr = pyr.shared_grad()          # the step's four gather backwards scatter into ONE set of level gradients
        loss: Dict[str, torch.Tensor] = {}
        out: Dict[str, torch.Tensor] = {}
        root, ocen, K = meta_info["mano_root"], meta_info["obj_center_cam"], meta_info["cam_intr"]
        hs_, os_ = c.hand_sdf_scale, c.obj_sdf_scale
        nh, no = c.num_samp_hand, c.num_samp_obj
        B = root.shape[0]

        # Everything that starts from the OBJECT points (their SDF query, input MLP, evaluation in the hand field and,
        # below, the object encoder stack) is independent of the hand-point work until the tokens are assembled: it is
        # issued on a second HIP stream, so its small grids (16 384 rows) share the chip with the hand stream's kernels
        # and kernel tails overlap.  Autograd replays every op's backward on its forward stream.
        two = bool(getattr(c, "overlap_streams", True)) and root.is_cuda and os.environ.get("HOISDF_TWO_STREAMS", "1") != "0" \
            and not ops.deterministic()
        cur = side = None
        if two:
            cur = torch.cuda.current_stream(root.device)
            if getattr(self, "_side_stream", None) is None:
                # HIGH priority = a hardware queue of its own.  HIP maps normal-priority streams onto 4 hardware queues
                # round-robin: once a RCCL process group has created its streams, a normal-priority second stream lands on
                # the compute stream's queue and the two never overlap (measured with world 1 through RCCL: 95.6 vs 91.3
                # ms/step, zero concurrent kernels in the trace; GPU_MAX_HW_QUEUES=8 cures it as well).  Without a process
                # group the priority changes nothing (92.8 vs 92.9 ms/step).  HOISDF_SIDE_PRIORITY=0: normal priority.
                prio = 0 if os.environ.get("HOISDF_SIDE_PRIORITY") == "0" else -1
                self._side_stream = torch.cuda.Stream(device=root.device, priority=prio)
            side = self._side_stream
        on_side = (lambda: torch.cuda.stream(side)) if two else contextlib.nullcontext
        want_sdf_loss = training or c.dataset == "dexycb"                                # :370-402

        if branch_a is None:
            branch_a = self.draw_branch(mode, epoch_cnt)                               # :426-427
        if not branch_a and infer_counts is None:
            infer_counts = self.infer_counts_begin(meta_info)      # both fields queued before anything else of this path
        log = getattr(self, "branch_log", None)
        if log is not None and training:
            log.append("A" if branch_a else "B")             # bench.py --branch-mix reports the mix it measured
        if branch_a:
            d = c.random_move_dist[len([a for a in c.random_ratio if batch_ratio > a])]
            jit = self._jitter or (lambda like, dd: torch.empty_like(like).uniform_(-dd, dd))
            hand_points = inputs["hand_pre_points"] + jit(inputs["hand_pre_points"], d)
            obj_points = inputs["obj_pre_points"] + jit(inputs["obj_pre_points"], d)
        else:                                                                          # :462-481
            ic = infer_counts or {}
            hand_points, hand_sdf, hand_pe, _ = self.sdf_infer(pyr, root, K, meta_info["bbox_hand"], hs_, nh, "hand", ic.get("hand"))
            obj_points, obj_sdf, obj_pe, _ = self.sdf_infer(pyr, ocen, K, meta_info["bbox_obj"], os_, no, "obj", ic.get("obj"))
        self.hand_sigmoid_beta.data.clamp_(min=2e-3)                                   # :124
        self.obj_sigmoid_beta.data.clamp_(min=2e-3)

        so = None
        # both streams read the cached SDF-query weight descriptors: (re)build them HERE, on the ambient stream and
        # ahead of side.wait_stream, so neither stream can launch a query before the folded weights are written
        self._query_weights("hand").get()
        self._query_weights("obj").get()
        if two:
            side.wait_stream(cur)                       # the points, the pyramid, the inputs and the query weights are ready
        with on_side():                                 # ---- object points ----
            if want_sdf_loss:
                so, _, _ = self.sdf_forward(pyr, inputs["obj_sdf_points"], ocen, K, os_, "obj")
            # ONE gather of the object points' pixels feeds the token MLP (with gradient), the object field and - the
            # camera points being the same - the evaluation of those points in the hand field (the reference gathers
            # them three times, :445/:486/:499; its detached queries run as single hoisdf_sdf_query_fwd calls)
            obj_feat, obj_cam = ops.project_gather(pyr, obj_points, ocen, K, os_, c.input_img_shape)
            obj_cam = obj_cam.view(B, no, 3)
            fused_tok = ops.tokens_ok(obj_feat)     # K7 + K8 as one C call per point set (hoisdf_tokens_fwd), issued once sdf / pe exist
            if not fused_tok:
                obj_fea = self.linear_transformerin(obj_feat).view(B, no, -1)                           # :486-493
            fd = obj_feat.detach()
            if branch_a:                # the reference tracks these calls but only ever uses them detached
                obj_sdf, _, obj_pe, _ = self._sdf_query(pyr, obj_points, ocen, K, os_, "obj", feat=fd)
                obj_sdf, obj_pe = obj_sdf.view(B, no, 1), obj_pe.view(B, no, -1)
            obj_h_pts = (obj_cam - root[:, None, :]) * hs_                                              # :495-518
            obj_h_sdf, _, obj_h_pe, _ = self._sdf_query(pyr, obj_h_pts, root, K, hs_, "hand", feat=fd)
            obj_h_sdf, obj_h_pe = obj_h_sdf.view(B, no, 1), obj_h_pe.view(B, no, -1)
            S, D = nh + no, c.hidden_dim
            tin_w = [l.weight for l in self.linear_transformerin.layers]
            tin_b = [l.bias for l in self.linear_transformerin.layers]
            if fused_tok:                       # the object points' own token rows (+ their MLP output for the hand stream's cross rows)
                obj_tok = torch.empty(B, S, D, device=root.device)
                obj_tok, obj_fea = ops.tokens(obj_tok, obj_feat, obj_cam, ocen, obj_pe, obj_sdf.detach(), self.obj_sigmoid_beta, 0,
                                              tin_w, tin_b)
                obj_fea = obj_fea.view(B, no, -1)
        # ---- hand points (ambient stream) ----
        if want_sdf_loss:
            sh, _, _ = self.sdf_forward(pyr, inputs["hand_sdf_points"], root, K, hs_, "hand")
        hand_feat, hand_cam = ops.project_gather(pyr, hand_points, root, K, hs_, c.input_img_shape)
        hand_cam = hand_cam.view(B, nh, 3)
        if not fused_tok:
            hand_fea = self.linear_transformerin(hand_feat).view(B, nh, -1)
        fd = hand_feat.detach()
        if branch_a:
            hand_sdf, _, hand_pe, _ = self._sdf_query(pyr, hand_points, root, K, hs_, "hand", feat=fd)
            hand_sdf, hand_pe = hand_sdf.view(B, nh, 1), hand_pe.view(B, nh, -1)
        hand_rel = hand_cam - root[:, None, :]
        hand_o_pts = (hand_cam - ocen[:, None, :]) * os_
        hand_o_sdf, _, hand_o_pe, _ = self._sdf_query(pyr, hand_o_pts, ocen, K, os_, "obj", feat=fd)
        hand_o_sdf, hand_o_pe = hand_o_sdf.view(B, nh, 1), hand_o_pe.view(B, nh, -1)
        if two:
            cur.wait_stream(side)
            for t in (so, obj_sdf, obj_pe, obj_fea, obj_cam, obj_h_sdf, obj_h_pe):          # (obj_tok, when built there: below)
                if t is not None:
                    t.record_stream(cur)
        if want_sdf_loss:
            loss["sdfhand_loss"], loss["sdfobj_loss"] = self.sdf_loss(
                sh, so, targets["hand_sdf"], targets["obj_sdf"], clamp=c.ClampingDistance)       # :393-402, clamp fused

        # token streams (batch-first).  The appended cross-field tokens are detached (:540,:558) and use
        # the *other* centre for xyz ("# bug" lines :498,:508 replicated).
        S, D = nh + no, c.hidden_dim
        dev = root.device
        if fused_tok:
            hand_tok = torch.empty(B, S, D, device=dev)
            hand_tok, hand_fea = ops.tokens(hand_tok, hand_feat, hand_cam, root, hand_pe, hand_sdf.detach(), self.hand_sigmoid_beta, 0,
                                            tin_w, tin_b)
            hand_fea = hand_fea.view(B, nh, -1)
            if two:
                obj_tok.record_stream(cur)
            with torch.no_grad():               # the cross rows, written into the buffers the two calls above returned
                ops.token_build(hand_tok, obj_cam.reshape(-1, 3), root, obj_h_pe, obj_fea, obj_h_sdf,
                                self.hand_sigmoid_beta.detach(), nh)
                ops.token_build(obj_tok, hand_cam.reshape(-1, 3), ocen, hand_o_pe, hand_fea, hand_o_sdf,
                                self.obj_sigmoid_beta.detach(), no)
        else:
            hand_tok = torch.empty(B, S, D, device=dev)
            obj_tok = torch.empty(B, S, D, device=dev)
            with torch.no_grad():
                ops.token_build(hand_tok, obj_cam.reshape(-1, 3), root, obj_h_pe, obj_fea.detach(), obj_h_sdf,
                                self.hand_sigmoid_beta.detach(), nh)
                ops.token_build(obj_tok, hand_cam.reshape(-1, 3), ocen, hand_o_pe, hand_fea.detach(), hand_o_sdf,
                                self.obj_sigmoid_beta.detach(), no)
            hand_tok = ops.token_build(hand_tok, hand_cam.reshape(-1, 3), root, hand_pe, hand_fea, hand_sdf.detach(),
                                       self.hand_sigmoid_beta, 0)
            obj_tok = ops.token_build(obj_tok, obj_cam.reshape(-1, 3), ocen, obj_pe, obj_fea, obj_sdf.detach(),
                                      self.obj_sigmoid_beta, 0)

        tgt_mask = None if c.use_inverse_kinematics else get_mano_tgt_mask(c)         # :564-569
        # Only rows < nh (hand stream) / < no (object stream) of the encoder outputs are ever read (:587-593 and
        # the memory mask), so the last layer of each stack skips the other query rows - same values, less work.
        # The object encoder stack (+ its heads) goes to the second stream as well (127.3 -> 126.0 ms/step on its own).
        if two:
            side.wait_stream(cur)
            obj_tok.record_stream(side)
            with torch.cuda.stream(side):
                _, obj_enc = self.obj_transformer.forward_batch_first(obj_tok, n_keep=no)      # :582-584
                obj_rot = self.linear_obj_rot(obj_enc)                                         # (L,B,no,3)
                obj_trans = self.linear_obj_rel_trans(obj_enc)
        hs, memory, hand_enc = self.hand_transformer.forward_batch_first(
            hand_tok, self.mano_query_embed.weight, tgt_mask, nh, n_keep=nh)           # :571-581
        if two:
            side.wait_stream(cur)                       # hs is ready; the side stream already holds the object stack
            hs.record_stream(side)
        else:
            _, obj_enc = self.obj_transformer.forward_batch_first(obj_tok, n_keep=no)      # :582-584
            obj_rot = self.linear_obj_rot(obj_enc)                                         # (L,B,no,3)
            obj_trans = self.linear_obj_rel_trans(obj_enc)

        # ---- MANO head (two launches: ground truth, predictions + fused losses) + the object pose losses, on the second
        # stream under the big vote-head GEMMs of the ambient stream
        pred_m = gt_m = None
        side_made = []
        with on_side():
            if c.use_inverse_kinematics:                                               # :595-597
                mano_shape = self.linear_shape(hs[:, :, 0])
                out["mano_shape_out"] = mano_shape[-1]
            else:                                                                      # :599-620
                pose6d = self.linear_pose(hs[:, :, :c.mano_shape_indx])                # (L,B,16,6)
                mano_shape = self.linear_shape(hs[:, :, c.mano_shape_indx])            # (L,B,10)
                mp = targets["mano_param"] if (training or c.dataset == "dexycb") else None
                pred_m, gt_m = self.mano_head.forward_batch_first(pose6d, mano_shape, mp)
                out["mano_mesh_out"] = pred_m["verts3d"][-1]
                out["mano_joints_out"] = pred_m["joints3d"][-1]
                if c.dataset == "dexycb":
                    out["mano_joints_gt_out"] = gt_m["joints3d"]
                    out["mano_mesh_gt_out"] = gt_m["verts3d"]
            if not training:                                                           # :622-624
                out["obj_rot_out"] = obj_rot[-1].contiguous()
                out["obj_trans_out"] = obj_trans[-1].contiguous()
            if training or c.dataset == "dexycb":                                      # :640-654
                if c.use_inverse_kinematics:
                    loss["shape_param_loss"], loss["shape_reg_loss"] = self.mano_shape_loss(
                        mano_shape, targets["mano_param"][:, -10:])
                else:
                    (loss["mano_mesh_loss"], loss["mano_joint_loss"], loss["pose_param_loss"],
                     loss["shape_param_loss"], _, _) = self.mano_loss(pred_m, gt_m)
            loss["obj_rot"] = ops.smooth_l1_loss_broadcast(obj_rot, targets["obj_rot"], no)            # :656-662 (a15: HIP reductions)
            loss["obj_trans"] = ops.smooth_l1_loss_broadcast(obj_trans, targets["rel_obj_trans"], no)
            side_made = [t for t in list(loss.values()) + list(out.values()) if torch.is_tensor(t)]

        # ---- hand vote heads + vote aggregation / losses (ambient stream)
        if training or c.dataset == "dexycb":                                          # :626-638
            joints_gt = targets["joint_cam_no_trans"][:, 1:]
        else:
            joints_gt = torch.zeros(B, 20, 3, device=dev)
        if ops.tokens_ok(hand_enc):             # K11 + K12 as one C call per direction (hoisdf_heads_vote_fwd / _bwd)
            (loss["loss_joint_3d"], loss["loss_joint_cls"], loss["loss_all_joint_3d"],
             joints) = self.joints_vote_loss.forward_fused(hand_rel, hand_enc, self.linear_handvote, self.linear_handcls, joints_gt)
        else:
            hand_off = self.linear_handvote(hand_enc)                                  # :587-593 (L,B,nh,60)
            hand_cls = self.linear_handcls(hand_enc)
            (loss["loss_joint_3d"], loss["loss_joint_cls"], loss["loss_all_joint_3d"],
             joints) = self.joints_vote_loss(hand_rel, hand_off, hand_cls, joints_gt, batch_first=True)
        out["hand_joints_out"] = joints[-1]
        if two:
            cur.wait_stream(side)
            for t in side_made:
                t.record_stream(cur)
        return loss, out

    def forward(self, inputs, targets, meta_info, mode, epoch_cnt=1e8, batch_ratio=0):
        """reference :357-665."""
        c = self.cfg
        branch_a = self.draw_branch(mode, epoch_cnt)                                  # :426-427, drawn ahead of the encoder
        infer_counts = None if branch_a else self.infer_counts_begin(meta_info)       # read back under the encoder's kernels
        img_feat, skips = self.backbone_net(inputs["img"])                            # :367-368 (PyTorch / MIOpen)
        feature_pyramid, decoder_out = self.decoder_net(img_feat, skips)
        pyr = self._pyramid(feature_pyramid)
        ops.set_attention_f16_eval(bool(getattr(c, "attention_f16_eval", False)) and mode != "train")
        if getattr(c, "gemm_emu", None) is not None:
            ops.set_gemm_emu(bool(c.gemm_emu))
        if getattr(c, "attention_emu", None) is not None:
            ops.set_attention_emu(bool(c.attention_emu))
        loss, out = self.hot_path(pyr, inputs, targets, meta_info, mode, epoch_cnt, batch_ratio, branch_a, infer_counts)
        if mode == "train" or c.dataset == "dexycb":                                   # :404-422 aux image losses
            out["joint_heatmap_out"] = decoder_out[:, 0]
            out["hand_seg_gt_out"] = targets["hand_seg"]
            out["hand_seg_pred_out"] = decoder_out[:, 1]
            out["obj_seg_gt_out"] = targets["obj_seg"]
            out["obj_seg_pred_out"] = decoder_out[:, 2]
            # (f4) one HIP pass: Gaussian heat-map target + MSE + the two BCE maps
            loss["joint_heatmap"], loss["obj_seg"], loss["hand_seg"], _ = ops.aux_image_losses(
                decoder_out, targets["joint_coord"], targets["hand_seg"], targets["obj_seg"], c.sigma)
        return {**loss, **out}


def init_weights(m):
    """reference :668-679."""
    if isinstance(m, (nn.ConvTranspose2d,)):
        nn.init.normal_(m.weight, std=0.001)
    elif isinstance(m, nn.Conv2d):
        nn.init.normal_(m.weight, std=0.001)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.constant_(m.weight, 1)
        nn.init.constant_(m.bias, 0)
    elif type(m) is nn.Linear:
        nn.init.normal_(m.weight, std=0.01)
        nn.init.constant_(m.bias, 0)


def get_model(mode, cfg=_global_cfg, mano_layer=None, with_encoder=True):
    """reference :682-766.  ``mano_layer``: a module with manopth's ManoLayer interface; defaults to
    the synthetic MANO-shaped asset (the licensed MANO_RIGHT.pkl is not redistributable)."""
    backbone = BackboneNet(cfg.resnet_type) if with_encoder else None
    decoder = DecoderNet(cfg.resnet_type, big=cfg.use_big_decoder) if with_encoder else None
    mk = lambda: SDFDecoder(cfg.hidden_dim, cfg.PointFeatSize, use_classifier=cfg.ClassifierBranch)
    hand_dec, obj_dec = mk(), mk()
    hand_tr = Transformer(d_model=cfg.hidden_dim, dropout=cfg.dropout, nhead=cfg.nheads,
                          dim_feedforward=cfg.dim_feedforward, num_encoder_layers=cfg.enc_layers,
                          num_decoder_layers=cfg.dec_layers, normalize_before=cfg.pre_norm,
                          return_intermediate_dec=True)
    obj_tr = VoteTransformer(d_model=cfg.hidden_dim, dropout=cfg.dropout, nhead=cfg.nheads,
                             dim_feedforward=cfg.dim_feedforward, num_encoder_layers=cfg.enc_layers // 2,
                             normalize_before=cfg.pre_norm, return_intermediate_dec=True)
    if mano_layer is None:
        mano_layer = ManoLayer()
    if mode == "train":
        if decoder is not None:
            decoder.apply(init_weights)
        for m in (hand_tr, obj_tr, hand_dec, obj_dec):
            m.apply(init_weights)
        for dec in (hand_dec, obj_dec):      # weight-normed layers: only the bias is re-initialised
            for i in range(4):
                nn.init.constant_(getattr(dec, f"linh{i}").bias, 0)
    return Model(backbone, decoder, hand_dec, obj_dec, hand_tr, obj_tr, mano_layer, cfg=cfg)
