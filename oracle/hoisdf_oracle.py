"""CPU oracle for the HOISDF SDF-query + field-guided pose-regression hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``hoisdf_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
only as the checker / the timed CPU baseline - never as the product path.

What it is: a functional (no nn.Module) PyTorch-CPU restatement of everything the
reference executes in ``Model.forward`` after the CNN encoder/decoder
(/root/reference/main/model.py:370-665), written from the math in SURVEY.md
Appendix A.  It keeps the reference's *op sequence* (NCHW ``grid_sample``, a
per-sample dense-grid loop with a full sort, materialised SxS attention, dropout
in train mode) so that timing it on host cores is a fair stand-in for the
reference's CPU path, and so that autograd gives the reference's gradients.

Parity status: **pinned** against golden vectors produced by importing the real
reference in the build container (tests/golden/make_golden.py -> tests/golden/*.npz;
checked by tests/test_oracle_golden.py).  The reference itself ships no tests or
golden vectors for this path, and most of its arithmetic lives
in un-vendored PyTorch (README pin torch==1.12.1; the goldens were produced with
torch 2.10).

Parameters are passed as a flat ``dict`` keyed by the reference's state-dict names
(SURVEY.md Appendix D), without the ``module.`` prefix.
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


@dataclass
class OracleCfg:
    """The subset of /root/reference/main/config.py:38-189 the hot path reads."""

    num_samp_hand: int = 600            # config.py:64
    num_samp_obj: int = 200             # config.py:65
    hand_sdf_scale: float = 3.1         # config.py:74,82
    obj_sdf_scale: float = 3.1          # config.py:75,83
    hand_cls_dist: float = 0.04         # config.py:76,84
    bins_n: int = 64                    # config.py:88
    PointFeatSize: int = 33             # config.py:90
    ClampingDistance: float = 0.15      # config.py:92
    use_inverse_kinematics: bool = False  # config.py:97
    dataset: str = "dexycb"             # config.py:41-44
    mutliscale_layers: Tuple[str, ...] = ("stride2", "stride4", "stride8", "stride16", "stride32")
    input_img_shape: Tuple[int, int] = (256, 256)   # config.py:111
    hidden_dim: int = 256               # config.py:116
    dropout: float = 0.1                # config.py:117
    nheads: int = 4                     # config.py:118
    enc_layers: int = 6                 # config.py:120
    dec_layers: int = 4                 # config.py:121
    mano_num_queries: int = 17          # config.py:125
    mano_shape_indx: int = 16           # config.py:126
    point_sampling_epoch: int = 40      # config.py:130
    random_ratio: Tuple[float, float] = (0.3, 0.7)            # config.py:68
    random_move_dist: Tuple[float, float, float] = (0.03, 0.05, 0.07)  # config.py:69
    sdf_dropout: float = 0.2            # common/nets/sdf_net.py:20
    pre_norm: bool = False              # config.py:122 (no released configuration turns it on)
    ClassifierBranch: bool = False      # config.py:91  (dito)
    lambda_verts3d: float = 1e4         # config.py:150-154
    lambda_joints3d: float = 1e4
    lambda_manopose: float = 10
    lambda_manoshape: float = 0.1
    mano_lambda_regulshape: float = 0.000001


# --------------------------------------------------------------------------------------
# A.1 projection + multi-scale bilinear gather        (main/model.py:148-175, 190-214)
# --------------------------------------------------------------------------------------
def project_points(points: Tensor, center: Tensor, cam_intr: Tensor, scale: float,
                   img_shape=(256, 256)) -> Tuple[Tensor, Tensor]:
    """cam = p/s + c; uv = (K cam)_xy / (K cam)_z; grid = (uv - n)/n, n = ((W-1)/2,(H-1)/2).
    main/model.py:148-157.  ``grid`` carries no gradient (``.detach().clone()`` at :157)."""
    cam = points / scale + center[:, None, :]
    q = torch.bmm(cam, cam_intr.transpose(1, 2))
    uv = q[:, :, :2] / q[:, :, [2]]
    n = torch.tensor([img_shape[1] - 1, img_shape[0] - 1], dtype=uv.dtype) / 2
    grid = (uv.detach().clone() - n) / n
    return cam, grid


def sample_pyramid(pyramid: Dict[str, Tensor], grid: Tensor, layers) -> Tensor:
    """Bilinear, border-clamped, align_corners=True gather from each NCHW level, concatenated
    in ``layers`` order -> (B, P, C).  main/model.py:159-175."""
    g = grid.unsqueeze(1)
    feats = [F.grid_sample(pyramid[name], g, padding_mode="border", align_corners=True)
             for name in layers]
    f = torch.cat(feats, dim=1).squeeze(2)
    return f.permute(0, 2, 1).contiguous()


def bilinear_gather_explicit(fmap: Tensor, grid: Tensor) -> Tensor:
    """Same gather written out tap by tap (the formula the HIP kernel implements; used by the
    tests to cross-check ``F.grid_sample`` semantics: clip the coordinate first, then floor).
    fmap (B,C,H,W), grid (B,P,2) -> (B,P,C)."""
    B, C, H, W = fmap.shape
    x = ((grid[..., 0] + 1) / 2 * (W - 1)).clamp(0, W - 1)
    y = ((grid[..., 1] + 1) / 2 * (H - 1)).clamp(0, H - 1)
    x0 = x.floor()
    y0 = y.floor()
    wx = x - x0
    wy = y - y0
    x0 = x0.long()
    y0 = y0.long()
    x1 = (x0 + 1).clamp(max=W - 1)
    y1 = (y0 + 1).clamp(max=H - 1)
    flat = fmap.reshape(B, C, H * W)

    def tap(yy, xx):
        idx = (yy * W + xx)[:, None, :].expand(B, C, -1)
        return flat.gather(2, idx)

    out = (tap(y0, x0) * ((1 - wx) * (1 - wy))[:, None]
           + tap(y0, x1) * (wx * (1 - wy))[:, None]
           + tap(y1, x0) * ((1 - wx) * wy)[:, None]
           + tap(y1, x1) * (wx * wy)[:, None])
    return out.permute(0, 2, 1).contiguous()


# --------------------------------------------------------------------------------------
# A.2 MLP, A.3 posenc, A.4 SDF decoder
# --------------------------------------------------------------------------------------
def mlp(x: Tensor, P: Params, prefix: str, num_layers: int, act_last: bool) -> Tensor:
    """Linear(+ReLU) chain.  common/nets/layer.py:168-201."""
    for i in range(num_layers):
        x = F.linear(x, P[f"{prefix}.layers.{i}.weight"], P[f"{prefix}.layers.{i}.bias"])
        if i < num_layers - 1 or act_last:
            x = F.relu(x)
    return x


def posenc(p: Tensor, n_freq: int = 5) -> Tensor:
    """[sin(2^k p), cos(2^k p)]_{k=0..n_freq-1}, each a 3-vector -> 6*n_freq columns.
    common/utils/sdf_utils.py:96-141 (include_input=False, log sampling)."""
    cols = []
    for k in range(n_freq):
        f = 2.0 ** k
        cols.append(torch.sin(p * f))
        cols.append(torch.cos(p * f))
    return torch.cat(cols, dim=-1)


def weightnorm_weight(P: Params, prefix: str) -> Tensor:
    """W[r,:] = g[r] * v[r,:] / ||v[r,:]||_2   (nn.utils.weight_norm, dim=0).
    common/nets/sdf_net.py:57-62."""
    v = P[prefix + ".weight_v"]
    g = P[prefix + ".weight_g"]
    return v * (g / v.norm(dim=1, keepdim=True))


def sdf_decoder(x0: Tensor, P: Params, prefix: str, training: bool = False,
                p_drop: float = 0.2, classifier: bool = False):
    """x0 (P,289) -> tanh output (P,1).  common/nets/sdf_net.py:87-122.
    Layers 0..3 weight-normed + ReLU + dropout; input re-concatenated before layer 2.
    ``classifier`` (use_classifier, :73-75,93-94,119-122): also the 6 class logits of ``classifier_head`` applied to the input of
    the last layer -> (sdf, logits)."""
    h = x0
    logits = None
    for layer in range(5):
        if classifier and layer == 4:
            logits = F.linear(h, P[f"{prefix}.classifier_head.weight"], P[f"{prefix}.classifier_head.bias"])
        if layer == 2:
            h = torch.cat([h, x0], dim=1)
        if layer < 4:
            w = weightnorm_weight(P, f"{prefix}.linh{layer}")
        else:
            w = P[f"{prefix}.linh4.weight"]
        h = F.linear(h, w, P[f"{prefix}.linh{layer}.bias"])
        if layer < 4:
            h = F.relu(h)
            h = F.dropout(h, p=p_drop, training=training)
    out = torch.tanh(h)[:, 0:1]
    return (out, logits) if classifier else out


def sdf_decoder_input(points_fea: Tensor, pts: Tensor) -> Tuple[Tensor, Tensor]:
    """[feat256 | posenc30 | xyz3] rows.  main/model.py:218-228."""
    pe = posenc(pts.reshape(-1, 3))
    x0 = torch.cat([points_fea.reshape(-1, points_fea.shape[-1]), pe, pts.reshape(-1, 3)], 1)
    return x0.contiguous(), pe


def sdf_forward(P: Params, cfg: OracleCfg, pyramid, points: Tensor, center: Tensor,
                cam_intr: Tensor, scale: float, kind: str, training: bool = False):
    """main/model.py:181-244 -> (sdf (B,P,1) clamped, posenc (B,P,30))."""
    B, Np, _ = points.shape
    _, grid = project_points(points, center, cam_intr, scale, cfg.input_img_shape)
    feats = sample_pyramid(pyramid, grid, cfg.mutliscale_layers)
    points_fea = mlp(feats, P, "linear_sdfin", 2, True)
    x0, pe = sdf_decoder_input(points_fea, points)
    sdf = sdf_decoder(x0, P, f"{kind}_sdf_decoder", training, cfg.sdf_dropout, cfg.ClassifierBranch)
    if cfg.ClassifierBranch:            # main/model.py:236-240: the logits ride along (nothing downstream reads them)
        sdf, logits = sdf
        sdf = sdf.reshape(B, Np, 1).clamp(-cfg.ClampingDistance, cfg.ClampingDistance)
        return sdf, pe.reshape(B, Np, -1), logits.reshape(B, Np, 6)
    sdf = sdf.reshape(B, Np, 1).clamp(-cfg.ClampingDistance, cfg.ClampingDistance)
    return sdf, pe.reshape(B, Np, -1)


# --------------------------------------------------------------------------------------
# A.5 / A.6 dense lattice + per-sample select           (main/model.py:246-355)
# --------------------------------------------------------------------------------------
def dense_lattice(bins_n: int) -> Tensor:
    """The sheared lattice of main/model.py:257-273: z index is integer, y and x indices come
    from *true* division of an int64 index (fractional), all arithmetic in float32."""
    v = 2.0 / (bins_n - 1)
    idx = torch.arange(0, bins_n ** 3, 1, dtype=torch.long)
    s = torch.zeros(bins_n ** 3, 3)
    s[:, 2] = idx % bins_n
    s[:, 1] = (idx / bins_n) % bins_n
    s[:, 0] = ((idx / bins_n) / bins_n) % bins_n
    s[:, 0] = s[:, 0] * v - 1
    s[:, 1] = s[:, 1] * v - 1
    s[:, 2] = s[:, 2] * v - 1
    return s


def lattice_bbox_mask(lattice: Tensor, center_b: Tensor, cam_intr_b: Tensor, bbox_b: Tensor,
                      scale: float) -> Tuple[Tensor, Tensor]:
    """Strict bbox test of the projected lattice for one sample.  main/model.py:286-300."""
    cam = lattice / scale + center_b.unsqueeze(0)
    q = torch.mm(cam, cam_intr_b.transpose(0, 1))
    uv = q[:, :2] / q[:, [2]]
    keep = ((uv[:, 0] > bbox_b[0]) & (uv[:, 0] < bbox_b[2])
            & (uv[:, 1] > bbox_b[1]) & (uv[:, 1] < bbox_b[3]))
    return keep, uv


def sdf_infer(P: Params, cfg: OracleCfg, pyramid, center: Tensor, cam_intr: Tensor,
              bbox: Tensor, scale: float, num_points: int, kind: str,
              return_debug: bool = False, training: bool = False):
    """Dense-grid SDF evaluation + K smallest |sdf| per sample.  main/model.py:246-355.
    Returns points (B,K,3) [scaled space], sdf (B,K,1) clamped, posenc (B,K,30).
    ``training``: the reference calls this under ``torch.no_grad()`` but with the MODULE still in train mode during a
    branch-B training step (main/model.py:462-481), so the SDF decoder's dropout (p = cfg.sdf_dropout after each of its four
    hidden layers, common/nets/sdf_net.py:112-113) is live while the points are being ranked - the selected set of a
    training step is a random function of the lattice.  ``linear_sdfin`` has no dropout."""
    B = center.shape[0]
    lattice = dense_lattice(cfg.bins_n)
    n = torch.tensor([cfg.input_img_shape[1] - 1, cfg.input_img_shape[0] - 1],
                     dtype=torch.float32) / 2
    pts_o = torch.zeros(B, num_points, 3)
    sdf_o = torch.zeros(B, num_points, 1)
    pe_o = torch.zeros(B, num_points, cfg.PointFeatSize - 3)
    dbg = []
    for b in range(B):
        keep, uv = lattice_bbox_mask(lattice, center[b], cam_intr[b], bbox[b], scale)
        uv_k = uv[keep].unsqueeze(0)
        samp = lattice[keep].clone()
        if samp.shape[0] < num_points:
            raise ValueError(
                f"sdf_infer: only {samp.shape[0]} lattice points fall inside the bbox of sample "
                f"{b}, fewer than num_points={num_points} (reference fails at main/model.py:348)")
        grid = (uv_k - n) / n
        feats = sample_pyramid({k: v[b:b + 1] for k, v in pyramid.items()}, grid,
                               cfg.mutliscale_layers)
        fea = mlp(feats, P, "linear_sdfin", 2, True).squeeze(0)
        pe = posenc(samp)
        x0 = torch.cat([fea, pe, samp], 1).contiguous()
        sdf = sdf_decoder(x0, P, f"{kind}_sdf_decoder", training, cfg.sdf_dropout).squeeze(1)
        order = torch.sort(sdf.abs())[1][:num_points]
        pts_o[b] = samp[order]
        sdf_o[b] = sdf[order].unsqueeze(-1)
        pe_o[b] = pe[order]
        if return_debug:
            dbg.append(dict(keep=keep, sdf=sdf, order=order,
                            lattice_idx=torch.nonzero(keep).squeeze(1)[order]))
    sdf_o = sdf_o.clamp(-cfg.ClampingDistance, cfg.ClampingDistance)
    if return_debug:
        return pts_o, sdf_o, pe_o, dbg
    return pts_o, sdf_o, pe_o


# --------------------------------------------------------------------------------------
# A.7 sigma gate, tokens                                 (main/model.py:123-126, 483-562)
# --------------------------------------------------------------------------------------
def sdf_activation(sdf: Tensor, beta: Tensor) -> Tensor:
    """sigma = sigmoid(sdf/beta)/beta with beta floored at 2e-3 (in place in the reference,
    main/model.py:123-126)."""
    beta.data.copy_(torch.clamp(beta.data, min=2e-3))   # .data: invisible to autograd, as in the reference
    return torch.sigmoid(sdf / beta) / beta


def token_mlp(P: Params, cfg: OracleCfg, pyramid, points, center, cam_intr, scale):
    """get_input_transformer: main/model.py:145-179 -> (feat (B,P,223), cam points (B,P,3))."""
    cam, grid = project_points(points, center, cam_intr, scale, cfg.input_img_shape)
    feats = sample_pyramid(pyramid, grid, cfg.mutliscale_layers)
    return mlp(feats, P, "linear_transformerin", 4, True), cam


# --------------------------------------------------------------------------------------
# A.8 attention stack                                    (common/nets/transformer.py)
# --------------------------------------------------------------------------------------
def mha(query: Tensor, key: Tensor, value: Tensor, P: Params, prefix: str, nheads: int,
        attn_mask: Optional[Tensor] = None, dropout_p: float = 0.0, training: bool = False):
    """nn.MultiheadAttention semantics, seq-first (L,B,E): packed in-proj, q scaled by
    1/sqrt(d_h), bool mask True = masked (-inf), fp32 softmax, dropout on probabilities,
    out-proj.  Materialises (B*h, L, S) scores like the reference."""
    L, B, E = query.shape
    S = key.shape[0]
    dh = E // nheads
    w = P[prefix + ".in_proj_weight"]
    b = P[prefix + ".in_proj_bias"]
    q = F.linear(query, w[:E], b[:E])
    k = F.linear(key, w[E:2 * E], b[E:2 * E])
    v = F.linear(value, w[2 * E:], b[2 * E:])
    q = q.reshape(L, B * nheads, dh).transpose(0, 1) * (1.0 / math.sqrt(dh))
    k = k.reshape(S, B * nheads, dh).transpose(0, 1)
    v = v.reshape(S, B * nheads, dh).transpose(0, 1)
    scores = torch.bmm(q, k.transpose(1, 2))
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask.unsqueeze(0), float("-inf"))
    prob = torch.softmax(scores, dim=-1)
    prob = F.dropout(prob, p=dropout_p, training=training)
    out = torch.bmm(prob, v).transpose(0, 1).reshape(L, B, E)
    return F.linear(out, P[prefix + ".out_proj.weight"], P[prefix + ".out_proj.bias"])


def _ln(x: Tensor, P: Params, prefix: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[prefix + ".weight"], P[prefix + ".bias"], 1e-5)


def encoder_layer(x: Tensor, P: Params, prefix: str, cfg: OracleCfg, training: bool) -> Tensor:
    """Post-norm encoder layer (pos == 0), common/nets/transformer.py:286-302; with cfg.pre_norm forward_pre, :304-321."""
    p = cfg.dropout
    if cfg.pre_norm:
        x2 = _ln(x, P, prefix + ".norm1")
        x = x + F.dropout(mha(x2, x2, x2, P, prefix + ".self_attn", cfg.nheads, None, p, training), p, training)
        x2 = _ln(x, P, prefix + ".norm2")
        h = F.relu(F.linear(x2, P[prefix + ".linear1.weight"], P[prefix + ".linear1.bias"]))
        h = F.linear(F.dropout(h, p, training), P[prefix + ".linear2.weight"], P[prefix + ".linear2.bias"])
        return x + F.dropout(h, p, training)
    a = mha(x, x, x, P, prefix + ".self_attn", cfg.nheads, None, p, training)
    x = _ln(x + F.dropout(a, p, training), P, prefix + ".norm1")
    h = F.relu(F.linear(x, P[prefix + ".linear1.weight"], P[prefix + ".linear1.bias"]))
    h = F.linear(F.dropout(h, p, training), P[prefix + ".linear2.weight"],
                 P[prefix + ".linear2.bias"])
    return _ln(x + F.dropout(h, p, training), P, prefix + ".norm2")


def encoder(src: Tensor, P: Params, prefix: str, n_layers: int, cfg: OracleCfg,
            training: bool) -> Tuple[Tensor, Tensor]:
    """Returns (memory = un-normed last activation, stacked inter_norm(x_l)).
    common/nets/transformer.py:158-202."""
    x = src
    inter = []
    for l in range(n_layers):
        x = encoder_layer(x, P, f"{prefix}.layers.{l}", cfg, training)
        inter.append(_ln(x, P, prefix + ".inter_norm"))
    if cfg.pre_norm:                    # encoder.norm exists only with normalize_before (:82-84) and closes the stack (:199-200)
        x = _ln(x, P, prefix + ".norm")
    return x, torch.stack(inter)


def decoder_layer(tgt, memory, query_pos, P, prefix, cfg: OracleCfg, tgt_mask, memory_mask,
                  training: bool) -> Tensor:
    """Post-norm decoder layer, common/nets/transformer.py:366-395 (pos == 0); with cfg.pre_norm forward_pre, :397-424."""
    p = cfg.dropout
    if cfg.pre_norm:
        t2 = _ln(tgt, P, prefix + ".norm1")
        qk = t2 + query_pos
        tgt = tgt + F.dropout(mha(qk, qk, t2, P, prefix + ".self_attn", cfg.nheads, tgt_mask, p, training), p, training)
        t2 = _ln(tgt, P, prefix + ".norm2")
        tgt = tgt + F.dropout(mha(t2 + query_pos, memory, memory, P, prefix + ".multihead_attn", cfg.nheads, memory_mask, p, training),
                              p, training)
        t2 = _ln(tgt, P, prefix + ".norm3")
        h = F.relu(F.linear(t2, P[prefix + ".linear1.weight"], P[prefix + ".linear1.bias"]))
        h = F.linear(F.dropout(h, p, training), P[prefix + ".linear2.weight"], P[prefix + ".linear2.bias"])
        return tgt + F.dropout(h, p, training)
    qk = tgt + query_pos
    a = mha(qk, qk, tgt, P, prefix + ".self_attn", cfg.nheads, tgt_mask, p, training)
    tgt = _ln(tgt + F.dropout(a, p, training), P, prefix + ".norm1")
    a = mha(tgt + query_pos, memory, memory, P, prefix + ".multihead_attn", cfg.nheads,
            memory_mask, p, training)
    tgt = _ln(tgt + F.dropout(a, p, training), P, prefix + ".norm2")
    h = F.relu(F.linear(tgt, P[prefix + ".linear1.weight"], P[prefix + ".linear1.bias"]))
    h = F.linear(F.dropout(h, p, training), P[prefix + ".linear2.weight"],
                 P[prefix + ".linear2.bias"])
    return _ln(tgt + F.dropout(h, p, training), P, prefix + ".norm3")


def decoder(memory: Tensor, query_embed: Tensor, P: Params, prefix: str, n_layers: int,
            cfg: OracleCfg, tgt_mask, memory_mask, training: bool) -> Tensor:
    """tgt = 0, query_pos = learned embedding; decoder.norm on every layer output.
    common/nets/transformer.py:205-254, 135-153."""
    B = memory.shape[1]
    qpos = query_embed.unsqueeze(1).repeat(1, B, 1)
    x = torch.zeros_like(qpos)
    outs = []
    for l in range(n_layers):
        x = decoder_layer(x, memory, qpos, P, f"{prefix}.layers.{l}", cfg, tgt_mask,
                          memory_mask, training)
        outs.append(_ln(x, P, prefix + ".norm"))
    return torch.stack(outs)


def mano_tgt_mask(nq: int = 17, shape_idx: int = 16) -> Tensor:
    """common/utils/misc.py:11-31 (True = masked)."""
    m = torch.ones(nq, nq, dtype=torch.bool)
    m[0, 0] = False
    for i in range(5):
        s, e = 3 * i + 1, 3 * i + 4
        m[s:e, s:e] = False
    m[shape_idx, shape_idx] = False
    return m


def memory_mask(nq: int, n_hand: int, n_obj: int) -> Tensor:
    """common/utils/misc.py:34-47: keys >= n_hand are masked for every query."""
    m = torch.zeros(nq, n_hand + n_obj, dtype=torch.bool)
    m[:, n_hand:] = True
    return m


# --------------------------------------------------------------------------------------
# A.9 vote aggregation + losses                          (common/nets/loss.py:23-78)
# --------------------------------------------------------------------------------------
def joint_vote(hand_points: Tensor, hand_off: Tensor, hand_cls: Tensor, joint_gt: Tensor,
               cls_dist: float):
    """hand_points (B,P,3) m; hand_off (L,P,B,60); hand_cls (L,P,B,20); joint_gt (B,20,3) mm.
    Returns loss_joint_3d, loss_joint_cls, loss_all_joint_3d, joints (L,B,20,3)."""
    L, Np, B, J = hand_cls.shape
    vote = hand_points[None, :, :, None, :] + hand_off.reshape(L, Np, B, J, 3).permute(0, 2, 1, 3, 4)
    gt_m = joint_gt / 1000
    near = ((hand_points[:, :, None, :] - gt_m[:, None]).norm(dim=-1) < cls_dist).float()
    tgt = joint_gt[None, :, None].expand(L, B, Np, J, 3)
    l3d = F.smooth_l1_loss(vote * 1000, tgt, reduction="none") * near[None, ..., None]
    l3d = l3d.sum((1, 2, 3)) / near.sum()
    lcls = F.binary_cross_entropy_with_logits(hand_cls.permute(0, 2, 1, 3),
                                              near[None].expand(L, B, Np, J))
    w = torch.softmax(hand_cls, dim=1).permute(0, 2, 1, 3).unsqueeze(-1)
    joints = (vote * w).sum(dim=2)
    lall = F.smooth_l1_loss(joints * 1000, joint_gt[None].expand(L, B, J, 3))
    return l3d.mean(), lcls, lall, joints


# --------------------------------------------------------------------------------------
# a16 MANO head conversions                              (common/nets/mano_head.py:54-278)
# --------------------------------------------------------------------------------------
def rot6d_to_mat(x: Tensor) -> Tensor:
    """Gram-Schmidt; columns are b1,b2,b3.  mano_head.py:185-194."""
    a1, a2 = x[:, 0:3], x[:, 3:6]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def mat_to_quat(R: Tensor, eps: float = 1e-6) -> Tensor:
    """Branch-on-trace conversion operating on R^T, as mano_head.py:90-182 does."""
    T = R.transpose(1, 2)
    t00, t11, t22 = T[:, 0, 0], T[:, 1, 1], T[:, 2, 2]
    d2 = t22 < eps
    d01 = t00 > t11
    d0n1 = t00 < -t11
    tr0 = 1 + t00 - t11 - t22
    tr1 = 1 - t00 + t11 - t22
    tr2 = 1 - t00 - t11 + t22
    tr3 = 1 + t00 + t11 + t22
    q0 = torch.stack([T[:, 1, 2] - T[:, 2, 1], tr0, T[:, 0, 1] + T[:, 1, 0], T[:, 2, 0] + T[:, 0, 2]], -1)
    q1 = torch.stack([T[:, 2, 0] - T[:, 0, 2], T[:, 0, 1] + T[:, 1, 0], tr1, T[:, 1, 2] + T[:, 2, 1]], -1)
    q2 = torch.stack([T[:, 0, 1] - T[:, 1, 0], T[:, 2, 0] + T[:, 0, 2], T[:, 1, 2] + T[:, 2, 1], tr2], -1)
    q3 = torch.stack([tr3, T[:, 1, 2] - T[:, 2, 1], T[:, 2, 0] - T[:, 0, 2], T[:, 0, 1] - T[:, 1, 0]], -1)
    c0 = (d2 & d01).float()[:, None]
    c1 = (d2 & ~d01).float()[:, None]
    c2 = (~d2 & d0n1).float()[:, None]
    c3 = (~d2 & ~d0n1).float()[:, None]
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(tr0[:, None] * c0 + tr1[:, None] * c1 + tr2[:, None] * c2 + tr3[:, None] * c3)
    return q * 0.5


def quat_to_aa(q: Tensor) -> Tensor:
    """mano_head.py:54-87."""
    v = q[..., 1:]
    s2 = (v * v).sum(-1)
    s = torch.sqrt(s2)
    c = q[..., 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, 2.0 * torch.ones_like(s))
    return v * k[..., None]


def mat_to_aa(R: Tensor) -> Tensor:
    aa = quat_to_aa(mat_to_quat(R))
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)


def rodrigues_via_quat(theta: Tensor) -> Tensor:
    """mano_head.py:12-52 (axis-angle -> rotation matrix through a unit quaternion)."""
    ang = (theta + 1e-8).norm(dim=1, keepdim=True)
    nrm = theta / ang
    half = ang * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * nrm], dim=1)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                     2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
                     2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], 1)
    return R.view(-1, 3, 3)


def mano_head(pose6d: Tensor, shape: Tensor, mano_layer: Callable, hands_mean: Tensor,
              mano_params: Optional[Tensor] = None):
    """pose6d (L,16,B,6), shape (L,B,10) -> dicts like mano_head.py:232-278.
    ``mano_layer(pose48, betas10) -> (verts_mm, joints_mm)``."""
    L, N, B, C = pose6d.shape
    R = rot6d_to_mat(pose6d.permute(0, 2, 1, 3).reshape(L * B * N, C))
    pose = mat_to_aa(R).reshape(-1, 48)
    betas = shape.reshape(-1, 10)
    verts, joints = mano_layer(pose, betas)
    pred = dict(verts3d=verts.view(L, B, -1, 3) / 1000, joints3d=joints.view(L, B, -1, 3) / 1000,
                mano_pose=R.view(L, B, N, 3, 3), mano_shape=betas.view(L, B, 10))
    gt = None
    if mano_params is not None:
        gt_shape = mano_params[:, 48:]
        gt_pose = mano_params[:, :48].clone()
        gt_pose[:, 3:] = gt_pose[:, 3:] - hands_mean
        gv, gj = mano_layer(gt_pose, gt_shape)
        gt = dict(verts3d=gv / 1000, joints3d=gj / 1000, mano_shape=gt_shape,
                  mano_pose=rodrigues_via_quat(gt_pose.reshape(-1, 3)).view(-1, 16, 3, 3))
    return pred, gt


# --------------------------------------------------------------------------------------
# f4 encoder-side auxiliary image losses                (main/model.py:128-143, :404-422)
# --------------------------------------------------------------------------------------
def render_gaussian_heatmap(joint_coord: Tensor, hm_hw=(128, 128), sigma: float = 2.5) -> Tensor:
    """main/model.py:128-143: 255 * sum_j exp(-((x - jx) / sigma)^2 / 2 - ((y - jy) / sigma)^2 / 2) on the
    cfg.output_hm_shape grid (main/config.py: output_hm_shape (64, 128, 128), sigma 2.5)."""
    y = torch.arange(hm_hw[0], dtype=joint_coord.dtype)
    x = torch.arange(hm_hw[1], dtype=joint_coord.dtype)
    yy, xx = torch.meshgrid(y, x, indexing="ij")
    jx, jy = joint_coord[:, :, 0, None, None], joint_coord[:, :, 1, None, None]
    hm = torch.exp(-(((xx[None, None] - jx) / sigma) ** 2) / 2 - (((yy[None, None] - jy) / sigma) ** 2) / 2)
    return hm.sum(1) * 255


def aux_image_losses(decoder_out: Tensor, targets: Dict[str, Tensor], sigma: float = 2.5) -> Dict[str, Tensor]:
    """main/model.py:404-422: joint_heatmap = (decoder_out[:, 0] - heatmap)^2 (common/nets/loss.py:14-20), obj_seg / hand_seg =
    BCELoss(reduction="none") of channels 2 / 1 against the masks (main/model.py:94-95)."""
    hm = render_gaussian_heatmap(targets["joint_coord"], decoder_out.shape[-2:], sigma)
    return {"joint_heatmap": (decoder_out[:, 0] - hm) ** 2,
            "obj_seg": F.binary_cross_entropy(decoder_out[:, 2], targets["obj_seg"], reduction="none"),
            "hand_seg": F.binary_cross_entropy(decoder_out[:, 1], targets["hand_seg"], reduction="none"),
            "heatmap": hm}


# --------------------------------------------------------------------------------------
# a1 the whole hot path                                  (main/model.py:370-665)
# --------------------------------------------------------------------------------------
def hot_path_forward(P: Params, cfg: OracleCfg, feature_pyramid: Dict[str, Tensor],
                     inputs: Dict[str, Tensor], targets: Dict[str, Tensor],
                     meta_info: Dict[str, Tensor], mode: str, epoch_cnt: float = 1e8,
                     batch_ratio: float = 0.0, mano_layer: Optional[Callable] = None,
                     hands_mean: Optional[Tensor] = None, rng: Optional[random.Random] = None,
                     return_internals: bool = False) -> Dict[str, Tensor]:
    """Everything ``Model.forward`` does after ``decoder_net`` except the encoder-side aux
    losses (heat-map / segmentation, main/model.py:404-422).  Keys follow the reference's
    ``_out`` convention."""
    training = mode == "train"
    loss: Dict[str, Tensor] = {}
    out: Dict[str, Tensor] = {}
    _sdf_forward = globals()["sdf_forward"]

    def sdf_forward(*a, **k):            # Model.forward drops the class logits of cfg.ClassifierBranch (main/model.py:376,445,499)
        return _sdf_forward(*a, **k)[:2]
    mano_root = meta_info["mano_root"]
    obj_center = meta_info["obj_center_cam"]
    K = meta_info["cam_intr"]
    hs, os_ = cfg.hand_sdf_scale, cfg.obj_sdf_scale
    cd = cfg.ClampingDistance
    nh, no = cfg.num_samp_hand, cfg.num_samp_obj

    if training or cfg.dataset == "dexycb":                                   # :370-402
        sdf_h, _ = sdf_forward(P, cfg, feature_pyramid, inputs["hand_sdf_points"], mano_root, K,
                               hs, "hand", training)
        sdf_o, _ = sdf_forward(P, cfg, feature_pyramid, inputs["obj_sdf_points"], obj_center, K,
                               os_, "obj", training)
        loss["sdfhand_loss"] = F.l1_loss(sdf_h, targets["hand_sdf"].clamp(-cd, cd).unsqueeze(-1))
        loss["sdfobj_loss"] = F.l1_loss(sdf_o, targets["obj_sdf"].clamp(-cd, cd).unsqueeze(-1))

    p = (rng or random).uniform(0, 1)                                          # :426
    if (p < 0.4 or epoch_cnt < cfg.point_sampling_epoch) and training:         # :427-460
        d = cfg.random_move_dist[len([a for a in cfg.random_ratio if batch_ratio > a])]
        hand_points = inputs["hand_pre_points"] + torch.empty_like(inputs["hand_pre_points"]).uniform_(-d, d)
        obj_points = inputs["obj_pre_points"] + torch.empty_like(inputs["obj_pre_points"]).uniform_(-d, d)
        hand_sdf, hand_pe = sdf_forward(P, cfg, feature_pyramid, hand_points, mano_root, K, hs, "hand", training)
        obj_sdf, obj_pe = sdf_forward(P, cfg, feature_pyramid, obj_points, obj_center, K, os_, "obj", training)
    else:                                                                      # :462-481
        with torch.no_grad():
            hand_points, hand_sdf, hand_pe = sdf_infer(P, cfg, feature_pyramid, mano_root, K,
                                                       meta_info["bbox_hand"], hs, nh, "hand", training=training)
            obj_points, obj_sdf, obj_pe = sdf_infer(P, cfg, feature_pyramid, obj_center, K,
                                                    meta_info["bbox_obj"], os_, no, "obj", training=training)

    sig_h = sdf_activation(hand_sdf.detach(), P["hand_sigmoid_beta"])         # :483-484
    sig_o = sdf_activation(obj_sdf.detach(), P["obj_sigmoid_beta"])
    hand_fea, hand_cam = token_mlp(P, cfg, feature_pyramid, hand_points, mano_root, K, hs)
    hand_rel = hand_cam - mano_root[:, None, :]
    obj_fea, obj_cam = token_mlp(P, cfg, feature_pyramid, obj_points, obj_center, K, os_)
    obj_rel = obj_cam - obj_center[:, None, :]

    hand_o_pts = (hand_cam - obj_center[:, None, :]) * os_                     # :495-518
    hand_o_rel = hand_cam - obj_center[:, None, :]
    hand_o_sdf, hand_o_pe = sdf_forward(P, cfg, feature_pyramid, hand_o_pts, obj_center, K, os_, "obj", training)
    obj_h_pts = (obj_cam - mano_root[:, None, :]) * hs
    obj_h_rel = obj_cam - mano_root[:, None, :]
    obj_h_sdf, obj_h_pe = sdf_forward(P, cfg, feature_pyramid, obj_h_pts, mano_root, K, hs, "hand", training)
    sig_h_o = sdf_activation(hand_o_sdf.detach(), P["obj_sigmoid_beta"])
    sig_o_h = sdf_activation(obj_h_sdf.detach(), P["hand_sigmoid_beta"])

    def tok(rel, pe, fea, sig):
        return torch.cat([rel, pe, fea * sig], dim=2).permute(1, 0, 2).contiguous()

    hand_in = torch.cat([tok(hand_rel, hand_pe, hand_fea, sig_h),              # :520-562
                         tok(obj_h_rel, obj_h_pe, obj_fea, sig_o_h).detach()], dim=0)
    obj_in = torch.cat([tok(obj_rel, obj_pe, obj_fea, sig_o),
                        tok(hand_o_rel, hand_o_pe, hand_fea, sig_h_o).detach()], dim=0)

    if cfg.use_inverse_kinematics:                                             # :564-569
        tmask = None
        mmask = memory_mask(1, nh, no)
    else:
        tmask = mano_tgt_mask(cfg.mano_num_queries, cfg.mano_shape_indx)
        mmask = memory_mask(cfg.mano_num_queries, nh, no)

    memory, hand_enc = encoder(hand_in, P, "hand_transformer.encoder", cfg.enc_layers, cfg, training)
    hs_out = decoder(memory, P["mano_query_embed.weight"], P, "hand_transformer.decoder",
                     cfg.dec_layers, cfg, tmask, mmask, training)
    _, obj_enc = encoder(obj_in, P, "obj_transformer.encoder", cfg.enc_layers // 2, cfg, training)

    hand_off = mlp(hand_enc[:, :nh], P, "linear_handvote", 4, False)          # :587-593
    hand_cls = mlp(hand_enc[:, :nh], P, "linear_handcls", 3, False)
    obj_rot = mlp(obj_enc[:, :no], P, "linear_obj_rot", 3, False)
    obj_trans = mlp(obj_enc[:, :no], P, "linear_obj_rel_trans", 3, False)

    pred_m = gt_m = pose6d = None
    if cfg.use_inverse_kinematics:                                             # :595-597
        mano_shape = mlp(hs_out[:, 0], P, "linear_shape", 3, False)
        out["mano_shape_out"] = mano_shape[-1]
    else:                                                                      # :599-620
        pose6d = mlp(hs_out[:, :cfg.mano_shape_indx], P, "linear_pose", 3, False)
        mano_shape = mlp(hs_out[:, cfg.mano_shape_indx], P, "linear_shape", 3, False)
        if mano_layer is not None:
            mp = targets["mano_param"] if (training or cfg.dataset == "dexycb") else None
            pred_m, gt_m = mano_head(pose6d, mano_shape, mano_layer, hands_mean, mp)
            out["mano_mesh_out"] = pred_m["verts3d"][-1]
            out["mano_joints_out"] = pred_m["joints3d"][-1]
            if cfg.dataset == "dexycb":
                out["mano_joints_gt_out"] = gt_m["joints3d"]
                out["mano_mesh_gt_out"] = gt_m["verts3d"]

    if not training:                                                           # :622-624
        out["obj_rot_out"] = obj_rot[-1].permute(1, 0, 2).contiguous()
        out["obj_trans_out"] = obj_trans[-1].permute(1, 0, 2).contiguous()

    if training or cfg.dataset == "dexycb":                                    # :626-638
        joints_gt = targets["joint_cam_no_trans"][:, 1:]
    else:
        joints_gt = torch.zeros(mano_root.shape[0], 20, 3)
    (loss["loss_joint_3d"], loss["loss_joint_cls"], loss["loss_all_joint_3d"],
     joints) = joint_vote(hand_rel, hand_off, hand_cls, joints_gt, cfg.hand_cls_dist)
    out["hand_joints_out"] = joints[-1]

    if training or cfg.dataset == "dexycb":                                    # :640-654
        if cfg.use_inverse_kinematics:
            gt_shape = targets["mano_param"][:, -10:]
            loss["shape_param_loss"] = cfg.lambda_manoshape * F.mse_loss(
                mano_shape, gt_shape.unsqueeze(0).expand(mano_shape.shape))
            loss["shape_reg_loss"] = cfg.mano_lambda_regulshape * F.mse_loss(
                mano_shape, torch.zeros_like(mano_shape))
        elif pred_m is not None:
            def _mse(a, b):
                return F.mse_loss(a, b.unsqueeze(0).expand(a.shape))
            loss["mano_mesh_loss"] = cfg.lambda_verts3d * _mse(pred_m["verts3d"], gt_m["verts3d"])
            loss["mano_joint_loss"] = cfg.lambda_joints3d * _mse(pred_m["joints3d"], gt_m["joints3d"])
            loss["pose_param_loss"] = cfg.lambda_manopose * _mse(pred_m["mano_pose"], gt_m["mano_pose"])
            loss["shape_param_loss"] = cfg.lambda_manoshape * _mse(pred_m["mano_shape"], gt_m["mano_shape"])

    loss["obj_rot"] = F.smooth_l1_loss(obj_rot, targets["obj_rot"][None, None].expand_as(obj_rot))
    loss["obj_trans"] = F.smooth_l1_loss(obj_trans, targets["rel_obj_trans"][None, None].expand_as(obj_trans))

    res = {**loss, **out}
    if return_internals:
        res["_internals"] = dict(hand_points=hand_points, obj_points=obj_points, hand_sdf=hand_sdf,
                                 obj_sdf=obj_sdf, hand_in=hand_in, obj_in=obj_in, hand_enc=hand_enc,
                                 obj_enc=obj_enc, hs=hs_out, memory=memory,
                                 pose6d=pose6d, mano_shape=mano_shape, obj_rot=obj_rot,
                                 obj_trans=obj_trans, hand_off=hand_off, hand_cls=hand_cls)
    return res
