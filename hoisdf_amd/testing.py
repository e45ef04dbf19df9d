"""Deterministic synthetic weights / inputs shared by tests, the golden-vector generator,
``bench.py`` and ``__graft_entry__.smoke()``.

Weights are a pure function of (parameter name, shape, seed) through numpy's PCG64 - they
are never stored in fixtures, both sides of every parity test regenerate them.  Input
distributions follow SURVEY.md section 8(d) (DexYCB / HO3D shaped synthetic batches).
No oracle import here (this module is part of the shipped package).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

PYRAMID_SMALL = OrderedDict(stride2=(32, 128), stride4=(64, 64), stride8=(128, 32),
                            stride16=(256, 16), stride32=(512, 8))
PYRAMID_BIG = OrderedDict(stride2=(128, 128), stride4=(256, 64), stride8=(512, 32),
                          stride16=(1024, 16), stride32=(2048, 8))


def hot_path_param_shapes(C: int = 992, ik: bool = False, hidden: int = 256,
                          enc_layers: int = 6, dec_layers: int = 4,
                          ffn: int = 1024, pre_norm: bool = False,
                          classifier: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """State-dict schema of the hot path (reference
    main/model.py:49-90, common/nets/sdf_net.py:50-62, common/nets/transformer.py)."""
    S: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    D = hidden
    S["hand_sigmoid_beta"] = (1,)
    S["obj_sigmoid_beta"] = (1,)
    S["norm1.weight"] = (C,)
    S["norm1.bias"] = (C,)

    def mlp(prefix, dims):
        for i in range(len(dims) - 1):
            S[f"{prefix}.layers.{i}.weight"] = (dims[i + 1], dims[i])
            S[f"{prefix}.layers.{i}.bias"] = (dims[i + 1],)

    mlp("linear_transformerin", [C, 1024, 512, 256, D - 33])
    mlp("linear_sdfin", [C, 512, D])
    for kind in ("hand", "obj"):
        dims = [(512, D + 33), (D - 33, 512), (512, 512), (512, 512)]
        for i, (o, n) in enumerate(dims):
            S[f"{kind}_sdf_decoder.linh{i}.bias"] = (o,)
            S[f"{kind}_sdf_decoder.linh{i}.weight_g"] = (o, 1)
            S[f"{kind}_sdf_decoder.linh{i}.weight_v"] = (o, n)
        if classifier:                      # common/nets/sdf_net.py:73-75 (cfg.ClassifierBranch)
            S[f"{kind}_sdf_decoder.classifier_head.weight"] = (6, 512)
            S[f"{kind}_sdf_decoder.classifier_head.bias"] = (6,)
        S[f"{kind}_sdf_decoder.linh4.weight"] = (1, 512)
        S[f"{kind}_sdf_decoder.linh4.bias"] = (1,)

    def attn(prefix):
        S[prefix + ".in_proj_weight"] = (3 * D, D)
        S[prefix + ".in_proj_bias"] = (3 * D,)
        S[prefix + ".out_proj.weight"] = (D, D)
        S[prefix + ".out_proj.bias"] = (D,)

    def ln(prefix):
        S[prefix + ".weight"] = (D,)
        S[prefix + ".bias"] = (D,)

    def enc(prefix, n):
        for l in range(n):
            p = f"{prefix}.layers.{l}"
            attn(p + ".self_attn")
            S[p + ".linear1.weight"] = (ffn, D)
            S[p + ".linear1.bias"] = (ffn,)
            S[p + ".linear2.weight"] = (D, ffn)
            S[p + ".linear2.bias"] = (D,)
            ln(p + ".norm1")
            ln(p + ".norm2")
        if pre_norm:                        # common/nets/transformer.py:33,87: encoder.norm only with normalize_before
            ln(prefix + ".norm")
        ln(prefix + ".inter_norm")

    enc("hand_transformer.encoder", enc_layers)
    for l in range(dec_layers):
        p = f"hand_transformer.decoder.layers.{l}"
        attn(p + ".self_attn")
        attn(p + ".multihead_attn")
        S[p + ".linear1.weight"] = (ffn, D)
        S[p + ".linear1.bias"] = (ffn,)
        S[p + ".linear2.weight"] = (D, ffn)
        S[p + ".linear2.bias"] = (D,)
        ln(p + ".norm1")
        ln(p + ".norm2")
        ln(p + ".norm3")
    ln("hand_transformer.decoder.norm")
    enc("obj_transformer.encoder", enc_layers // 2)

    S["mano_query_embed.weight"] = (1 if ik else 17, D)
    if not ik:
        mlp("linear_pose", [D, D, D, 6])
    mlp("linear_shape", [D, D, D, 10])
    mlp("linear_handvote", [D, D, D, D, 60])
    mlp("linear_handcls", [D, D, D, 20])
    mlp("linear_objvote", [D, D, D, D, 24])
    mlp("linear_objcls", [D, D, D, 8])
    mlp("linear_obj_rel_trans", [D, D, D, 3])
    mlp("linear_obj_rot", [D, D, D, 3])
    return S


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def det_param(name: str, shape, seed: int = 0) -> torch.Tensor:
    """One deterministic parameter.  Scales keep activations O(1) through the stack and the
    SDF head un-saturated (|sdf| mostly inside the +-0.15 clamp with some values outside)."""
    r = _rng(name, seed)
    shape = tuple(shape)
    if name.endswith("sigmoid_beta"):
        v = np.array([0.08 if name.startswith("hand") else 0.12], np.float32)
    elif name.endswith("weight_g"):
        v = (0.8 + 0.4 * r.random(shape)).astype(np.float32)
    elif name.endswith("weight_v"):
        v = r.standard_normal(shape).astype(np.float32) / np.sqrt(shape[1])
    elif name.endswith("linh4.weight"):     # zero-sum so the SDF head is roughly centred
        v = r.standard_normal(shape)
        v = (v - v.mean()) / np.sqrt(shape[1])
    elif name.endswith("linh4.bias"):
        v = np.array([-0.15 if name.startswith("hand") else -0.18])
    elif "norm" in name and name.endswith(".weight"):
        v = (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
    elif name.endswith("mano_query_embed.weight"):
        v = (0.5 * r.standard_normal(shape)).astype(np.float32)
    elif len(shape) == 2:
        gain = 1.4 if ("layers" in name or "linear1" in name) else 1.0
        v = r.standard_normal(shape).astype(np.float32) * (gain / np.sqrt(shape[1]))
    else:
        v = (0.05 * r.standard_normal(shape)).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float32)))


def det_params(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, det_param(k, s, seed)) for k, s in shapes.items())


OUTLIER_CHANNELS = (3, 17, 40)          # (modulo a level's channel count)


def synthetic_pyramid(B: int, big: bool = False, seed: int = 0, scale: float = 1.0,
                      nonneg: bool = True, outliers: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """Random NCHW feature maps shaped like the CNN decoder's pyramid (post-ReLU => >= 0).
    outliers != 1: channels OUTLIER_CHANNELS of every level are that many times louder (trained CNN features have such channels;
    the "_smallbeta" fixtures use x 100)."""
    spec = PYRAMID_BIG if big else PYRAMID_SMALL
    out = OrderedDict()
    for name, (c, hw) in spec.items():
        a = _rng("pyr." + name, seed).standard_normal((B, c, hw, hw)).astype(np.float32) * scale
        if nonneg:
            a = np.maximum(a, 0)
        if outliers != 1.0:
            a[:, [ch % c for ch in OUTLIER_CHANNELS]] *= np.float32(outliers)
        out[name] = torch.from_numpy(a)
    return out


SMALL_BETA = {"hand_sigmoid_beta": 2e-3, "obj_sigmoid_beta": 1e-2}     # a trained model's gates: sigma up to 500 (main/model.py:123-126)
# "_trainedlike" fixtures: SMALL_BETA + outlier channels + the FIRST encoder layer's query / key projections scaled down to the token
# magnitudes they then see (sigma-gated rows up to ~5e3 / 2.4e4), as a network trained on such inputs would have them: attention scores
# of O(10-100).  With det_param's unit-gain q / k weights the same tokens give scores of 7e6 (hand) / 2.4e8 (object) in the log2 domain:
# one fp32 ulp of such a score is 0.5 / 16 - softmax is then decided by rounding, in any fp32 implementation
TRAINED_LIKE_QK = {"hand_transformer.encoder.layers.0.self_attn": 3e-3, "obj_transformer.encoder.layers.0.self_attn": 7e-4}


def apply_trained_like(get, E: int = 256):
    """scale rows [0, 2E) (q and k) of the two first-layer in-projections in place; ``get(name)`` returns the tensor of a parameter name"""
    with torch.no_grad():
        for prefix, f in TRAINED_LIKE_QK.items():
            get(prefix + ".in_proj_weight")[:2 * E].mul_(f)
            get(prefix + ".in_proj_bias")[:2 * E].mul_(f)


def synthetic_batch(B: int, n_hand: int, n_obj: int, seed: int = 1234):
    """(inputs, targets, meta_info) with the dataset schema of data/dexycb.py:627-655 and the
    distributions of SURVEY.md section 8(d).  CPU float32 tensors."""
    r = _rng("batch", seed)

    def U(lo, hi, *s):
        return torch.from_numpy((lo + (hi - lo) * r.random(s)).astype(np.float32))

    def N(std, *s):
        return torch.from_numpy((std * r.standard_normal(s)).astype(np.float32))

    K = torch.tensor([[600.0, 0, 128], [0, 600.0, 128], [0, 0, 1]]).repeat(B, 1, 1)
    inputs = dict(
        img=U(0, 1, B, 3, 256, 256),
        hand_sdf_points=U(-1, 1, B, n_hand, 3), obj_sdf_points=U(-1, 1, B, n_obj, 3),
        hand_pre_points=U(-0.3, 0.3, B, n_hand, 3), obj_pre_points=U(-0.3, 0.3, B, n_obj, 3))
    targets = dict(
        hand_sdf=U(0, 0.1, B, n_hand), obj_sdf=U(0, 0.1, B, n_obj),
        joint_cam_no_trans=N(50.0, B, 21, 3), mano_param=N(0.1, B, 58), obj_rot=N(1.0, B, 3),
        rel_obj_trans=N(0.05, B, 3),
        hand_seg=(U(0, 1, B, 128, 128) > 0.5).float(), obj_seg=(U(0, 1, B, 128, 128) > 0.5).float(),
        joint_coord=U(20, 108, B, 21, 2))
    meta = dict(
        mano_root=torch.tensor([0.0, 0.0, 0.7]).repeat(B, 1) + N(0.01, B, 3),
        obj_center_cam=torch.tensor([0.03, 0.02, 0.72]).repeat(B, 1) + N(0.01, B, 3),
        cam_intr=K,
        bbox_hand=torch.tensor([40.0, 40, 220, 220]).repeat(B, 1),
        bbox_obj=torch.tensor([60.0, 60, 200, 200]).repeat(B, 1))
    return inputs, targets, meta


def synthetic_decoder_out(B: int, seed: int = 13):
    """a seeded stand-in for decoder_net's second output (B, 3, 128, 128): channel 0 = heat-map logits on the scale of the
    255-peaked target, channels 1 / 2 = segmentation probabilities in (0.01, 0.99) with a few saturated pixels (BCELoss
    clamps its logs at -100)."""
    r = _rng("decoder_out", seed)
    d = torch.from_numpy((0.01 + 0.98 * r.random((B, 3, 128, 128))).astype(np.float32))
    d[:, 0] = d[:, 0] * 300.0
    d[0, 1, 0, 0], d[0, 2, 0, 1], d[-1, 1, 5, 7], d[-1, 2, 9, 3] = 0.0, 1.0, 1.0, 0.0
    return d


def synthetic_sdf_frames(n_frames: int, seed: int = 14):
    """``sdf_processed``-shaped frames ((N_h + N_o, 6) float32 rows [x y z sdf_hand sdf_obj label], tool/pre_process_sdf.py:
    140-148) + their ``sdf_index`` rows [N_h, N_o]; about a third of the rows pass the |sdf| < 0.05 pre-filter."""
    r = _rng("sdf_frames", seed)
    frames, index = [], []
    for _ in range(n_frames):
        nh, no = int(r.integers(300, 500)), int(r.integers(200, 400))
        a = np.zeros((nh + no, 6), np.float32)
        a[:, :3] = (r.uniform(-0.1, 0.1, (nh + no, 3)) + np.array([0.0, 0.0, 0.7])).astype(np.float32)
        a[:, 3] = r.uniform(-0.15, 0.15, nh + no)
        a[:, 4] = r.uniform(-0.15, 0.15, nh + no)
        a[:, 5] = r.integers(0, 6, nh + no)
        frames.append(a)
        index.append([nh, no])
    return frames, np.asarray(index)


def to_device(tree, device):
    if isinstance(tree, dict):
        return type(tree)((k, to_device(v, device)) for k, v in tree.items())
    if isinstance(tree, (list, tuple)):
        return type(tree)(to_device(v, device) for v in tree)
    if torch.is_tensor(tree):
        return tree.to(device)
    return tree
