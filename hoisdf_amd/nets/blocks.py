"""Network blocks of the hot path as thin ``nn.Module`` parameter holders whose ``forward``
runs on the HIP kernels (hoisdf_amd.ops).  Parameter names and shapes equal the reference's
state dict so checkpoints are drop-in:
  MLP              common/nets/layer.py:168-201      -> ``layers.{i}.weight/bias``
  SDFDecoder       common/nets/sdf_net.py:12-122     -> ``linh{0..3}.weight_g/weight_v/bias``, ``linh4.*``
  Transformer /    common/nets/transformer.py:15-459 -> ``encoder.layers.{i}.self_attn.in_proj_*`` ...
  VoteTransformer
Internally tokens are batch-first (B, S, 256); the seq-first public ``forward`` of the
transformers keeps the reference's call signature and return layout.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

import os

from .. import ops

# one autograd node per encoder layer (ops.encoder_layer); HOISDF_FUSED_LAYERS=0 keeps the op-by-op graph (A/B, debugging)
FUSED_LAYER_NODES = os.environ.get("HOISDF_FUSED_LAYERS", "1") != "0"
INTER_ROWS = os.environ.get("HOISDF_INTER_ROWS", "1") != "0"        # A/B switch: inter_norm on the kept rows only


class MLP(nn.Module):
    """Linear(+ReLU) chain; last ReLU iff ``is_activation_last``."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, is_activation_last=False):
        super().__init__()
        h = hidden_dim if isinstance(hidden_dim, list) else [hidden_dim] * (num_layers - 1)
        assert len(h) == num_layers - 1, "len(hidden_dim) != num_layers-1"
        dims = [input_dim] + h + [output_dim]
        self.num_layers = num_layers
        self.is_activation_last = is_activation_last
        self.layers = nn.ModuleList(nn.Linear(dims[i], dims[i + 1]) for i in range(num_layers))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            act = i < self.num_layers - 1 or self.is_activation_last
            x = ops.linear(x, layer.weight, layer.bias, act=act)
        return x


class _WNLinear(nn.Module):
    """Parameters of a weight-normed nn.Linear in the legacy ``weight_g / weight_v`` naming the
    reference's checkpoints use (torch.nn.utils.weight_norm, dim=0)."""

    def __init__(self, in_f, out_f):
        super().__init__()
        lin = nn.Linear(in_f, out_f)
        self.bias = nn.Parameter(lin.bias.detach().clone())
        self.weight_g = nn.Parameter(lin.weight.detach().norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(lin.weight.detach().clone())

    def effective_weight(self):
        return ops.weight_norm(self.weight_v, self.weight_g)


class SDFDecoder(nn.Module):
    """x0 = [feat256 | posenc30 | xyz3] -> 512 -> 223 (+x0) -> 512 -> 512 -> 1, tanh.
    Layers 0-3 weight-normed, ReLU + dropout(0.2) in training; skip-concat before layer 2."""

    def __init__(self, latent_size, point_feat_size, dims=(512, 512, 512, 512), dropout_prob=0.2,
                 use_classifier=False, num_class=6):
        super().__init__()
        d0 = latent_size + point_feat_size
        self.in_dim = d0
        self.linh0 = _WNLinear(d0, dims[0])
        self.linh1 = _WNLinear(dims[0], dims[1] - d0)
        self.linh2 = _WNLinear(dims[1], dims[2])
        self.linh3 = _WNLinear(dims[2], dims[3])
        self.use_classifier = bool(use_classifier)
        # registration order = the reference's (common/nets/sdf_net.py:58-75: linh4 inside the layer loop, classifier_head after it):
        # parameters() order is what optimizer state in a checkpoint is indexed by
        self.linh4 = nn.Linear(dims[3], 1)
        if self.use_classifier:              # cfg.ClassifierBranch (common/nets/sdf_net.py:73-75): 6 class logits from the last hidden layer
            self.classifier_head = nn.Linear(dims[3], num_class)
        self.dropout_prob = dropout_prob

    def hidden(self, x0):
        """all layers up to the 512-wide h3 (input of the scalar head)"""
        p = self.dropout_prob if self.training else 0.0
        h0 = ops.linear(x0, self.linh0.effective_weight(), self.linh0.bias, act=True, drop_p=p)
        h1 = ops.linear(h0, self.linh1.effective_weight(), self.linh1.bias, act=True, drop_p=p)
        h2 = ops.linear(torch.cat([h1, x0], dim=1), self.linh2.effective_weight(), self.linh2.bias, act=True,
                        drop_p=p)
        return ops.linear(h2, self.linh3.effective_weight(), self.linh3.bias, act=True, drop_p=p)

    def forward(self, input):
        """(P, 289) -> (tanh sdf (P,1), class logits (P,6) | None): the reference's call signature
        (it returns a dummy tensor as second element when the classifier is off, common/nets/sdf_net.py:119-122)."""
        h3 = self.hidden(input)
        _, raw = ops.sdf_head(h3, self.linh4.weight, self.linh4.bias, 1e30)
        return raw.unsqueeze(1), self.classify(h3)

    def classify(self, h3):
        """common/nets/sdf_net.py:93-94: the logits are read off the INPUT of the last layer"""
        return ops.linear(h3, self.classifier_head.weight, self.classifier_head.bias) if self.use_classifier else None

    def forward_clamped(self, x0, clamp, want_class=False):
        h3 = self.hidden(x0)
        out = ops.sdf_head(h3, self.linh4.weight, self.linh4.bias, clamp)
        return (*out, self.classify(h3)) if want_class else out


# -------------------------------------------------------------------------------------------------
class _OutProj(nn.Linear):
    """distinct type, like torch's NonDynamicallyQuantizableLinear: ``init_weights`` (type(m) is
    nn.Linear) skips it, as in the reference."""


class _MHA(nn.Module):
    """nn.MultiheadAttention's parameter layout (packed in-projection)."""

    def __init__(self, d_model, nhead):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = _OutProj(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, normalize_before=False):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        self.self_attn = _MHA(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout

    def forward(self, x, n_query=None):
        """post-norm layer on batch-first tokens (B,S,D); pos embedding is identically zero on
        this path (main/model.py:542-562).  ``n_query``: only the first n_query rows are produced
        (all S rows still act as keys/values) - used for the last layer of a stack, whose other rows
        nothing downstream reads (heads use rows < num_samp_*, the decoder masks keys >= num_samp_hand)."""
        p = self.p if self.training else 0.0
        a = self.self_attn
        E = x.shape[-1]
        if self.normalize_before:
            return self.forward_pre(x, p)
        if n_query is None or n_query >= x.shape[1]:
            qkv = ops.linear(x, a.in_proj_weight, a.in_proj_bias)
            o = ops.attention_self(qkv, a.num_heads, drop_p=p)
        else:
            xq = x[:, :n_query].contiguous()
            q = ops.linear(xq, a.in_proj_weight[:E], a.in_proj_bias[:E])
            kv = ops.linear(x, a.in_proj_weight[E:], a.in_proj_bias[E:])
            o = ops.attention_cross(q, kv, a.num_heads, drop_p=p)
            x = xq
        o = ops.linear(o, a.out_proj.weight, a.out_proj.bias)
        x = ops.add_layernorm(x, o, self.norm1.weight, self.norm1.bias, self.norm1.eps, p)
        h = ops.linear(x, self.linear1.weight, self.linear1.bias, act=True, drop_p=p)
        h = ops.linear(h, self.linear2.weight, self.linear2.bias)
        return ops.add_layernorm(x, h, self.norm2.weight, self.norm2.bias, self.norm2.eps, p)

    def forward_pre(self, x, p):
        """cfg.pre_norm (common/nets/transformer.py:304-321): x += drop(attn(LN1 x)); x += drop(FFN(LN2 x)) - every row is produced
        (the residual stream of all rows feeds the next layer), op by op on the same kernels"""
        a = self.self_attn
        x2 = ops.add_layernorm(x, None, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        o = ops.attention_self(ops.linear(x2, a.in_proj_weight, a.in_proj_bias), a.num_heads, drop_p=p)
        x = ops.residual_dropout(x, ops.linear(o, a.out_proj.weight, a.out_proj.bias), p)
        x2 = ops.add_layernorm(x, None, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = ops.linear(x2, self.linear1.weight, self.linear1.bias, act=True, drop_p=p)
        return ops.residual_dropout(x, ops.linear(h, self.linear2.weight, self.linear2.bias), p)


class TransformerEncoder(nn.Module):
    def __init__(self, d_model, nhead, num_layers, dim_feedforward, dropout, normalize_before=False):
        super().__init__()
        self.layers = nn.ModuleList(TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, normalize_before)
                                    for _ in range(num_layers))
        # common/nets/transformer.py:86: the stack's closing norm exists only with normalize_before; registered BEFORE inter_norm as in
        # the reference's TransformerEncoder.__init__ (:171-172) - parameters() order is what checkpointed optimizer state follows
        self.norm = nn.LayerNorm(d_model) if normalize_before else None
        self.inter_norm = nn.LayerNorm(d_model)
        self.normalize_before = bool(normalize_before)
        self.num_layers = num_layers

    def forward(self, x, n_keep=None):
        """``n_keep``: the caller only reads rows < n_keep of the outputs -> the last layer computes only
        those rows and every returned tensor is cut to n_keep rows (results for those rows are unchanged)."""
        inter = []
        n = self.inter_norm
        last = len(self.layers) - 1
        if self.normalize_before:
            # pre-norm stack (cfg.pre_norm): op by op, all rows through every layer, encoder.norm on the output (:199-200)
            for layer in self.layers:
                x = layer(x)
                inter.append(ops.add_layernorm(x, None, n.weight, n.bias, n.eps))
            x = ops.add_layernorm(x, None, self.norm.weight, self.norm.bias, self.norm.eps)
            if n_keep is not None and n_keep < x.shape[1]:
                x, inter = x[:, :n_keep].contiguous(), [y[:, :n_keep] for y in inter]
            return x, torch.stack(inter)
        if FUSED_LAYER_NODES:
            # one autograd node per layer (+ its inter_norm): multi-consumer gradients are summed inside the kernels
            for i, l in enumerate(self.layers):
                a = l.self_attn
                x, y = ops.encoder_layer(x, n_keep if i == last else None, l.p if l.training else 0.0, a.num_heads,
                                         a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                         l.norm1.weight, l.norm1.bias, l.linear1.weight, l.linear1.bias, l.linear2.weight,
                                         l.linear2.bias, l.norm2.weight, l.norm2.bias, n.weight, n.bias, l.norm1.eps,
                                         n_inter=n_keep if INTER_ROWS else None)   # inter_norm only on the rows the caller reads
                inter.append(y if n_keep is None or y.shape[1] == n_keep else y[:, :n_keep])
            return x, torch.stack(inter)
        for i, layer in enumerate(self.layers):
            x = layer(x, n_keep if i == last else None)
            y = ops.add_layernorm(x, None, n.weight, n.bias, n.eps)
            inter.append(y if n_keep is None or y.shape[1] == n_keep else y[:, :n_keep])
        return x, torch.stack(inter)            # memory (B,S|n_keep,D), intermediates (L,B,S|n_keep,D)


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, normalize_before=False):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        self.self_attn = _MHA(d_model, nhead)
        self.multihead_attn = _MHA(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.p = dropout

    def forward(self, tgt, memory, query_pos, tgt_mask_u8, kv_len):
        p = self.p if self.training else 0.0
        E = tgt.shape[-1]
        sa, ca = self.self_attn, self.multihead_attn
        if self.normalize_before:
            # cfg.pre_norm (common/nets/transformer.py:397-424)
            t2 = ops.add_layernorm(tgt, None, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            qk = ops.linear(t2 + query_pos, sa.in_proj_weight[:2 * E], sa.in_proj_bias[:2 * E])
            v = ops.linear(t2, sa.in_proj_weight[2 * E:], sa.in_proj_bias[2 * E:])
            o = ops.attention_small(qk[..., :E], qk[..., E:], v, tgt_mask_u8, sa.num_heads, p)
            tgt = ops.residual_dropout(tgt, ops.linear(o, sa.out_proj.weight, sa.out_proj.bias), p)
            t2 = ops.add_layernorm(tgt, None, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            q = ops.linear(t2 + query_pos, ca.in_proj_weight[:E], ca.in_proj_bias[:E])
            kv = ops.linear(memory, ca.in_proj_weight[E:], ca.in_proj_bias[E:])
            o = ops.attention_cross(q, kv, ca.num_heads, kv_len, p)
            tgt = ops.residual_dropout(tgt, ops.linear(o, ca.out_proj.weight, ca.out_proj.bias), p)
            t2 = ops.add_layernorm(tgt, None, self.norm3.weight, self.norm3.bias, self.norm3.eps)
            h = ops.linear(t2, self.linear1.weight, self.linear1.bias, act=True, drop_p=p)
            return ops.residual_dropout(tgt, ops.linear(h, self.linear2.weight, self.linear2.bias), p)
        # masked self-attention over the queries: q = k = tgt + query_pos, v = tgt
        qk = ops.linear(tgt + query_pos, sa.in_proj_weight[:2 * E], sa.in_proj_bias[:2 * E])
        v = ops.linear(tgt, sa.in_proj_weight[2 * E:], sa.in_proj_bias[2 * E:])
        o = ops.attention_small(qk[..., :E], qk[..., E:], v, tgt_mask_u8, sa.num_heads, p)
        o = ops.linear(o, sa.out_proj.weight, sa.out_proj.bias)
        tgt = ops.add_layernorm(tgt, o, self.norm1.weight, self.norm1.bias, self.norm1.eps, p)
        # cross-attention to the encoder memory; only keys < kv_len (the hand points) are visible
        q = ops.linear(tgt + query_pos, ca.in_proj_weight[:E], ca.in_proj_bias[:E])
        kv = ops.linear(memory, ca.in_proj_weight[E:], ca.in_proj_bias[E:])
        o = ops.attention_cross(q, kv, ca.num_heads, kv_len, p)
        o = ops.linear(o, ca.out_proj.weight, ca.out_proj.bias)
        tgt = ops.add_layernorm(tgt, o, self.norm2.weight, self.norm2.bias, self.norm2.eps, p)
        h = ops.linear(tgt, self.linear1.weight, self.linear1.bias, act=True, drop_p=p)
        h = ops.linear(h, self.linear2.weight, self.linear2.bias)
        return ops.add_layernorm(tgt, h, self.norm3.weight, self.norm3.bias, self.norm3.eps, p)


class TransformerDecoder(nn.Module):
    def __init__(self, d_model, nhead, num_layers, dim_feedforward, dropout, normalize_before=False):
        super().__init__()
        self.layers = nn.ModuleList(TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, normalize_before)
                                    for _ in range(num_layers))
        self.norm = nn.LayerNorm(d_model)
        self.normalize_before = bool(normalize_before)

    def forward(self, memory, query_embed, tgt_mask_u8, kv_len):
        B = memory.shape[0]
        n = self.norm
        l0 = self.layers[0]
        if FUSED_LAYER_NODES and not self.normalize_before and ops.decoder_layer_ok(l0.p if l0.training else 0.0, memory, query_embed, l0.linear1.weight):
            # one C-ABI call per layer and direction (csrc/layers.hip hoisdf_decoder_layer_fwd / _bwd)
            x = torch.zeros(B, query_embed.shape[0], query_embed.shape[1], device=memory.device)
            outs = []
            for l in self.layers:
                sa, ca = l.self_attn, l.multihead_attn
                x, y = ops.decoder_layer(x, memory, query_embed, tgt_mask_u8, kv_len, l.p if l.training else 0.0, sa.num_heads, l.norm1.eps,
                                         sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
                                         ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias,
                                         l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias,
                                         l.norm1.weight, l.norm1.bias, l.norm2.weight, l.norm2.bias, l.norm3.weight, l.norm3.bias,
                                         n.weight, n.bias)
                outs.append(y)
            return torch.stack(outs)
        qpos = query_embed.unsqueeze(0).expand(B, -1, -1).contiguous()
        x = torch.zeros_like(qpos)
        outs = []
        for layer in self.layers:
            x = layer(x, memory, qpos, tgt_mask_u8, kv_len)
            outs.append(ops.add_layernorm(x, None, n.weight, n.bias, n.eps))
        return torch.stack(outs)                 # (L,B,Q,D)


_MASK_CACHE = {}


def _require_zero_pos(pos_embed):
    """The reference adds ``pos`` to the QUERY and KEY inputs of every encoder layer and to the decoder's memory keys, never to
    the values (common/nets/transformer.py:283-302,366-395); it only ever passes zeros (main/model.py:542,560), for which
    that equals no embedding at all - the only case the fused layers implement.  Anything else is refused rather than
    silently computed with other semantics (one device read, on the reference-signature entry only)."""
    if pos_embed is not None and bool((pos_embed != 0).any()):
        raise NotImplementedError("a non-zero pos_embed is not supported: the reference adds it to q and k of every layer "
                                  "(common/nets/transformer.py:283-302) and always passes zeros (main/model.py:542,560)")


def _mask_to_u8(mask: Optional[torch.Tensor], nq: int, device) -> torch.Tensor:
    """uint8 device copy of a (constant) boolean attention mask, uploaded once per (mask, device):
    a per-step host->device copy of pageable memory would synchronise the host with the stream."""
    key = (None if mask is None else mask.cpu().numpy().tobytes(), nq, str(device))
    m = _MASK_CACHE.get(key)
    if m is None:
        if mask is None:
            m = torch.zeros(nq, nq, dtype=torch.uint8, device=device)
        else:
            m = mask.to(dtype=torch.uint8).contiguous().to(device)
        _MASK_CACHE[key] = m
    return m


def _kv_len_from_memory_mask(memory_mask: Optional[torch.Tensor], S: int) -> int:
    """The reference's memory masks (common/utils/misc.py:34-47) hide a suffix of the keys for
    every query; the kernels take that as a key-count.  Anything else is rejected loudly."""
    if memory_mask is None:
        return S
    m = memory_mask.to(torch.bool).cpu()
    kv = int((~m[0]).sum())
    expect = torch.zeros_like(m)
    expect[:, kv:] = True
    if not torch.equal(m, expect) or kv == 0:
        raise ValueError("memory_mask must mask exactly a suffix of the keys, identically for all queries")
    return kv


class Transformer(nn.Module):
    """Encoder + decoder (hand stream).  ``forward`` keeps the reference signature (seq-first)."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False,
                 return_intermediate_dec=False):
        super().__init__()
        assert activation == "relu", "the reference builds relu layers (main/model.py:704-720)"
        self.encoder = TransformerEncoder(d_model, nhead, num_encoder_layers, dim_feedforward, dropout, normalize_before)
        self.decoder = TransformerDecoder(d_model, nhead, num_decoder_layers, dim_feedforward, dropout, normalize_before)
        self.d_model, self.nhead = d_model, nhead
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward_batch_first(self, tokens, query_embed, tgt_mask, kv_len, n_keep=None):
        memory, inter = self.encoder(tokens, n_keep)
        hs = self.decoder(memory, query_embed, _mask_to_u8(tgt_mask, query_embed.shape[0], tokens.device), kv_len)
        return hs, memory, inter

    def forward(self, src, mask, query_embed, pos_embed, tgt_mask=None, src_mask=None, memory_mask=None):
        assert mask is None and src_mask is None, "padding / source masks are unused on this path"
        _require_zero_pos(pos_embed)
        x = src
        kv_len = _kv_len_from_memory_mask(memory_mask, src.shape[0])
        hs, memory, inter = self.forward_batch_first(x.permute(1, 0, 2).contiguous(), query_embed, tgt_mask,
                                                     kv_len)
        return hs.permute(0, 2, 1, 3), memory.permute(1, 0, 2), inter.permute(0, 2, 1, 3), None


class VoteTransformer(nn.Module):
    """Encoder only (object stream)."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, dim_feedforward=2048, dropout=0.1,
                 activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        assert activation == "relu"
        self.encoder = TransformerEncoder(d_model, nhead, num_encoder_layers, dim_feedforward, dropout, normalize_before)
        self.d_model, self.nhead = d_model, nhead
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward_batch_first(self, tokens, n_keep=None):
        return self.encoder(tokens, n_keep)

    def forward(self, src, mask, pos_embed, src_mask=None):
        assert mask is None and src_mask is None
        _require_zero_pos(pos_embed)
        x = src
        memory, inter = self.encoder(x.permute(1, 0, 2).contiguous())
        return memory.permute(1, 0, 2), inter.permute(0, 2, 1, 3)
