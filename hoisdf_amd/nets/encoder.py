"""Image encoder: ResNet backbone + U-Net style CNN decoder that emits the 5-level feature
pyramid.  North-star: "the image encoder stays PyTorch-ROCm" - this file is plain torch.nn
(MIOpen convolutions); it is *not* part of the HIP hot path.

Drop-in contract: parameter names and shapes equal the
reference's ``backbone_net.resnet.*`` / ``decoder_net.resnet_decoder.*`` so released
checkpoints load with ``strict=True``.
Reference: common/nets/resnet.py:13-98, common/nets/module.py:18-218, common/nets/layer.py:23-63.
torchvision is not available in this environment, so the two residual block types are
defined here (standard He et al. v1.5 blocks: stride on the 3x3 conv).
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

_PENDING_NBT = []          # num_batches_tracked buffers touched in this forward: bumped with ONE multi-tensor add


# HOISDF_BN=torch: BatchNorm / add / ReLU stay the library sequence (MIOpen BatchNorm + ATen add + clamp, seven passes over the map
# forward, eight backward); default: the fused two-pass-per-direction HIP kernels (ops.bn_act, csrc/bnact.hip)
_BN_FUSED = [os.environ.get("HOISDF_BN", "hip") != "torch"]


def set_bn_fused(on: bool) -> None:
    _BN_FUSED[0] = bool(on)


def bn_act(bn: nn.BatchNorm2d, x: torch.Tensor, relu: bool, residual=None) -> torch.Tensor:
    """relu?(bn(x) (+ residual)): the fused HIP passes on a CUDA map (training: batch statistics; evaluation without a gradient
    path: running statistics), otherwise the plain torch (MIOpen) BatchNorm + add + relu"""
    if _BN_FUSED[0] and x.is_cuda:
        from .. import ops
        train = bn.training or bn.running_mean is None
        grad = torch.is_grad_enabled() and (x.requires_grad or (bn.weight is not None and bn.weight.requires_grad) or
                                            (residual is not None and residual.requires_grad))
        # (a map that is not channels_last-dense - an NCHW encoder - would be transposed on the way in and handed on in another layout: torch's path)
        if (ops.bn_act_supported(x) and x.is_contiguous(memory_format=torch.channels_last) and (train or not grad)
                and (not train or bn.momentum is not None or not bn.track_running_stats)):
            track = train and bn.track_running_stats and bn.running_mean is not None
            if track:
                _PENDING_NBT.append(bn.num_batches_tracked)
            return ops.bn_act(x, bn.weight, bn.bias, bn.running_mean if (track or not train) else None,
                              bn.running_var if (track or not train) else None, train, bn.momentum, bn.eps, relu, residual)
    if bn.training and bn.track_running_stats and bn.momentum is not None and x.is_cuda:
        # what nn.BatchNorm2d.forward does, minus its per-module `num_batches_tracked += 1` kernel (69 tiny launches per
        # step for ResNet-50 + decoder): the counters are bumped together by flush_bn_counters()
        _PENDING_NBT.append(bn.num_batches_tracked)
        y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)
    else:
        y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def flush_bn_counters() -> None:
    """num_batches_tracked += 1 for every BatchNorm the fused path ran since the last flush (what nn.BatchNorm2d.forward
    does one tiny kernel at a time)"""
    if _PENDING_NBT:
        with torch.no_grad():
            torch._foreach_add_(list(_PENDING_NBT), 1)
        _PENDING_NBT.clear()


class _FusedSeq(nn.Sequential):
    """nn.Sequential (same child indices = same state-dict keys) whose forward runs every BatchNorm2d -> ReLU pair (or a
    trailing BatchNorm2d) through bn_act"""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.BatchNorm2d):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = bn_act(m, x, relu)
                i += 2 if relu else 1
            else:
                x = m(x)
                i += 1
        return x


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = bn_act(self.bn1, self.conv1(x), True)
        return bn_act(self.bn2, self.conv2(y), True, residual=idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = bn_act(self.bn1, self.conv1(x), True)
        y = bn_act(self.bn2, self.conv2(y), True)
        return bn_act(self.bn3, self.conv3(y), True, residual=idt)


_SPEC = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]),
         50: (Bottleneck, [3, 4, 6, 3]), 101: (Bottleneck, [3, 4, 23, 3]),
         152: (Bottleneck, [3, 8, 36, 3])}


class ResNetBackbone(nn.Module):
    """Returns (stride-32 map, dict of the 5 skip taps).  common/nets/resnet.py:70-87."""

    def __init__(self, resnet_type: int):
        super().__init__()
        block, counts = _SPEC[resnet_type]
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        stages = []
        for i, (planes, n) in enumerate(zip([64, 128, 256, 512], counts)):
            stride = 1 if i == 0 else 2
            blocks = []
            for j in range(n):
                ds = None
                s = stride if j == 0 else 1
                if j == 0 and (s != 1 or inpl != planes * block.expansion):
                    ds = _FusedSeq(nn.Conv2d(inpl, planes * block.expansion, 1, s, bias=False),
                                   nn.BatchNorm2d(planes * block.expansion))
                blocks.append(block(inpl, planes, s, ds))
                inpl = planes * block.expansion
            stages.append(nn.Sequential(*blocks))
        self.layer1, self.layer2, self.layer3, self.layer4 = stages
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, mean=0, std=0.001)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        skips = {}
        x = bn_act(self.bn1, self.conv1(x), True)
        skips["stride2"] = x
        x = self.layer1(self.maxpool(x))
        skips["stride4"] = x
        x = self.layer2(x)
        skips["stride8"] = x
        x = self.layer3(x)
        skips["stride16"] = x
        x = self.layer4(x)
        skips["stride32"] = x
        flush_bn_counters()
        return x, skips


class BackboneNet(nn.Module):
    def __init__(self, resnet_type: int):
        super().__init__()
        self.resnet = ResNetBackbone(resnet_type)

    def forward(self, img):
        return self.resnet(img)


def _convs(dims: List[int], kernel=3, padding=1, bnrelu_final=True) -> nn.Sequential:
    """Conv(+BN+ReLU) chain; Sequential indices match common/nets/layer.py:23-42."""
    mods = []
    for i in range(len(dims) - 1):
        mods.append(nn.Conv2d(dims[i], dims[i + 1], kernel, 1, padding))
        if i < len(dims) - 2 or bnrelu_final:
            mods += [nn.BatchNorm2d(dims[i + 1]), nn.ReLU(inplace=True)]
    return _FusedSeq(*mods)


def _deconvs(dims: List[int]) -> nn.Sequential:
    """ConvTranspose2d(4,2,1)+BN+ReLU chain; indices match common/nets/layer.py:45-63."""
    mods = []
    for i in range(len(dims) - 1):
        mods += [nn.ConvTranspose2d(dims[i], dims[i + 1], 4, 2, 1, 0, bias=False),
                 nn.BatchNorm2d(dims[i + 1]), nn.ReLU(inplace=True)]
    return _FusedSeq(*mods)


class _PyramidDecoder(nn.Module):
    """Shared top-down pass: at each level up-sample, concatenate the (optionally 1x1-reduced)
    skip, fuse with a 3x3 conv; every fused map is one pyramid level."""

    levels = ("stride16", "stride8", "stride4", "stride2")

    def _heads(self, c, hidden):
        self.convOut_hm = _convs([c] + hidden + [1], 1, 0, False)
        self.convOut_hand_seg = _convs([c] + hidden + [1], 1, 0, False)
        self.convOut_obj_seg = _convs([c] + hidden + [1], 1, 0, False)

    def _top_down(self, top, skips, pyramid):
        x = top
        for i, name in enumerate(self.levels, start=1):
            red = getattr(self, f"conv{i}d", None)
            skip = skips[name] if red is None else red(skips[name])
            x = getattr(self, f"conv{i}")(torch.cat((skip, getattr(self, f"deconv{i}")(x)), 1))
            pyramid[name] = x
        aux = torch.cat([self.convOut_hm(x), self.convOut_hand_seg(x).sigmoid(),
                         self.convOut_obj_seg(x).sigmoid()], dim=1)
        flush_bn_counters()
        return pyramid, aux


class Decoder(_PyramidDecoder):
    """Small decoder (pyramid channels 512/256/128/64/32 -> C=992).  module.py:51-144."""

    def __init__(self, resnet_type: int):
        super().__init__()
        deep = resnet_type >= 50
        self.deep = deep
        if deep:
            self.conv0d = _convs([2048, 512], 1, 0)
        e = [1024, 512, 256, 64] if deep else [256, 128, 64, 64]
        self.conv1d = _convs([e[0], 256], 1, 0)
        self.deconv1 = _deconvs([2048 if deep else 512, 256])
        self.conv1 = _convs([512, 256])
        self.conv2d = _convs([e[1], 128], 1, 0)
        self.deconv2 = _deconvs([256, 128])
        self.conv2 = _convs([256, 128])
        self.conv3d = _convs([e[2], 64], 1, 0)
        self.deconv3 = _deconvs([128, 64])
        self.conv3 = _convs([128, 64])
        self.conv4d = _convs([e[3], 32], 1, 0)
        self.deconv4 = _deconvs([64, 64])
        self.conv4 = _convs([64 + 32, 32])
        self._heads(32, [32])

    def forward(self, img_feat, skips):
        pyr = {"stride32": self.conv0d(img_feat) if self.deep else img_feat}
        return self._top_down(img_feat, skips, pyr)


class Decoder_big(_PyramidDecoder):
    """setting="ho3d" decoder (2048/1024/512/256/128 -> C=3968), ResNet-50+ only.
    module.py:147-218."""

    def __init__(self):
        super().__init__()
        self.deconv1 = _deconvs([2048, 1024])
        self.conv1 = _convs([2048, 1024])
        self.deconv2 = _deconvs([1024, 512])
        self.conv2 = _convs([1024, 512])
        self.deconv3 = _deconvs([512, 256])
        self.conv3 = _convs([512, 256])
        self.deconv4 = _deconvs([256, 128])
        self.conv4 = _convs([64 + 128, 128])
        self._heads(128, [128, 64])

    def forward(self, img_feat, skips):
        return self._top_down(img_feat, skips, {"stride32": img_feat})


class DecoderNet(nn.Module):
    def __init__(self, resnet_type: int, big: bool = False):
        super().__init__()
        self.resnet_decoder = Decoder_big() if big else Decoder(resnet_type)

    def forward(self, img_feat, skips):
        return self.resnet_decoder(img_feat, skips)
