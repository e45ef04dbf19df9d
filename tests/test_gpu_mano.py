"""-m gpu: the one-kernel MANO head (csrc/mano.hip through hoisdf_mano_head_fwd / _bwd) against
  * fixture g9 = manopth's own ManoLayer.forward and the reference head's 6D -> matrix -> axis-angle functions,
  * the plain PyTorch chain (nets/heads.py + nets/mano.py, the path foreign MANO layers still take) in fp64: forward values,
    the four fused ManoLoss terms, and the gradients - for which the kernel uses the closed form "exp(log(R)) = R on SO(3)"
    while autograd differentiates the whole quaternion / atan2 chain; both are compared with the fp64 chain."""
import copy

import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from hoisdf_amd.nets import heads as HEADS
from hoisdf_amd.nets import mano as MANO

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Foreign(nn.Module):
    """a MANO layer that is not this package's class: ManoHead keeps the PyTorch chain for it"""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer
        self.th_hands_mean = layer.th_hands_mean

    def forward(self, th_pose_coeffs, th_betas):
        return self.layer(th_pose_coeffs, th_betas)


def _inputs(L, B, seed):
    g = torch.Generator().manual_seed(seed)
    pose6d = torch.randn(L, B, 16, 6, generator=g)
    shape = 0.7 * torch.randn(L, B, 10, generator=g)
    mano_param = torch.cat([0.4 * torch.randn(B, 48, generator=g), torch.randn(B, 10, generator=g)], 1)
    return pose6d, shape, mano_param


def _chain(layer, pose6d, shape, mano_param, dtype):
    """the PyTorch chain + ManoLoss in `dtype` on the CPU; returns pred, gt, the four losses"""
    head = HEADS.ManoHead(_Foreign(copy.deepcopy(layer).to(dtype)))
    pred, gt = head.forward_batch_first(pose6d.to(dtype), shape.to(dtype), None if mano_param is None else mano_param.to(dtype))
    losses = None
    if gt is not None:
        losses = HEADS.ManoLoss(1e4, 1e4, 10, 0.1)(pred, gt)[:4]
    return pred, gt, losses


def _close(a, b, atol, what):
    err = float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.1e}"


def test_ground_truth_mode_matches_manopth_fixture():
    """mode 1 (axis-angle coefficients) against manopth's ManoLayer.forward on the same synthetic asset (g9, millimetres)"""
    from hoisdf_amd import ops as O
    g = load_golden("g9_mano")
    layer = MANO.ManoLayer(MANO.synthetic_assets(0)).to(DEV)
    assets = layer.kernel_assets()
    assert assets is not None
    mp = torch.cat([g["pose"], g["betas"]], 1).to(DEV)
    verts, joints, rot = O.mano_gt(mp, assets)
    _close(verts * 1000, g["verts"], 2e-3, "verts (mm)")
    _close(joints * 1000, g["joints"], 2e-3, "joints (mm)")
    from oracle import hoisdf_oracle as R
    _close(rot.view(-1, 3, 3), R.rodrigues_via_quat(g["pose"].reshape(-1, 3)), 1e-6, "gt mano_pose")


def test_predicted_rotations_match_the_reference_head_fixture():
    """mode 0: the rotation output is the reference's rot6d2mat (g9: R), and the skinned result equals the layer run on the
    reference's own mat2aa output (g9: aa) - i.e. the kernel's quaternion / axis-angle chain lands where the reference's does"""
    from hoisdf_amd import ops as O
    g = load_golden("g9_mano")
    layer = MANO.ManoLayer(MANO.synthetic_assets(0)).to(DEV)
    assets = layer.kernel_assets()
    x6 = g["x6"].view(4, 16, 6).to(DEV)
    betas = g["betas"][:4].to(DEV).contiguous()
    verts, joints, rot, sums = O.mano_head(x6, betas, assets)
    assert sums is None
    _close(rot.view(-1, 3, 3), g["R"], 1e-6, "mano_pose")
    v_ref, j_ref = layer.cpu().double()(g["aa"].view(4, 48).double(), g["betas"][:4].double())
    _close(verts * 1000, v_ref, 2e-3, "verts (mm)")
    _close(joints * 1000, j_ref, 2e-3, "joints (mm)")


@pytest.mark.parametrize("L,B", [(3, 4), (1, 1), (3, 32)])
def test_head_and_fused_losses_match_the_fp64_chain(L, B):
    layer = MANO.ManoLayer(MANO.synthetic_assets(0))
    pose6d, shape, mp = _inputs(L, B, seed=L * 100 + B)
    ref_pred, ref_gt, ref_loss = _chain(layer, pose6d, shape, mp, torch.float64)
    head = HEADS.ManoHead(copy.deepcopy(layer)).to(DEV)
    pred, gt = head.forward_batch_first(pose6d.to(DEV), shape.to(DEV), mp.to(DEV))
    assert "loss_sums" in pred                                  # the kernel path ran
    for k, tol in (("verts3d", 1e-6), ("joints3d", 1e-6), ("mano_pose", 2e-6), ("mano_shape", 0.0)):
        _close(pred[k], ref_pred[k], tol, "pred " + k)
        _close(gt[k], ref_gt[k], max(tol, 1e-6) if k != "mano_shape" else 0.0, "gt " + k)
    losses = HEADS.ManoLoss(1e4, 1e4, 10, 0.1)(pred, gt)[:4]
    for name, a, b in zip(("mesh", "joint", "pose", "shape"), losses, ref_loss):
        assert abs(float(a) - float(b)) <= 2e-5 * abs(float(b)) + 1e-9, (name, float(a), float(b))


@pytest.mark.parametrize("L,B,explicit", [(3, 4, False), (3, 4, True), (3, 32, False)])
def test_backward_matches_the_fp64_chain_at_least_as_well_as_fp32_autograd(L, B, explicit):
    layer = MANO.ManoLayer(MANO.synthetic_assets(0))
    pose6d, shape, mp = _inputs(L, B, seed=7 + B)
    g = torch.Generator().manual_seed(3)
    gv, gj, gr = (torch.randn(L, B, 778, 3, generator=g), torch.randn(L, B, 21, 3, generator=g), torch.randn(L, B, 16, 3, 3, generator=g))

    def objective(pred, gt, dtype, dev):
        tot = sum(HEADS.ManoLoss(1e4, 1e4, 10, 0.1)(pred, gt)[:4])
        if explicit:                                            # gradients arriving at the outputs themselves as well
            tot = tot + (pred["verts3d"] * gv.to(dev, dtype)).sum() + (pred["joints3d"] * gj.to(dev, dtype)).sum() \
                + (pred["mano_pose"] * gr.to(dev, dtype)).sum()
        return tot

    grads = {}
    for name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        p, s = pose6d.clone().to(dtype).requires_grad_(True), shape.clone().to(dtype).requires_grad_(True)
        head = HEADS.ManoHead(_Foreign(copy.deepcopy(layer).to(dtype)))
        pred, gt = head.forward_batch_first(p, s, mp.to(dtype))
        objective(pred, gt, dtype, "cpu").backward()
        grads[name] = (p.grad.double(), s.grad.double())
    p, s = pose6d.detach().to(DEV).requires_grad_(True), shape.detach().to(DEV).requires_grad_(True)
    head = HEADS.ManoHead(copy.deepcopy(layer)).to(DEV)
    pred, gt = head.forward_batch_first(p, s, mp.to(DEV))
    assert "loss_sums" in pred
    objective(pred, gt, torch.float32, DEV).backward()
    for i, what in enumerate(("d pose6d", "d shape")):
        truth = grads["f64"][i]
        scale = float(truth.abs().max())
        e_kernel = float((([p, s][i].grad.cpu().double()) - truth).abs().max()) / scale
        e_autograd = float((grads["f32"][i] - truth).abs().max()) / scale
        # fp32 rounding through the chain: 1e-5 of the gradient's scale, or what fp32 autograd of the long chain itself achieves
        print(f"{what}: kernel {e_kernel:.2e}, fp32 autograd {e_autograd:.2e} of max |g|")
        assert e_kernel <= max(1e-5, 1.5 * e_autograd), f"{what}: kernel {e_kernel:.2e} vs fp32 autograd {e_autograd:.2e} (of max |g|)"


def test_layer_with_a_hand_mean_keeps_the_pytorch_chain():
    assets = MANO.synthetic_assets(0)
    assets["th_hands_mean"] = 0.1 * torch.ones(1, 45)
    layer = MANO.ManoLayer(assets).to(DEV)
    assert layer.kernel_assets() is None
    pose6d, shape, mp = _inputs(1, 2, seed=5)
    pred, gt = HEADS.ManoHead(layer).to(DEV).forward_batch_first(pose6d.to(DEV), shape.to(DEV), mp.to(DEV))
    assert "loss_sums" not in pred
    ref_pred, ref_gt, _ = _chain(layer.cpu(), pose6d, shape, mp, torch.float64)
    _close(pred["verts3d"], ref_pred["verts3d"], 2e-6, "verts")
