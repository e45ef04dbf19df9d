"""-m gpu: HOISDF_DETERMINISTIC (ops.set_deterministic): two identical training steps - encoder, hot path forward,
backward, every parameter gradient and the pyramid gradient - are BIT-identical, and the order-fixed kernels agree with the
default (atomic) ones to float-summation noise."""
import random

import pytest
import torch

from hoisdf_amd import testing as T
from hoisdf_amd.config import Config

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def det_mode():
    from hoisdf_amd import ops
    ops.set_deterministic(True)
    yield
    ops.set_deterministic(False)


def _model(nh, no, with_encoder=True):
    from hoisdf_amd.model import get_model
    c = Config()
    c.resnet_type = 18
    c.apply_setting("dexycb")
    c.num_samp_hand, c.num_samp_obj = nh, no
    torch.manual_seed(0)
    return get_model("train", cfg=c, with_encoder=with_encoder).to(DEV).train(), c


def _step(model, batch):
    from hoisdf_amd import ops
    model.zero_grad(set_to_none=True)
    model._py_random = random.Random(0)
    torch.manual_seed(3)
    ops.manual_seed(77)                                   # dropout ON, same stream ids both times
    out = model(*batch, "train", 0, 0.1)
    total = sum(v.mean() for k, v in out.items() if "_out" not in k)
    total.backward()
    torch.cuda.synchronize()
    return float(total.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def _hot_step(model, levels, batch):
    """the HIP scope: everything after the CNN (hot_path) on a given pyramid; returns loss, parameter grads, pyramid grads"""
    from hoisdf_amd import ops
    model.zero_grad(set_to_none=True)
    lv = [l.detach().clone().requires_grad_(True) for l in levels]
    model._py_random = random.Random(0)
    torch.manual_seed(3)
    ops.manual_seed(77)
    loss, out = model.hot_path(ops.PyramidNHWC(lv), *batch, "train", 0, 0.1)
    total = sum(v.mean() for v in loss.values())
    total.backward()
    torch.cuda.synchronize()
    return (float(total.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
            [l.grad for l in lv])


def test_two_identical_hot_path_steps_are_bit_identical(det_mode):
    """B = 4, 384 + 128 points, dropout on: forward value, all 260+ hot-path parameter gradients and the five pyramid-level
    gradients of two identical steps agree bit for bit."""
    from hoisdf_amd import ops
    model, c = _model(384, 128, with_encoder=False)
    levels = [v.to(DEV).permute(0, 2, 3, 1).contiguous() for v in T.synthetic_pyramid(4, seed=3).values()]
    batch = tuple(T.to_device(x, DEV) for x in T.synthetic_batch(4, 384, 128, seed=5))
    l1, g1, p1 = _hot_step(model, levels, batch)
    l2, g2, p2 = _hot_step(model, levels, batch)
    assert l1 == l2
    assert set(g1) == set(g2) and len(g1) > 250
    bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    assert not bad, bad[:10]
    assert all(torch.equal(a, b) for a, b in zip(p1, p2))
    assert all(float(a.abs().max()) > 0 for a in p1)


def test_full_model_steps_are_bit_identical_when_the_encoder_is(det_mode):
    """with the PyTorch / MIOpen encoder in front: its convolutions are outside the HIP scope and are not run-to-run
    reproducible on every MIOpen build; where they are (torch.backends.cudnn.deterministic), the whole step is."""
    det, bench = torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    try:
        model, c = _model(384, 128)
        batch = tuple(T.to_device(x, DEV) for x in T.synthetic_batch(4, 384, 128, seed=5))
        with torch.no_grad():
            e1 = model.decoder_net(*model.backbone_net(batch[0]["img"]))[0]
            e2 = model.decoder_net(*model.backbone_net(batch[0]["img"]))[0]
        if not all(torch.equal(e1[k], e2[k]) for k in e1):
            pytest.skip("the MIOpen encoder forward is not bit-reproducible on this build (outside the HIP hot path)")
        l1, g1 = _step(model, batch)
        l2, g2 = _step(model, batch)
        assert l1 == l2
        bad = [n for n in g1 if not n.startswith(("backbone_net", "decoder_net")) and not torch.equal(g1[n], g2[n])]
        assert not bad, bad[:10]
    finally:
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = det, bench


def test_default_mode_is_not_bit_identical_but_close():
    """sanity of the test above: without the switch the float atomics do show up, at noise level"""
    from hoisdf_amd import ops
    assert not ops.deterministic()
    model, c = _model(384, 128)
    batch = tuple(T.to_device(x, DEV) for x in T.synthetic_batch(4, 384, 128, seed=5))
    _, g1 = _step(model, batch)
    _, g2 = _step(model, batch)
    hot = [n for n in g1 if not n.startswith(("backbone_net", "decoder_net"))]
    num = sum(float((g1[n] - g2[n]).double().pow(2).sum()) for n in hot)
    den = sum(float(g1[n].double().pow(2).sum()) for n in hot)
    assert (num / den) ** 0.5 < 1e-4


def test_deterministic_kernels_match_the_atomic_ones(det_mode):
    """gather backward (pixel-tile owners), LayerNorm / SDF-head column sums (ordered block reduction), 17-query attention
    (query-ordered dk/dv), grad-weight (workspace) against the default kernels."""
    from hoisdf_amd import ops as O
    g = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g, device=DEV)

    def both(fn):
        O.set_deterministic(True)
        a = fn()
        b = fn()
        O.set_deterministic(False)
        r = fn()
        O.set_deterministic(True)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        for x, y in zip(a, r):
            err = float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30)
            assert err < 2e-5, err

    # gather backward, B = 3, all five levels, points partly outside the image
    B, P = 3, 500
    pyr = T.synthetic_pyramid(B, seed=4, nonneg=False)
    inputs, _, meta = T.synthetic_batch(B, P, 8, seed=5)
    pts = (inputs["hand_sdf_points"] * 2.0).to(DEV)
    root, K = meta["mano_root"].to(DEV), meta["cam_intr"].to(DEV)
    gy = rnd(B * P, 992)

    def gather():
        lv = [v.to(DEV).permute(0, 2, 3, 1).contiguous().requires_grad_(True) for v in pyr.values()]
        feat, _ = O.project_gather(O.PyramidNHWC(lv), pts, root, K, 3.1)
        feat.backward(gy)
        return [l.grad for l in lv]
    both(gather)

    x, r = rnd(3000, 256), rnd(3000, 256)
    gam, bet, dy = rnd(256), rnd(256), rnd(3000, 256)

    def ln():
        xs = [t.clone().requires_grad_(True) for t in (x, r, gam, bet)]
        O.manual_seed(5)
        O.add_layernorm(xs[0], xs[1], xs[2], xs[3], 1e-5, 0.1).backward(dy)
        return [t.grad for t in xs]
    both(ln)

    h, w, b0, ds = rnd(5000, 512), rnd(1, 512) * 0.05, rnd(1), rnd(5000)

    def head():
        xs = [t.clone().requires_grad_(True) for t in (h, w, b0)]
        sdf, _ = O.sdf_head(xs[0], xs[1], xs[2], 0.15)
        sdf.backward(ds)
        return [t.grad for t in xs]
    both(head)

    q, k, v, go = rnd(4, 17, 256), rnd(4, 17, 256), rnd(4, 17, 256), rnd(4, 17, 256)
    from hoisdf_amd.model import get_mano_tgt_mask
    m8 = get_mano_tgt_mask().to(torch.uint8).to(DEV)

    def small():
        xs = [t.clone().requires_grad_(True) for t in (q, k, v)]
        O.manual_seed(6)
        O.attention_small(xs[0], xs[1], xs[2], m8, 4, 0.1).backward(go)
        return [t.grad for t in xs]
    both(small)

    xx, W, bb, dyy = rnd(8192, 256), rnd(256, 256) / 16, rnd(256), rnd(8192, 256)

    def lin():
        xs = [t.clone().requires_grad_(True) for t in (xx, W, bb)]
        O.manual_seed(7)
        O.linear(xs[0], xs[1], xs[2], act=True, drop_p=0.1).backward(dyy)
        return [t.grad for t in xs]
    both(lin)

    qkv, go2 = rnd(2, 700, 768), rnd(2, 700, 256)

    def attn():
        xs = qkv.clone().requires_grad_(True)
        O.manual_seed(8)
        O.attention_self(xs, 4, drop_p=0.1).backward(go2)
        return [xs.grad]
    both(attn)
