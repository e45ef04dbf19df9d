"""-m gpu: (f4) the fused BatchNorm2d (+ residual) (+ ReLU) passes of a channels_last encoder map (csrc/bnact.hip, ops.bn_act)
against torch's own batch_norm + add + relu evaluated in fp64 (forward, running statistics, every gradient), at ragged row counts,
channel counts whose thread groups do not divide the block, strided (channel-sliced) inputs and gradients, and through the encoder."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def reference(x, w, b, res, rm, rv, training, momentum, eps, relu):
    """torch's composition in fp64 -> (y, running mean, running var) with autograd attached"""
    rm64, rv64 = (None if rm is None else rm.double().clone()), (None if rv is None else rv.double().clone())
    y = F.batch_norm(x, rm64, rv64, w, b, training, momentum, eps)
    if res is not None:
        y = y + res
    return (F.relu(y) if relu else y), rm64, rv64


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


CASES = [
    # N, C, H, W, relu, residual
    (4, 64, 16, 16, True, False),
    (4, 64, 16, 16, True, True),
    (2, 2048, 8, 8, True, True),
    (3, 32, 20, 12, True, False),          # 720 rows: ragged against every block slice
    (2, 96, 7, 5, True, True),             # 12 threads per row: 21 rows per block iteration, 4 idle threads
    (5, 256, 9, 11, False, False),         # trailing BatchNorm of a downsample branch: no ReLU
    (2, 512, 6, 6, False, True),
    (1, 8, 3, 3, True, True),              # one thread per row
]


@pytest.mark.parametrize("n,c,h,w,relu,has_res", CASES)
def test_training_forward_backward_match_torch_fp64(n, c, h, w, relu, has_res):
    from hoisdf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1234 + c + h)
    dev = "cuda"
    x = (torch.randn(n, c, h, w, device=dev, generator=g) * 1.7 + 0.6).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    res = torch.randn(n, c, h, w, device=dev, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True) if has_res else None
    wt = (torch.rand(c, device=dev, generator=g) + 0.5).requires_grad_(True)
    bs = (torch.randn(c, device=dev, generator=g) * 0.3).requires_grad_(True)
    rm, rv = torch.randn(c, device=dev, generator=g) * 0.1, torch.rand(c, device=dev, generator=g) + 0.5
    gy = torch.randn(n, c, h, w, device=dev, generator=g).contiguous(memory_format=torch.channels_last)

    rm1, rv1 = rm.clone(), rv.clone()
    y = ops.bn_act(x, wt, bs, rm1, rv1, True, 0.1, 1e-5, relu, res)
    assert y.is_contiguous(memory_format=torch.channels_last) or c == 1
    y.backward(gy)
    got = [y.detach(), x.grad, wt.grad, bs.grad] + ([res.grad] if has_res else [])

    x64, w64, b64 = x.detach().double().requires_grad_(True), wt.detach().double().requires_grad_(True), bs.detach().double().requires_grad_(True)
    r64 = res.detach().double().requires_grad_(True) if has_res else None
    y64, rm64, rv64 = reference(x64, w64, b64, r64, rm, rv, True, 0.1, 1e-5, relu)
    y64.backward(gy.double())
    want = [y64.detach(), x64.grad, w64.grad, b64.grad] + ([r64.grad] if has_res else [])

    names = ["y", "dx", "dgamma", "dbeta", "dres"]
    # ReLU gates at rounding level may flip: compare where the fp64 pre-activation is clear of zero
    for name, a, b_ in zip(names, got, want):
        assert rel(a, b_) < 2e-5, (name, rel(a, b_))
    assert rel(rm1, rm64) < 1e-6 and rel(rv1, rv64) < 1e-5


def test_statistics_survive_a_large_mean():
    """|mean| = 1000 sigma: a plain sum of squares in f32 would lose the variance entirely"""
    from hoisdf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(7)
    x = (torch.randn(8, 64, 32, 32, device="cuda", generator=g) * 0.5 + 500.0).contiguous(memory_format=torch.channels_last)
    rm, rv = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    y = ops.bn_act(x, None, None, rm, rv, True, 1.0, 1e-5, False, None)
    x64 = x.double()
    var = x64.var(dim=(0, 2, 3), unbiased=False)
    want = (x64 - x64.mean(dim=(0, 2, 3), keepdim=True)) / (var.view(1, -1, 1, 1) + 1e-5).sqrt()
    assert float((y.double() - want).abs().max()) < 2e-3           # (the INPUT's own f32 spacing at 500 is 3e-5 = 6e-5 sigma)
    assert rel(rv, x64.var(dim=(0, 2, 3), unbiased=True)) < 1e-4 and rel(rm, x64.mean(dim=(0, 2, 3))) < 1e-7


def test_two_runs_are_bit_identical():
    from hoisdf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(32, 64, 64, 64, device="cuda", generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = torch.rand(64, device="cuda", generator=g).requires_grad_(True)
    bs = torch.randn(64, device="cuda", generator=g).requires_grad_(True)
    gy = torch.randn_like(x)
    outs = []
    for _ in range(2):
        x.grad = wt.grad = bs.grad = None
        y = ops.bn_act(x, wt, bs, None, None, True, 0.1, 1e-5, True, None)
        y.backward(gy)
        outs.append((y.detach().clone(), x.grad.clone(), wt.grad.clone(), bs.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_channel_slices_keep_their_row_stride():
    """the input and the upstream gradient are channel slices of wider channels_last maps (what torch.cat hands around)"""
    from hoisdf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    wide = torch.randn(2, 160, 12, 10, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    x = wide[:, 32:96].detach().requires_grad_(True)             # row stride 160, 64 channels
    gwide = torch.randn(2, 192, 12, 10, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    gy = gwide[:, 64:128]
    wt, bs = torch.rand(64, device="cuda", generator=g) + 0.5, torch.randn(64, device="cuda", generator=g)
    y = ops.bn_act(x, wt, bs, None, None, True, 0.1, 1e-5, True, None)
    y.backward(gy)
    x64 = x.detach().double().requires_grad_(True)
    y64, _, _ = reference(x64, wt.double(), bs.double(), None, None, None, True, 0.1, 1e-5, True)
    y64.backward(gy.double())
    assert rel(y, y64) < 2e-5 and rel(x.grad, x64.grad) < 2e-5


def test_evaluation_mode_uses_the_running_statistics():
    from hoisdf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(4, 128, 10, 10, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    res = torch.randn(4, 128, 10, 10, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    wt, bs = torch.rand(128, device="cuda", generator=g) + 0.5, torch.randn(128, device="cuda", generator=g)
    rm, rv = torch.randn(128, device="cuda", generator=g), torch.rand(128, device="cuda", generator=g) + 0.2
    rm0, rv0 = rm.clone(), rv.clone()
    y = ops.bn_act(x, wt, bs, rm, rv, False, 0.1, 1e-5, True, res)
    want = F.relu(F.batch_norm(x.double(), rm.double(), rv.double(), wt.double(), bs.double(), False, 0.1, 1e-5) + res.double())
    assert rel(y, want) < 2e-6
    assert torch.equal(rm, rm0) and torch.equal(rv, rv0)


@pytest.mark.parametrize("resnet", [18, 50])
def test_encoder_with_the_fused_passes_is_as_accurate_as_the_library_sequence(resnet):
    """backbone + pyramid decoder, training mode, forward + backward on the same weights three ways: the fused passes (f32), the
    library sequence (f32: MIOpen BatchNorm + ATen add / relu) and the library sequence in fp64 = the truth.  Forty layers deep,
    ReLU gates at rounding level flip in either f32 run, so the two f32 runs are each held to the truth, the fused one no worse."""
    from hoisdf_amd.nets import encoder as E
    torch.manual_seed(0)
    bb, dec = E.BackboneNet(resnet).cuda(), E.DecoderNet(resnet).cuda()
    for m in list(bb.modules()) + list(dec.modules()):
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)           # (the reference's std = 0.001 init makes every activation vanish)
    bb.to(memory_format=torch.channels_last); dec.to(memory_format=torch.channels_last)
    img = torch.randn(4, 3, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    state = {k: v.clone() for k, v in list(bb.state_dict().items()) + [("d." + k, v) for k, v in dec.state_dict().items()]}

    def run(fused, dtype):
        E.set_bn_fused(fused)
        bb.to(dtype); dec.to(dtype)
        bb.load_state_dict({k: v for k, v in state.items() if not k.startswith("d.")})
        dec.load_state_dict({k[2:]: v for k, v in state.items() if k.startswith("d.")})
        bb.train(); dec.train()
        for p in list(bb.parameters()) + list(dec.parameters()):
            p.grad = None
        feat, skips = bb(img.to(dtype))
        pyr, aux = dec(feat, skips)
        loss = sum((v * v).mean() for v in pyr.values()) + aux.mean()
        loss.backward()
        grads = {n: p.grad.double() for n, p in list(bb.named_parameters()) + [("d." + n, p) for n, p in dec.named_parameters()] if p.grad is not None}
        bufs = {n: b.clone() for n, b in bb.named_buffers()}
        return loss.detach().double(), {k: v.detach().double() for k, v in pyr.items()}, grads, bufs

    try:
        l1, p1, g1, b1 = run(True, torch.float32)
        l0, p0, g0, b0 = run(False, torch.float32)
        lt, pt, gt, bt = run(False, torch.float64)
    finally:
        E.set_bn_fused(True)
        bb.float(); dec.float()
    assert abs(float(l1 - lt)) <= max(3 * abs(float(l0 - lt)), 1e-4 * abs(float(lt)))
    for k in pt:
        assert rel(p1[k], pt[k]) <= max(3 * rel(p0[k], pt[k]), 1e-4), (k, rel(p1[k], pt[k]), rel(p0[k], pt[k]))
    assert set(g1) == set(gt)
    # (a convolution bias in front of a BatchNorm has a ZERO gradient in exact arithmetic: what the runs hold there is rounding)
    gmax = max(float(v.norm()) for v in gt.values())

    def err(g):
        return {k: float((g[k] - gt[k]).norm() / gt[k].norm().clamp_min(1e-4 * gmax)) for k in gt}
    e1, e0 = err(g1), err(g0)
    # (forty layers of ReLU gates, convolution solvers picked by MIOpen's search when an earlier test of the process switched it on,
    # atomically accumulated weight gradients: the distance of EITHER f32 run to the truth scatters from run to run - the whole gradient
    # is compared, the median over the parameters, and a single parameter only against a coarse band)
    tot = lambda g: (sum(float((g[k] - gt[k]).norm()) ** 2 for k in gt) / sum(float(gt[k].norm()) ** 2 for k in gt)) ** 0.5
    assert tot(g1) <= max(2.0 * tot(g0), 1e-4), (tot(g1), tot(g0))
    assert sorted(e1.values())[len(e1) // 2] <= 2 * sorted(e0.values())[len(e0) // 2] + 1e-6         # median over the parameters
    worst = max((e1[k] / max(e0[k], 2e-3), k) for k in gt)
    assert worst[0] < 6.0, (worst, e1[worst[1]], e0[worst[1]])
    for k in bt:
        if bt[k].dtype.is_floating_point:
            assert rel(b1[k], bt[k]) <= max(3 * rel(b0[k], bt[k]), 1e-4), (k, rel(b1[k], bt[k]), rel(b0[k], bt[k]))
        else:
            assert torch.equal(b1[k], bt[k]), k
