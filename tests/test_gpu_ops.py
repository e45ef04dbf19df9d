"""-m gpu: every C-ABI kernel against the CPU oracle / plain fp32-fp64 PyTorch on the same seeded
inputs.  Tolerances are written next to each check (fp32 kernels: ~1e-5 relative to the
tensor's scale; the f32 MFMA is an exact fmaf chain, only the summation order differs)."""
import math

import pytest
import torch
import torch.nn.functional as F

from hoisdf_amd import testing as T

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from hoisdf_amd import ops as O
    return O


def oracle():
    from oracle import hoisdf_oracle as R
    return R


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def assert_close(a, b, rel=2e-5, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max())
    assert err <= rel * scale + 1e-12, f"{what}: max abs err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,act", [(300, 223, 289, True), (1000, 512, 992, True), (17, 60, 256, False),
                                       (4096, 768, 256, False), (129, 1, 512, False), (257, 3, 256, True),
                                       (544, 256, 1024, False), (544, 1024, 256, True), (544, 512, 256, False)])
def test_linear_fwd_bwd(M, N, K, act):
    O = ops()
    x = rnd(M, K, seed=1).requires_grad_(True)
    W = (rnd(N, K, seed=2) / math.sqrt(K)).requires_grad_(True)
    b = rnd(N, seed=3).requires_grad_(True)
    gy = rnd(M, N, seed=4)
    ref = F.linear(x.double(), W.double(), b.double())
    ref = F.relu(ref) if act else ref
    ref.backward(gy.double())
    xg, Wg, bg = (t.detach().to(DEV).requires_grad_(True) for t in (x, W, b))
    y = O.linear(xg, Wg, bg, act=act)
    y.backward(gy.to(DEV))
    assert_close(y, ref, what="y")
    assert_close(xg.grad, x.grad, what="dx")
    assert_close(Wg.grad, W.grad, what="dW")
    assert_close(bg.grad, b.grad, what="db")


def test_layernorm_on_the_first_rows_of_every_group():
    """hoisdf_layernorm_rows_fwd / _bwd (the encoder stack's inter_norm on the rows the caller reads): compact output and
    statistics, full-size dx with the untouched rows passing dx_add through, vs torch in fp64."""
    O = ops()
    G, R, Tk, D = 5, 37, 11, 256
    x = rnd(G * R, D, seed=71).to(DEV)
    gam, bet = (rnd(D, seed=72) * 0.3 + 1.0).to(DEV), rnd(D, seed=73).to(DEV)
    gy = rnd(G * Tk, D, seed=74).to(DEV)
    add = rnd(G * R, D, seed=75).to(DEV)
    y, mean, rstd = torch.empty(G * Tk, D, device=DEV), torch.empty(G * Tk, device=DEV), torch.empty(G * Tk, device=DEV)
    O.call("hoisdf_layernorm_rows_fwd", O._p(x), O._p(gam), O._p(bet), O._p(y), O._p(mean), O._p(rstd), G, R, Tk, D, 1e-5, O._st())
    x64 = x.double().view(G, R, D).requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = F.layer_norm(x64[:, :Tk], (D,), g64, b64, 1e-5)
    assert_close(y, ref.reshape(-1, D), what="y")
    ref.backward(gy.double().view(G, Tk, D))
    for use_add in (True, False):
        dx = torch.full((G * R, D), float("nan"), device=DEV)
        dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        O.call("hoisdf_layernorm_rows_bwd", O._p(gy), O._p(x), O._p(gam), O._p(mean), O._p(rstd), O._p(add) if use_add else None,
               O._p(dx), O._p(dg), O._p(db), G, R, Tk, D, O._st())
        want = x64.grad.reshape(-1, D) + (add.double() if use_add else 0.0)
        assert_close(dx, want, what="dx")
        assert_close(dg, g64.grad, what="dgamma"); assert_close(db, b64.grad, what="dbeta")


def test_linear_strided_input_and_weight_slices():
    """x rows with ld > K (the 292-wide decoder-input buffer) and W given as a row slice."""
    O = ops()
    buf = rnd(500, 292, seed=5)
    Wfull = rnd(768, 289, seed=6) / 17
    x = buf[:, :289]
    ref = F.linear(x.double(), Wfull[256:].double())
    y = O.linear(buf.to(DEV)[:, :289], Wfull.to(DEV)[256:])
    assert_close(y, ref, what="strided")


def test_linear_dropout_statistics_and_backward_mask():
    O = ops()
    O.manual_seed(123)
    M, N, K, p = 2048, 512, 256, 0.2
    x = rnd(M, K, seed=7).to(DEV).requires_grad_(True)
    W = (rnd(N, K, seed=8) / 16).to(DEV)
    b = torch.full((N,), 0.5, device=DEV)
    y = O.linear(x, W, b, act=True, drop_p=p)
    base = F.relu(F.linear(x.detach(), W, b))
    pos = base > 0
    kept = (y > 0) & pos
    frac = kept.sum().item() / pos.sum().item()
    assert abs(frac - (1 - p)) < 0.01, frac                       # keep probability
    assert_close(y[kept], base[kept] / (1 - p), what="kept scaling")
    assert float(y[pos & ~kept].abs().max()) == 0.0
    y.sum().backward()
    gref = ((kept.float() / (1 - p)) @ W)                          # d/dx of sum(y)
    assert_close(x.grad, gref, rel=1e-4, what="dropout dx")
    # a different seed draws a different mask
    y2 = O.linear(x.detach(), W, b, act=True, drop_p=p)
    assert ((y2 > 0) != (y > 0)).any()


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("big", [False, True])
def test_project_gather_vs_grid_sample(big):
    O, R = ops(), oracle()
    B, P = 2, 300
    pyr = T.synthetic_pyramid(B, big=big, seed=4, nonneg=False)
    inputs, _, meta = T.synthetic_batch(B, P, 8, seed=5)
    pts = inputs["hand_sdf_points"] * 2.5          # some project outside the image -> border clamp
    pyr_req = {k: v.clone().requires_grad_(True) for k, v in pyr.items()}
    cam_ref, grid = R.project_points(pts, meta["mano_root"], meta["cam_intr"], 3.1)
    feat_ref = R.sample_pyramid(pyr_req, grid, list(pyr))
    gy = rnd(B, P, feat_ref.shape[-1], seed=6)
    feat_ref.backward(gy)

    levels = [v.to(DEV).permute(0, 2, 3, 1).contiguous().requires_grad_(True) for v in pyr.values()]
    feat, cam = O.project_gather(O.PyramidNHWC(levels), pts.to(DEV), meta["mano_root"].to(DEV),
                                 meta["cam_intr"].to(DEV), 3.1)
    assert_close(feat.view(B, P, -1), feat_ref, rel=2e-5, what="gather fwd")
    assert_close(cam.view(B, P, 3), cam_ref, rel=1e-6, what="cam")
    feat.backward(gy.to(DEV).view(B * P, -1))
    for lv, (name, ref) in zip(levels, pyr_req.items()):
        assert_close(lv.grad.permute(0, 3, 1, 2), ref.grad, rel=5e-5, what=f"dpyr {name}")


def test_posenc_weightnorm_sdfhead():
    O, R = ops(), oracle()
    pts = rnd(777, 3, seed=9) * 1.2
    assert_close(O.posenc(pts.to(DEV)), R.posenc(pts), rel=2e-6, what="posenc")

    v = rnd(223, 512, seed=10).requires_grad_(True)
    g = (rnd(223, 1, seed=11).abs() + 0.5).requires_grad_(True)
    Wref = v * (g / v.norm(dim=1, keepdim=True))
    gw = rnd(223, 512, seed=12)
    Wref.backward(gw)
    vg, gg = v.detach().to(DEV).requires_grad_(True), g.detach().to(DEV).requires_grad_(True)
    W = O.weight_norm(vg, gg)
    W.backward(gw.to(DEV))
    assert_close(W, Wref, what="weightnorm")
    assert_close(vg.grad, v.grad, what="dv")
    assert_close(gg.grad, g.grad, what="dg")

    h = rnd(1001, 512, seed=13).requires_grad_(True)
    w = (rnd(1, 512, seed=14) / 40).requires_grad_(True)
    b = torch.tensor([-0.05], requires_grad=True)
    raw_ref = torch.tanh(F.linear(h, w, b))
    sdf_ref = raw_ref.clamp(-0.15, 0.15)
    gs = rnd(1001, 1, seed=15)
    sdf_ref.backward(gs)
    hg, wg, bg = (t.detach().to(DEV).requires_grad_(True) for t in (h, w, b))
    sdf, raw = O.sdf_head(hg, wg, bg, 0.15)
    sdf.backward(gs.to(DEV).view(-1))
    assert_close(raw, raw_ref.view(-1), what="sdf raw")
    assert_close(sdf, sdf_ref.view(-1), what="sdf")
    assert_close(hg.grad, h.grad, what="dh")
    assert_close(wg.grad, w.grad, rel=1e-4, what="dw4")
    assert_close(bg.grad, b.grad, rel=1e-4, what="db4")


# ---------------------------------------------------------------------------------------------
def _ref_attention(q, k, v, H, kv_len=None, mask=None):
    B, Lq, E = q.shape
    Lk = k.shape[1]
    dh = E // H
    qh = q.view(B, Lq, H, dh).transpose(1, 2) / math.sqrt(dh)
    kh = k.view(B, Lk, H, dh).transpose(1, 2)
    vh = v.view(B, Lk, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if kv_len is not None:
        s[..., kv_len:] = float("-inf")
    if mask is not None:
        s = s.masked_fill(mask[None, None], float("-inf"))
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, E)


@pytest.mark.parametrize("B,L,kv", [(2, 200, None), (1, 64, None), (3, 333, 250), (2, 1024, None)])
def test_attention_self(B, L, kv):
    O = ops()
    E, H = 256, 4
    qkv = rnd(B, L, 3 * E, seed=20).double().requires_grad_(True)
    ref = _ref_attention(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], H, kv)
    go = rnd(B, L, E, seed=21)
    ref.backward(go.double())
    x = qkv.detach().float().to(DEV).requires_grad_(True)
    o = O.attention_self(x, H, kv)
    o.backward(go.to(DEV))
    assert_close(o, ref, rel=2e-5, what="attn out")
    assert_close(x.grad, qkv.grad, rel=5e-5, what="attn dqkv")
    if kv is not None:
        assert float(x.grad[:, kv:, E:].abs().max()) == 0.0      # masked keys get exactly zero dk/dv


def test_attention_cross_17_queries():
    O = ops()
    B, Lq, Lk, E, H, kv = 2, 17, 300, 256, 4, 230
    q = rnd(B, Lq, E, seed=22).double().requires_grad_(True)
    kvt = rnd(B, Lk, 2 * E, seed=23).double().requires_grad_(True)
    ref = _ref_attention(q, kvt[..., :E], kvt[..., E:], H, kv)
    go = rnd(B, Lq, E, seed=24)
    ref.backward(go.double())
    qg = q.detach().float().to(DEV).requires_grad_(True)
    kg = kvt.detach().float().to(DEV).requires_grad_(True)
    o = O.attention_cross(qg, kg, H, kv)
    o.backward(go.to(DEV))
    assert_close(o, ref, what="cross out")
    assert_close(qg.grad, q.grad, rel=5e-5, what="cross dq")
    assert_close(kg.grad, kvt.grad, rel=5e-5, what="cross dkv")


def test_attention_spiked_scores_online_softmax():
    """one key dominates late in the sequence: forces a large running-max update (rescale path)."""
    O = ops()
    B, L, E, H = 1, 300, 256, 4
    qkv = rnd(B, L, 3 * E, seed=25)
    qkv[0, :, :E] *= 3.0
    qkv[0, 257, E:2 * E] = qkv[0, 5, :E] * 4.0
    ref = _ref_attention(qkv[..., :E].double(), qkv[..., E:2 * E].double(), qkv[..., 2 * E:].double(), H)
    o = O.attention_self(qkv.to(DEV), H)
    assert_close(o, ref, rel=5e-5, what="spiked")


def test_attention_small_masked():
    from hoisdf_amd.model import get_mano_tgt_mask
    O = ops()
    B, L, E, H = 3, 17, 256, 4
    mask = get_mano_tgt_mask()
    q, k, v = (rnd(B, L, E, seed=s).double().requires_grad_(True) for s in (26, 27, 28))
    ref = _ref_attention(q, k, v, H, mask=mask)
    go = rnd(B, L, E, seed=29)
    ref.backward(go.double())
    qg, kg, vg = (t.detach().float().to(DEV).requires_grad_(True) for t in (q, k, v))
    o = O.attention_small(qg, kg, vg, mask.to(torch.uint8).to(DEV), H)
    o.backward(go.to(DEV))
    assert_close(o, ref, what="small out")
    for a, b_, n in ((qg, q, "dq"), (kg, k, "dk"), (vg, v, "dv")):
        assert_close(a.grad, b_.grad, rel=5e-5, what="small " + n)


def test_attention_dropout_is_consistent_between_fwd_and_bwd():
    """with dropout the backward regenerates the same mask: check dV against a finite difference
    of the (deterministic-per-seed) forward."""
    O = ops()
    import hoisdf_amd.ops as OO
    B, L, E, H, p = 1, 96, 256, 4, 0.3
    qkv = rnd(B, L, 3 * E, seed=30).to(DEV)
    seed = 987654321
    o = OO._AttentionSelf.apply(qkv.clone().requires_grad_(True), H, L, p, seed)
    x = qkv.clone().requires_grad_(True)
    o = OO._AttentionSelf.apply(x, H, L, p, seed)
    go = rnd(B, L, E, seed=31).to(DEV)
    o.backward(go)
    # linear in V: o(V + dV) - o(V) = J dV exactly (same mask)
    dV = torch.zeros_like(qkv)
    dV[..., 2 * E:] = rnd(B, L, E, seed=32).to(DEV) * 0.5
    o2 = OO._AttentionSelf.apply(qkv + dV, H, L, p, seed)
    lhs = ((o2 - o.detach()) * go).sum()
    rhs = (x.grad * dV).sum()
    assert abs(float(lhs - rhs)) <= 2e-4 * abs(float(rhs)) + 1e-3, (float(lhs), float(rhs))
    # and the dropped fraction is about p: compare with the p = 0 output row sums via V = ones
    ones = qkv.clone()
    ones[..., 2 * E:] = 1.0
    od = OO._AttentionSelf.apply(ones, H, L, p, seed)
    assert abs(float(od.mean()) - 1.0) < 0.05


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,D", [(1500, 256), (1501, 256), (3, 256), (37, 64), (1500, 512)])      # (D <= 256: four rows of a wave in flight, ragged tails)
def test_add_layernorm_and_plain(M, D):
    O = ops()
    x, r = rnd(M, D, seed=40).double().requires_grad_(True), rnd(M, D, seed=41).double().requires_grad_(True)
    g, b = (1 + 0.1 * rnd(D, seed=42)).double().requires_grad_(True), rnd(D, seed=43).double().requires_grad_(True)
    ref = F.layer_norm(x + r, (D,), g, b, 1e-5)
    gy = rnd(M, D, seed=44)
    ref.backward(gy.double())
    xs = [t.detach().float().to(DEV).requires_grad_(True) for t in (x, r, g, b)]
    y = O.add_layernorm(*xs)
    y.backward(gy.to(DEV))
    assert_close(y, ref, what="ln")
    for a, c, n in zip(xs, (x, r, g, b), ("dx", "dr", "dgamma", "dbeta")):
        assert_close(a.grad, c.grad, rel=1e-4, what=n)
    y2 = O.add_layernorm(xs[0].detach(), None, xs[2].detach(), xs[3].detach())
    assert_close(y2, F.layer_norm(x, (D,), g, b, 1e-5), what="plain ln")


def test_token_build_and_vote():
    O, R = ops(), oracle()
    B, P, S, D = 2, 50, 70, 256
    cam, center = rnd(B, P, 3, seed=50), rnd(B, 3, seed=51)
    pe, feat = rnd(B, P, 30, seed=52), rnd(B, P, 223, seed=53).requires_grad_(True)
    sdf = rnd(B, P, 1, seed=54) * 0.1
    beta = torch.tensor([0.08], requires_grad=True)
    sig = torch.sigmoid(sdf / beta) / beta
    ref = torch.cat([cam - center[:, None], pe, feat * sig], 2)
    gt = rnd(B, S, D, seed=55)
    (ref * gt[:, 10:10 + P]).sum().backward()
    fg = feat.detach().to(DEV).requires_grad_(True)
    bg = beta.detach().to(DEV).requires_grad_(True)
    tok = torch.zeros(B, S, D, device=DEV)
    out = O.token_build(tok, cam.to(DEV).reshape(-1, 3), center.to(DEV), pe.to(DEV), fg, sdf.to(DEV), bg, 10)
    (out * gt.to(DEV)).sum().backward()
    assert_close(out[:, 10:10 + P], ref, what="tokens")
    assert float(out[:, :10].abs().max()) == 0.0
    assert_close(fg.grad, feat.grad, what="dfeat")
    assert_close(bg.grad, beta.grad, rel=1e-4, what="dbeta")

    L, J = 3, 20
    off, cls = rnd(L, B, P, J * 3, seed=56).requires_grad_(True), rnd(L, B, P, J, seed=57).requires_grad_(True)
    pts = rnd(B, P, 3, seed=58)
    w = torch.softmax(cls, dim=2).unsqueeze(-1)
    jref = ((pts[None, :, :, None] + off.view(L, B, P, J, 3)) * w).sum(2)
    gj = rnd(L, B, J, 3, seed=59)
    jref.backward(gj)
    og, cg = off.detach().to(DEV).requires_grad_(True), cls.detach().to(DEV).requires_grad_(True)
    j = O.vote_aggregate(og, cg, pts.to(DEV))
    j.backward(gj.to(DEV))
    assert_close(j, jref, what="joints")
    assert_close(og.grad, off.grad, what="doff")
    assert_close(cg.grad, cls.grad, rel=1e-4, what="dcls")


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bins", [16, 20, 64])          # (one chunk of 4096 lattice indices per block: 1, 2 with a ragged tail, 64)
def test_lattice_candidates_match_oracle(bins):
    O, R = ops(), oracle()
    B = 3
    _, _, meta = T.synthetic_batch(B, 8, 8, seed=60)
    if bins <= 20:
        meta["bbox_hand"] = torch.tensor([30.0, 20, 230, 240]).repeat(B, 1)
    pts, sidx, lidx, counts, offsets, _ = O.lattice_candidates(meta["mano_root"].to(DEV), meta["cam_intr"].to(DEV),
                                                               meta["bbox_hand"].to(DEV), 3.1, bins)
    lat = R.dense_lattice(bins)
    start = 0
    for b in range(B):
        keep, _ = R.lattice_bbox_mask(lat, meta["mano_root"][b], meta["cam_intr"][b], meta["bbox_hand"][b], 3.1)
        ref_idx = torch.nonzero(keep).squeeze(1)
        got_idx = lidx[start:start + counts[b]].cpu().long()
        # the strict bbox compare can flip for points whose projection lands on the box edge to 1 ulp
        sym = set(ref_idx.tolist()) ^ set(got_idx.tolist())
        assert len(sym) <= max(2, len(ref_idx) // 2000), (b, len(sym), len(ref_idx))
        assert torch.equal(pts[start:start + counts[b]].cpu(), lat[got_idx])          # coordinates bit-exact
        assert bool((got_idx[1:] > got_idx[:-1]).all())                              # ascending lattice order
        assert bool((sidx[start:start + counts[b]] == b).all())
        start += counts[b]


@pytest.mark.parametrize("n,k", [(5000, 600), (29000, 1536), (300, 300), (9000, 6144)])
def test_select_smallest_abs(n, k):
    O = ops()
    B = 3
    g = torch.Generator().manual_seed(n + k)
    counts = torch.tensor([n, max(k, n // 2), n - 7 if n - 7 >= k else n], dtype=torch.int32)
    offsets = torch.zeros(B, dtype=torch.int32)
    offsets[1:] = torch.cumsum(counts, 0)[:-1]
    total = int(counts.sum())
    vals = torch.tanh(torch.randn(total, generator=g) * 0.3)
    vals[::97] = vals[1::97][: len(vals[::97])]                   # exact ties
    sel = O.select_smallest_abs(vals.to(DEV), offsets.to(DEV), counts.to(DEV), k).cpu().long()
    for b in range(B):
        seg = vals[offsets[b]: offsets[b] + counts[b]].abs()
        order = torch.sort(seg, stable=True)[1][:k] + int(offsets[b])
        assert torch.equal(sel[b], order), f"sample {b}"


def test_errors_are_reported_not_swallowed():
    from hoisdf_amd import _lib
    O = ops()
    with pytest.raises(_lib.HoisdfError):
        _lib.call("hoisdf_linear_fwd", None, 4, None, 4, None, None, 4, 8, 4, 4, 0, 0.0, 0, None, None)
    with pytest.raises(RuntimeError):
        O.linear(torch.zeros(4, 4), torch.zeros(4, 4))               # CPU tensors: no fallback


@pytest.mark.parametrize("P", [130, 1700])          # (1700 points: four segments of 425 in separate blocks + the ordered merge)
def test_vote_loss_fused_matches_reference_loss(P):
    """K12 + JointvoteLoss reductions (common/nets/loss.py:31-56) vs the oracle's joint_vote, fwd + bwd."""
    O, R = ops(), oracle()
    L, B, J = 3, 2, 20
    pts = rnd(B, P, 3, seed=70) * 0.05
    gt = pts[:, :J] * 1000 + rnd(B, J, 3, seed=71) * 15            # some points fall inside the 40 mm radius
    off = (rnd(L, B, P, J * 3, seed=72) * 0.02).requires_grad_(True)
    cls = rnd(L, B, P, J, seed=73).requires_grad_(True)
    # oracle wants the reference's seq-first layout
    l1, l2, l3, joints = R.joint_vote(pts, off.permute(0, 2, 1, 3), cls.permute(0, 2, 1, 3), gt, 0.04)
    (0.3 * l1 + 0.7 * l2 + 0.5 * l3).backward()
    og, cg = off.detach().to(DEV).requires_grad_(True), cls.detach().to(DEV).requires_grad_(True)
    from hoisdf_amd.nets.heads import JointvoteLoss
    m1, m2, m3, mj = JointvoteLoss(0.04)(pts.to(DEV), og, cg, gt.to(DEV), batch_first=True)
    (0.3 * m1 + 0.7 * m2 + 0.5 * m3).backward()
    assert_close(mj, joints, what="joints")
    for a, b_, n in ((m1, l1, "loss_joint_3d"), (m2, l2, "loss_joint_cls"), (m3, l3, "loss_all_joint_3d")):
        assert abs(float(a) - float(b_)) <= 2e-5 * abs(float(b_)) + 1e-6, (n, float(a), float(b_))
    assert_close(og.grad, off.grad, rel=1e-4, what="doff")
    assert_close(cg.grad, cls.grad, rel=1e-4, what="dcls")


def test_edge_cases_empty_and_ragged():
    """empty batches are no-ops, ragged per-row sample indices gather from the right sample, kv_len = 1 works."""
    O, R = ops(), oracle()
    # empty linear
    y = O.linear(torch.zeros(0, 16, device=DEV), torch.zeros(8, 16, device=DEV), torch.zeros(8, device=DEV))
    assert y.shape == (0, 8)
    # ragged gather: rows of sample 1 first, then sample 0 (what sdf_infer's compaction produces per sample)
    B = 2
    pyr = T.synthetic_pyramid(B, seed=9, nonneg=False)
    levels = [v.to(DEV).permute(0, 2, 3, 1).contiguous() for v in pyr.values()]
    _, _, meta = T.synthetic_batch(B, 8, 8, seed=90)
    pts = rnd(7, 3, seed=91)
    sidx = torch.tensor([1, 1, 1, 0, 0, 1, 0], dtype=torch.int32)
    feat, _ = O.project_gather(O.PyramidNHWC(levels), pts.to(DEV), meta["mano_root"].to(DEV), meta["cam_intr"].to(DEV),
                               3.1, sample_idx=sidx.to(DEV))
    for i in range(7):
        b = int(sidx[i])
        _, grid = R.project_points(pts[i][None, None], meta["mano_root"][b:b + 1], meta["cam_intr"][b:b + 1], 3.1)
        ref = R.sample_pyramid({k: v[b:b + 1] for k, v in pyr.items()}, grid, list(pyr))
        assert_close(feat[i], ref[0, 0], what=f"ragged row {i}")
    # a single visible key: softmax is 1 on it, output = its value row
    E, H = 256, 4
    q = rnd(1, 5, E, seed=92).to(DEV)
    kv = rnd(1, 9, 2 * E, seed=93).to(DEV)
    o = O.attention_cross(q, kv, H, kv_len=1)
    assert_close(o, kv[:, :1, E:].expand(1, 5, E), rel=1e-6, what="kv_len=1")


@pytest.mark.parametrize("B,Lq,Lk,kv", [(2, 300, 300, 300), (2, 17, 700, 530), (1, 1000, 1000, 1000)])
def test_attention_f16_eval_kernel(B, Lq, Lk, kv):
    """configs[4] 'fp16 MFMA attention': f16 MFMA operands split hi + lo (3 products), f32 accumulation and softmax.
    Tolerance 1e-4 of max|o| against float64 softmax attention (the f32 kernel, 2e-5, stays the parity path)."""
    O = ops()
    E, H = 256, 4
    q = rnd(B, Lq, E, seed=41)
    k = rnd(B, Lk, E, seed=42)
    v = rnd(B, Lk, E, seed=43)
    qh, kh, vh = (t.double().view(t.shape[0], t.shape[1], H, 64).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) / 8.0
    s[..., kv:] = -float("inf")
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, E)
    O.set_attention_f16_eval(True)
    try:
        with torch.no_grad():
            kvp = torch.cat([k, v], -1).to(DEV)
            o = O.attention_cross(q.to(DEV), kvp, H, kv_len=kv)
            if Lq == Lk:
                o2 = O.attention_self(torch.cat([q, k, v], -1).to(DEV), H, kv_len=kv)
                assert_close(o2, ref, rel=1e-4, what="f16 self")
    finally:
        O.set_attention_f16_eval(False)
    assert_close(o, ref, rel=1e-4, what="f16 cross")
    o32 = O.attention_cross(q.to(DEV), kvp, H, kv_len=kv)
    assert_close(o32, ref, rel=2e-5, what="f32 path untouched")


@pytest.mark.parametrize("B,Lq,Lk,kv", [(1, 8192, 8192, 8192), (2, 17, 8192, 6144), (1, 2048, 8192, 6144)])
def test_attention_f16_eval_kernel_at_config4_size(B, Lq, Lk, kv):
    """BASELINE configs[4] at its own size (6144 + 2048 = 8192 query points): the f16-MFMA attention kernel against float64
    softmax attention with 8192 keys (self-attention of the encoder stacks) and in the decoder's 17-query form over the
    6144 hand keys.  Same 1e-4 bar as the short-sequence test; the fp64 reference is evaluated in query chunks."""
    O = ops()
    E, H = 256, 4
    q = rnd(B, Lq, E, seed=44)
    k = rnd(B, Lk, E, seed=45)
    v = rnd(B, Lk, E, seed=46)
    qh, kh, vh = (t.double().view(t.shape[0], t.shape[1], H, 64).transpose(1, 2) for t in (q, k, v))
    ref = torch.empty(B, H, Lq, 64, dtype=torch.float64)
    for i in range(0, Lq, 1024):
        s = (qh[:, :, i:i + 1024] @ kh.transpose(-1, -2)) / 8.0
        s[..., kv:] = -float("inf")
        ref[:, :, i:i + 1024] = torch.softmax(s, -1) @ vh
    ref = ref.transpose(1, 2).reshape(B, Lq, E)
    O.set_attention_f16_eval(True)
    try:
        with torch.no_grad():
            kvp = torch.cat([k, v], -1).to(DEV)
            o = O.attention_cross(q.to(DEV), kvp, H, kv_len=kv)
            if Lq == Lk:
                o2 = O.attention_self(torch.cat([q, k, v], -1).to(DEV), H, kv_len=kv)
                assert_close(o2, ref, rel=1e-4, what="f16 self 8192")
    finally:
        O.set_attention_f16_eval(False)
    assert_close(o, ref, rel=1e-4, what="f16 cross 8192")
    with torch.no_grad():
        o32 = O.attention_cross(q.to(DEV), kvp, H, kv_len=kv)
    assert_close(o32, ref, rel=2e-5, what="f32 kernel at 8192 keys")
    assert not torch.equal(o32, o)                       # the switch selected another kernel


def test_fused_adamw_matches_torch_adamw():
    """hoisdf_adamw_step vs torch.optim.AdamW (CPU, float64 reference) over 5 steps: dense, channels_last 4-D,
    odd sizes (unaligned tails), a parameter without gradient; grad_scale folds the 1/world averaging."""
    from hoisdf_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(1000, 333), (17,), (8, 5, 3, 3), (40000,), (3, 7)]
    ref_p = [torch.randn(s, generator=g, dtype=torch.float64).requires_grad_(True) for s in shapes]
    ref_unused = torch.randn(4, dtype=torch.float64).requires_grad_(True)
    gpu_p = [r.detach().float().to(DEV).requires_grad_(True) for r in ref_p]
    gpu_p[2] = gpu_p[2].detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gpu_unused = ref_unused.detach().float().to(DEV).requires_grad_(True)
    ref = torch.optim.AdamW(ref_p + [ref_unused], lr=1e-2, weight_decay=0.01)
    opt = FusedAdamW(gpu_p + [gpu_unused], lr=1e-2, weight_decay=0.01, grad_scale=0.5)
    for it in range(5):
        for r, q in zip(ref_p, gpu_p):
            gr = torch.randn(r.shape, generator=g, dtype=torch.float64)
            r.grad = gr.clone()
            gq = (gr * 2.0).float().to(DEV)                      # grad_scale = 0.5 undoes the factor 2
            q.grad = gq.contiguous(memory_format=torch.channels_last) if q.dim() == 4 else gq
        ref.step()
        opt.step()
    for r, q in zip(ref_p, gpu_p):
        assert_close(q, r, rel=3e-6, what=f"param {tuple(r.shape)}")
        assert_close(opt.state[q]["exp_avg_sq"], ref.state[r]["exp_avg_sq"], rel=3e-6, what="exp_avg_sq")
    assert torch.equal(gpu_unused.detach().cpu(), ref_unused.detach().float())      # no gradient -> untouched
    assert float(opt.state[gpu_p[0]]["step"]) == 5.0
    sd = opt.state_dict()                                         # torch layout: loads into a stock AdamW
    stock = torch.optim.AdamW(gpu_p + [gpu_unused], lr=1e-2)
    stock.load_state_dict(sd)
    assert float(stock.state[gpu_p[0]]["step"]) == 5.0


@pytest.mark.parametrize("channels_last", [False, True])
def test_aux_image_losses_match_torch(channels_last):
    """(f4) fused heat-map render + MSE + 2x BCE vs the torch formulation of main/model.py:128-143,404-422 (fwd + grad)."""
    O = ops()
    B, J, H, W, sigma = 3, 21, 128, 128, 1.25
    g = torch.Generator().manual_seed(9)
    dec = torch.rand(B, 3, H, W, generator=g) * 0.98 + 0.01
    dec[:, 0] = dec[:, 0] * 300.0
    dec[0, 1, 0, 0], dec[0, 2, 0, 1] = 0.0, 1.0                       # saturated probabilities: clamped logs / 1e-12 floor
    joints = torch.rand(B, J, 2, generator=g) * 128
    hs = (torch.rand(B, H, W, generator=g) > 0.5).float()
    osg = (torch.rand(B, H, W, generator=g) > 0.5).float()
    w = [torch.rand(B, H, W, generator=g) for _ in range(3)]
    d64 = dec.double().requires_grad_(True)
    yy, xx = torch.meshgrid(torch.arange(H).double(), torch.arange(W).double(), indexing="ij")
    jx, jy = joints.double()[:, :, 0, None, None], joints.double()[:, :, 1, None, None]
    hm = torch.exp(-(((xx - jx) / sigma) ** 2) / 2 - (((yy - jy) / sigma) ** 2) / 2).sum(1) * 255
    r_hm = (d64[:, 0] - hm) ** 2
    r_obj = F.binary_cross_entropy(d64[:, 2], osg.double(), reduction="none")
    r_hand = F.binary_cross_entropy(d64[:, 1], hs.double(), reduction="none")
    (r_hm * w[0].double()).sum().add((r_obj * w[1].double()).sum()).add((r_hand * w[2].double()).sum()).backward()
    dg = dec.to(DEV)
    if channels_last:
        dg = dg.contiguous(memory_format=torch.channels_last)
    dg.requires_grad_(True)
    l_hm, l_obj, l_hand, hmg = O.aux_image_losses(dg, joints.to(DEV), hs.to(DEV), osg.to(DEV), sigma)
    ((l_hm * w[0].to(DEV)).sum() + (l_obj * w[1].to(DEV)).sum() + (l_hand * w[2].to(DEV)).sum()).backward()
    assert_close(hmg, hm, rel=3e-6, what="heatmap")
    assert_close(l_hm, r_hm, rel=1e-5, what="mse")
    assert_close(l_obj, r_obj, rel=1e-5, what="bce obj")
    assert_close(l_hand, r_hand, rel=1e-5, what="bce hand")
    got = dg.grad.cpu().double()
    for ch in range(3):
        assert_close(got[:, ch], d64.grad[:, ch], rel=1e-5, what=f"d decoder_out[{ch}]")


def test_attention_few_queries_kernel_incl_dropout():
    """Lq <= 32 takes the few-query forward kernel (waves split the keys, merged through LDS): exact vs float64 without
    dropout for several key counts (fewer tiles than waves, ragged last tile), and with dropout the fused backward
    regenerates the same mask (linear-in-V identity)."""
    O = ops()
    import hoisdf_amd.ops as OO
    E, H = 256, 4
    # (1536 keys: three key splits of 16 tiles over separate blocks + the merge kernel; 3000 of 3100: five splits, the last one ragged)
    for B, Lq, Lk, kv in [(2, 17, 1536, 1536), (1, 1, 40, 33), (3, 32, 700, 650), (2, 5, 96, 96), (1, 17, 3100, 3000)]:
        q = rnd(B, Lq, E, seed=50)
        kvt = rnd(B, Lk, 2 * E, seed=51)
        ref = _ref_attention(q.double(), kvt[..., :E].double(), kvt[..., E:].double(), H, kv)
        o = O.attention_cross(q.to(DEV), kvt.to(DEV), H, kv)
        assert_close(o, ref, what=f"few-q {Lq}x{Lk}")
    B, Lq, Lk, p, seed = 2, 17, 1536, 0.3, 424242
    q = rnd(B, Lq, E, seed=52).to(DEV)
    kvt = rnd(B, Lk, 2 * E, seed=53).to(DEV)
    x = kvt.clone().requires_grad_(True)
    o = OO._AttentionCross.apply(q, x, H, Lk, p, seed)
    go = rnd(B, Lq, E, seed=54).to(DEV)
    o.backward(go)
    dV = torch.zeros_like(kvt)
    dV[..., E:] = rnd(B, Lk, E, seed=55).to(DEV) * 0.5
    o2 = OO._AttentionCross.apply(q, kvt + dV, H, Lk, p, seed)
    lhs, rhs = ((o2 - o.detach()) * go).sum(), (x.grad * dV).sum()
    assert abs(float(lhs - rhs)) <= 2e-4 * abs(float(rhs)) + 1e-3, (float(lhs), float(rhs))
    ones = kvt.clone()
    ones[..., E:] = 1.0
    od = OO._AttentionCross.apply(q, ones, H, Lk, p, seed)
    assert abs(float(od.mean()) - 1.0) < 0.03


def test_sdf_query_one_call_matches_the_op_chain():
    """hoisdf_sdf_query_fwd (K1-K4 in one C-ABI call, in-place concatenations, optional shared gather) against the
    differentiable op chain the model uses where gradients are needed, and against the oracle."""
    from hoisdf_amd.model import get_model
    from hoisdf_amd.config import Config
    from hoisdf_amd.nets import mano as MANO
    O, R = ops(), oracle()
    c = Config(); c.resnet_type = 18; c.apply_setting("dexycb"); c.num_samp_hand, c.num_samp_obj = 300, 100
    model = get_model("test", cfg=c, mano_layer=MANO.ManoLayer(MANO.synthetic_assets(0)), with_encoder=False)
    sd = model.state_dict()
    for k in sd:
        if not k.startswith("mano_head"):
            sd[k] = T.det_param(k, sd[k].shape)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    B, P = 2, 300
    pyr_cpu = T.synthetic_pyramid(B, seed=4)
    pyr = O.PyramidNHWC([v.to(DEV).permute(0, 2, 3, 1).contiguous() for v in pyr_cpu.values()])
    inputs, _, meta = T.synthetic_batch(B, P, 8, seed=5)
    pts = (inputs["hand_sdf_points"] * 1.2).to(DEV)
    root, K = meta["mano_root"].to(DEV), meta["cam_intr"].to(DEV)
    with torch.no_grad():
        ref_sdf, ref_raw, ref_pe, ref_cam = model._sdf_rows(pyr, pts, root, K, 3.1, "hand")
        sdf, raw, pe, feat = model._sdf_query(pyr, pts, root, K, 3.1, "hand", want_feat=True)
        assert_close(raw, ref_raw, rel=1e-5, what="raw"); assert_close(sdf, ref_sdf, rel=1e-5, what="sdf")
        assert_close(pe, ref_pe, rel=1e-6, what="pe")
        # shared gather: same rows handed back in, other field
        sdf2, raw2, _, _ = model._sdf_query(pyr, pts, root, K, 3.1, "obj", feat=feat)
        ref2 = model._sdf_rows(pyr, pts, root, K, 3.1, "obj")
        assert_close(raw2, ref2[1], rel=1e-5, what="raw obj via shared gather")
        # per-row sample index form (sdf_infer)
        sidx = torch.arange(B, device=DEV, dtype=torch.int32).repeat_interleave(P)
        sdf3, raw3, _, _ = model._sdf_query(pyr, pts.reshape(-1, 3), root, K, 3.1, "hand", sample_idx=sidx)
        assert torch.equal(raw3, raw)
    P_ = T.det_params(T.hot_path_param_shapes(992))
    o_sdf, _ = R.sdf_forward(P_, R.OracleCfg(), pyr_cpu, pts.cpu(), meta["mano_root"], meta["cam_intr"], 3.1, "hand", False)
    assert_close(sdf, o_sdf.reshape(-1), rel=2e-5, what="sdf vs oracle")
    # cached weights follow parameter updates (autograd-visible and FusedAdamW's raw-pointer update)
    with torch.no_grad():
        model.hand_sdf_decoder.linh3.bias.add_(0.05)
        raw4 = model._sdf_query(pyr, pts, root, K, 3.1, "hand")[1]
        assert float((raw4 - raw).abs().max()) > 1e-4
        assert_close(raw4, model._sdf_rows(pyr, pts, root, K, 3.1, "hand")[1], rel=1e-5, what="after update")
    # train() mode: decoder dropout is on inside the fused call as well (statistically different, finite)
    model.train()
    with torch.no_grad():
        raw5 = model._sdf_query(pyr, pts, root, K, 3.1, "hand")[1]
    assert bool(torch.isfinite(raw5).all()) and float((raw5 - raw4).abs().mean()) > 1e-3


@pytest.mark.parametrize("B,S,nq,ni,p,det", [(2, 64, None, None, 0.0, True), (4, 1024, None, 768, 0.1, False), (4, 1024, 768, 768, 0.1, False),
                                             (3, 100, 40, 17, 0.1, True), (4, 1024, 768, 768, 0.1, True)])
def test_coarse_encoder_layer_entry_equals_the_op_chain(B, S, nq, ni, p, det):
    """hoisdf_encoder_layer_fwd / _bwd (one C-ABI call per direction, csrc/layers.hip) against the op-by-op autograd node:
    the same kernels in the same order with the same dropout seeds.  Few-tile exact-f32 GEMMs split their k range and add
    with float atomics, so two runs of EITHER path differ in the last bits - and a last-bit change of a pre-activation at zero
    flips its ReLU gate (tools/dbg_enc2.py: one bitmap word, 1e-2 of the gradient's scale, 66 of 400 identical op-chain
    runs).  The small shapes therefore run in deterministic mode (order-fixed reductions, emulated attention backward - which
    also puts that branch of the entry under test) and everything must be bit-identical; the large shapes run the default
    mode: forward bit-identical, gradients to 2e-5 of each tensor's scale."""
    O = ops()
    E, F, H = 256, 1024, 4
    names = ["w_in", "b_in", "w_out", "b_out", "g1", "be1", "w1", "b1", "w2", "b2", "g2", "be2", "g3", "be3"]
    shapes = [(3 * E, E), (3 * E,), (E, E), (E,), (E,), (E,), (F, E), (F,), (E, F), (E,), (E,), (E,), (E,), (E,)]
    x0 = rnd(B, S, E, seed=11)
    P0 = [rnd(*s, seed=20 + i) * (1.0 / math.sqrt(s[-1]) if len(s) == 2 else 0.1) + (1.0 if n in ("g1", "g2", "g3") else 0.0)
          for i, (n, s) in enumerate(zip(names, shapes))]
    nqe = S if nq is None else nq
    nie = nqe if ni is None else ni
    gx2, gy = rnd(B, nqe, E, seed=5).to(DEV), rnd(B, nie, E, seed=6).to(DEV)
    res = {}
    keep, keep_det = O._ENCODER_LAYER_C, O.deterministic()
    O.set_deterministic(det)
    try:
        for coarse in (True, False):
            O._ENCODER_LAYER_C = coarse
            O.manual_seed(1234)
            x = x0.to(DEV).requires_grad_(True)
            P = [t.to(DEV).requires_grad_(True) for t in P0]
            picked = O._EncoderLayerC if O._coarse_layer_ok(p, x, P[0], P[2], P[6], P[8]) else O._EncoderLayer
            assert (picked is O._EncoderLayerC) == coarse
            x2, y = O.encoder_layer(x, nq, p, H, *P, eps=1e-5, n_inter=ni)
            ((x2 * gx2).sum() + (y * gy).sum()).backward()
            res[coarse] = [x2.detach(), y.detach(), x.grad] + [t.grad for t in P]
    finally:
        O._ENCODER_LAYER_C = keep
        O.set_deterministic(keep_det)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for what, a, b in zip(["dx"] + ["d" + n for n in names], res[True][2:], res[False][2:]):
        if det:
            assert torch.equal(a, b), what
        else:
            assert_close(a, b, rel=2e-5, what=what)


@pytest.mark.parametrize("B,S,kv,p", [(2, 64, 48, 0.0), (3, 700, 600, 0.1), (4, 2048, 1536, 0.1)])
def test_coarse_decoder_layer_entry_equals_the_op_chain(B, S, kv, p):
    """hoisdf_decoder_layer_fwd / _bwd (csrc/layers.hip) against the op-by-op decoder stack of nets/blocks.py: same kernels, same
    dropout seeds.  Deterministic mode (order-fixed kernels): forward bit-identical; the gradients differ only by the order in which
    the branches of a multi-consumer tensor are added (autograd's accumulation order vs the entry's epilogue adds): 1e-5 of scale."""
    from hoisdf_amd.nets import blocks as BL
    O = ops()
    torch.manual_seed(3)
    dec = BL.TransformerDecoder(256, 4, 2, 1024, p).to(DEV).train()
    for prm in dec.parameters():
        if prm.dim() > 1:
            torch.nn.init.xavier_uniform_(prm)
        else:
            torch.nn.init.normal_(prm, 1.0 if prm.shape[0] == 256 and prm.abs().max() > 0.5 else 0.0, 0.1)
    Q = 17
    mask = (torch.rand(Q, Q) < 0.2)
    mask[torch.arange(Q), torch.arange(Q)] = False
    mask_u8 = mask.to(torch.uint8).to(DEV)
    mem0, qe0 = rnd(B, S, 256, seed=8), rnd(Q, 256, seed=9) * 0.5
    g = rnd(2, B, Q, 256, seed=10).to(DEV)
    res = {}
    keep, keep_det = O._DECODER_LAYER_C, O.deterministic()
    O.set_deterministic(True)
    try:
        for coarse in (True, False):
            O._DECODER_LAYER_C = coarse
            O.manual_seed(77)
            mem = mem0.to(DEV).requires_grad_(True)
            qe = qe0.to(DEV).requires_grad_(True)
            dec.zero_grad(set_to_none=True)
            assert O.decoder_layer_ok(p, mem, qe) == coarse
            hs = dec(mem, qe, mask_u8, kv)
            (hs * g).sum().backward()
            res[coarse] = [hs.detach(), mem.grad.clone(), qe.grad.clone()] + [prm.grad.clone() for prm in dec.parameters()]
    finally:
        O._DECODER_LAYER_C = keep
        O.set_deterministic(keep_det)
    assert torch.equal(res[True][0], res[False][0])
    names = ["d memory", "d query_embed"] + ["d " + n for n, _ in dec.named_parameters()]
    for what, a, b in zip(names, res[True][1:], res[False][1:]):
        assert_close(a, b, rel=1e-5, what=what)


@pytest.mark.parametrize("B,P", [(2, 300), (2, 1536)])
def test_coarse_sdf_query_train_entries_equal_the_op_chain(B, P):
    """hoisdf_sdf_query_train_fwd / hoisdf_sdf_query_bwd (one call per direction) against the op-by-op training-time SDF query of
    Model._sdf_rows (gather, linear_sdfin, posenc, weight-normed decoder, head; dropout off so that both draw no seeds): clamped
    sdf, positional encoding, camera points, every parameter gradient and the pyramid gradient."""
    from hoisdf_amd.model import get_model
    from hoisdf_amd.config import Config
    from hoisdf_amd.nets import mano as MANO
    O = ops()
    c = Config(); c.resnet_type = 18; c.apply_setting("dexycb"); c.num_samp_hand, c.num_samp_obj = 64, 32
    model = get_model("test", cfg=c, mano_layer=MANO.ManoLayer(MANO.synthetic_assets(0)), with_encoder=False)
    sd = model.state_dict()
    for k in sd:
        if not k.startswith("mano_head"):
            sd[k] = T.det_param(k, sd[k].shape)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    inputs, _, meta = T.synthetic_batch(B, P, 8, seed=5)
    pts = (inputs["hand_sdf_points"] * 1.2).to(DEV)
    root, K = meta["mano_root"].to(DEV), meta["cam_intr"].to(DEV)
    gs = rnd(B * P, seed=3).to(DEV)
    res = {}
    keep, keep_det = O._SDF_QUERY_TRAIN_C, O.deterministic()
    O.set_deterministic(True)
    try:
        for coarse in (True, False):
            O._SDF_QUERY_TRAIN_C = coarse
            model.zero_grad(set_to_none=True)
            levels = [v.to(DEV).permute(0, 2, 3, 1).contiguous().requires_grad_(True) for v in T.synthetic_pyramid(B, seed=4).values()]
            pyr = O.PyramidNHWC(levels)
            sdf, raw, pe, cam = model._sdf_rows(pyr, pts, root, K, 3.1, "hand")
            assert (raw is None) == coarse                              # the coarse path does not hand the unclamped value back
            (sdf * gs).sum().backward()
            named = [(n, p.grad.clone()) for n, p in model.named_parameters() if p.grad is not None]
            res[coarse] = (sdf.detach(), pe, cam, [lv.grad.clone() for lv in levels], named)
    finally:
        O._SDF_QUERY_TRAIN_C = keep
        O.set_deterministic(keep_det)
    a, b = res[True], res[False]
    assert_close(a[0], b[0], rel=5e-6, what="sdf")
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for i, (x, y) in enumerate(zip(a[3], b[3])):
        assert_close(x, y, rel=2e-5, what=f"d pyramid level {i}")
    assert [n for n, _ in a[4]] == [n for n, _ in b[4]] and len(a[4]) == 18
    for (n, x), (_, y) in zip(a[4], b[4]):
        assert_close(x, y, rel=2e-5, what="d " + n)


# ---------------------------------------------------------------------------------------------
# a15: the scalar point losses (hoisdf_point_loss_fwd / _bwd) against torch in float64
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 257, 32 * 1536, 600_001])
def test_l1_loss_with_clamped_target_matches_torch(n):
    """SepSDFLoss on the clamped ground truth (common/nets/loss.py:64-78, main/model.py:393-400): value 1e-6 rel (an
    order-fixed f32 sum against float64), gradient exact (+-1/n, 0 where pred == target)."""
    O = ops()
    g = torch.Generator().manual_seed(n)
    pred = (torch.rand(n, 1, generator=g) - 0.5) * 0.1
    gt = (torch.rand(n, generator=g) - 0.5) * 0.3                      # beyond the clamp on both sides
    cd = 0.05
    if n > 4:
        pred[3, 0] = gt[3].clamp(-cd, cd)                              # an exact tie: torch's sign(0) = 0
    p64 = pred.double().requires_grad_(True)
    ref = F.l1_loss(p64, gt.clamp(-cd, cd).double().unsqueeze(-1))       # (clamped in float32, as the model's targets are)
    (ref * 3.0).backward()
    pg = pred.to(DEV).requires_grad_(True)
    got = O.l1_loss_clamped_target(pg, gt.to(DEV), cd)
    (got * 3.0).backward()
    assert abs(float(got) - float(ref)) <= 1e-6 * abs(float(ref)) + 1e-12
    assert_close(pg.grad, p64.grad, rel=1e-6, what="l1 grad")          # +-3/n (one float rounding of 1/n apart at most)
    if n > 4:
        assert float(pg.grad[3, 0]) == 0.0                              # the tie
    again = O.l1_loss_clamped_target(pg.detach(), gt.to(DEV), cd)
    assert float(again) == float(got)                                 # order-fixed: bit-reproducible


@pytest.mark.parametrize("L,B,P,C,scale", [(6, 3, 5, 3, 1.0), (6, 32, 512, 3, 1.0), (6, 2, 1, 60, 1000.0), (1, 1, 1, 1, 1.0)])
def test_smooth_l1_loss_broadcast_matches_torch(L, B, P, C, scale):
    """obj_rot / obj_trans (main/model.py:656-662: (L, B, P, 3) against (B, 3)) and loss_all_joint_3d (loss.py:57-59:
    joints * 1000 against (B, J * 3)): value 2e-6 rel, gradient 1e-6 of its max."""
    O = ops()
    g = torch.Generator().manual_seed(L * 1000 + P)
    pred = torch.randn(L, B, P, C, generator=g) * (2.0 / scale)        # both SmoothL1 branches
    tgt = torch.randn(B, C, generator=g)
    p64 = pred.double().requires_grad_(True)
    ref = F.smooth_l1_loss(p64 * scale, tgt.double()[None, :, None].expand(L, B, P, C))
    ref.backward()
    pg = pred.to(DEV).requires_grad_(True)
    got = O.smooth_l1_loss_broadcast(pg, tgt.to(DEV), P, scale)
    got.backward()
    assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-12
    assert_close(pg.grad, p64.grad, rel=1e-6, what="smooth-l1 grad")


def test_point_loss_rejects_shapes_that_are_not_a_broadcast():
    O = ops()
    with pytest.raises(ValueError):
        O.smooth_l1_loss_broadcast(torch.zeros(2, 3, 5, 3, device=DEV), torch.zeros(4, 3, device=DEV), 5)


def test_aux_image_losses_match_the_reference_fixture():
    """(f4) hoisdf_aux_image_losses_fwd / _bwd against tests/golden/g13_aux_losses.npz = the reference's own forward
    (main/model.py:128-143,404-422) on the seeded decoder output: maps on every second pixel, exact means, the gradient of
    the sum of the three means w.r.t. the decoder output."""
    from conftest import load_golden
    from hoisdf_amd.config import Config
    O = ops()
    g = load_golden("g13_aux_losses")
    dec = T.synthetic_decoder_out(2, seed=13).to(DEV).requires_grad_(True)
    _, targets, _ = T.synthetic_batch(2, 48, 16, seed=31)
    l_hm, l_obj, l_hand, hm = O.aux_image_losses(dec, targets["joint_coord"].to(DEV), targets["hand_seg"].to(DEV),
                                                 targets["obj_seg"].to(DEV), Config().sigma)
    sub = lambda t: t.detach()[..., ::2, ::2]
    assert_close(sub(hm), g["heatmap"], rel=3e-6, what="heatmap")
    assert_close(sub(l_hm), g["joint_heatmap"], rel=1e-5, what="mse map")
    assert_close(sub(l_obj), g["obj_seg"], rel=1e-5, what="bce obj map")
    assert_close(sub(l_hand), g["hand_seg"], rel=1e-5, what="bce hand map")
    for got, k in ((l_hm, "joint_heatmap"), (l_obj, "obj_seg"), (l_hand, "hand_seg")):
        ref = float(g["mean_" + k])
        assert abs(float(got.detach().double().mean()) - ref) <= 2e-6 * abs(ref), k
    (l_hm.mean() + l_obj.mean() + l_hand.mean()).backward()
    assert_close(sub(dec.grad), g["grad_decoder_out"], rel=1e-5, what="d decoder_out")
    assert abs(float(dec.grad.double().norm()) - float(g["grad_norm"])) <= 1e-5 * float(g["grad_norm"])


# ---------------------------------------------------------------------------------------------
# K7 + K8 and K11 + K12 as single C calls against the op-by-op chains they replace
# ---------------------------------------------------------------------------------------------
def _mlp_params(dims, seed):
    ws = [(rnd(dims[i + 1], dims[i], seed=seed + i) / math.sqrt(dims[i])).to(DEV).requires_grad_(True) for i in range(len(dims) - 1)]
    bs = [rnd(dims[i + 1], seed=seed + 50 + i, scale=0.1).to(DEV).requires_grad_(True) for i in range(len(dims) - 1)]
    return ws, bs


@pytest.mark.parametrize("B,P,C", [(2, 96, 992), (3, 1100, 992), (1, 40, 3968)])
def test_coarse_tokens_entry_equals_the_op_chain(B, P, C):
    """hoisdf_tokens_fwd / _bwd (linear_transformerin + sigma gate + token rows, one call per direction) against ops.linear x 4 +
    ops.token_build: token rows, the detached MLP output, and the gradients of the gathered rows, the eight MLP parameters and
    beta (deterministic mode: bit-identical; 3 x 1100 rows crosses the 2048-row switch to the emulated linear layers)."""
    O = ops()
    S, D, row0 = P + 24, 256, 0
    ws, bs = _mlp_params([C, 1024, 512, 256, D - 33], seed=60)
    feat = rnd(B * P, C, seed=7).to(DEV)
    cam, center = rnd(B * P, 3, seed=8).to(DEV), rnd(B, 3, seed=9).to(DEV)
    pe, sdf = rnd(B * P, 30, seed=10).to(DEV), rnd(B * P, seed=11, scale=0.05).to(DEV)
    gtok = rnd(B, S, D, seed=12).to(DEV)
    keep_det = O.deterministic()
    O.set_deterministic(True)
    res = {}
    try:
        for coarse in (True, False):
            for t in ws + bs:
                t.grad = None
            f = feat.clone().requires_grad_(True)
            beta = torch.full((1,), 0.07, device=DEV, requires_grad=True)
            tok = torch.zeros(B, S, D, device=DEV)
            if coarse:
                tok, fea = O.tokens(tok, f, cam, center, pe, sdf, beta, row0, ws, bs)
            else:
                h = f
                for w_, b_ in zip(ws, bs):
                    h = O.linear(h, w_, b_, act=True)
                fea = h.detach()
                tok = O.token_build(tok, cam, center, pe, h, sdf, beta, row0)
            (tok * gtok).sum().backward()
            res[coarse] = [tok.detach(), fea, f.grad, beta.grad] + [t.grad.clone() for t in ws + bs]
    finally:
        O.set_deterministic(keep_det)
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        assert torch.equal(a, b), f"output / gradient {i} differs: {float((a - b).abs().max()):.3e}"


@pytest.mark.parametrize("L,B,P", [(6, 2, 48), (3, 2, 700)])
def test_coarse_heads_vote_entry_equals_the_op_chain(L, B, P):
    """hoisdf_heads_vote_fwd / _bwd (vote + class MLPs on all depths + vote aggregation + JointvoteLoss sums) against ops.linear x 7
    + ops.vote_loss: joints, the three reductions, the gradient of the encoder rows and of the 14 MLP parameters."""
    O = ops()
    E, J = 256, 20
    vw, vb = _mlp_params([E, E, E, E, 3 * J], seed=70)
    cw, cb = _mlp_params([E, E, E, J], seed=80)
    enc0 = rnd(L, B, P, E, seed=13).to(DEV)
    pts = (rnd(B, P, 3, seed=14) * 0.05).to(DEV)
    gt = (pts[:, :J] * 1000 + rnd(B, J, 3, seed=15).to(DEV) * 10).contiguous()
    gj, gl, gb = rnd(L, B, J, 3, seed=16).to(DEV), rnd(L, B, seed=17).to(DEV), rnd(L, B, seed=18).to(DEV)
    keep_det = O.deterministic()
    O.set_deterministic(True)
    res = {}
    try:
        for coarse in (True, False):
            for t in vw + vb + cw + cb:
                t.grad = None
            enc = enc0.clone().requires_grad_(True)
            if coarse:
                joints, l3d, bce, near = O.heads_vote(enc, pts, gt, 0.04, vw, vb, cw, cb)
            else:
                h = enc
                for i, (w_, b_) in enumerate(zip(vw, vb)):
                    h = O.linear(h, w_, b_, act=i < 3)
                g_ = enc
                for i, (w_, b_) in enumerate(zip(cw, cb)):
                    g_ = O.linear(g_, w_, b_, act=i < 2)
                joints, l3d, bce, near = O.vote_loss(h, g_, pts, gt, 0.04)
            ((joints * gj).sum() + (l3d * gl).sum() + (bce * gb).sum()).backward()
            res[coarse] = [joints.detach(), l3d.detach(), bce.detach(), near, enc.grad] + [t.grad.clone() for t in vw + vb + cw + cb]
    finally:
        O.set_deterministic(keep_det)
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        assert torch.equal(a, b), f"output / gradient {i} differs: {float((a - b).abs().max()):.3e}"

@pytest.mark.parametrize("M,D,p", [(700, 256, 0.0), (4096, 256, 0.1), (33, 64, 0.3)])
def test_residual_dropout_matches_the_add_layernorm_mask(M, D, p):
    """hoisdf_residual_dropout (the residual of the pre-norm layers, common/nets/transformer.py:304-331): y = x + dropout(r) with the
    (seed, row, column) mask of hoisdf_add_layernorm_fwd - checked through LayerNorm of the same sum - and dr = mask(dy), dx = dy."""
    O = ops()
    import hoisdf_amd.ops as OO
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g).to(DEV).requires_grad_(True)
    r = torch.randn(M, D, generator=g).to(DEV).requires_grad_(True)
    gamma, beta = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    seed = 991
    y = OO._ResidualDropout.apply(x, r, p, seed)
    ln = OO._AddLayerNorm.apply(x.detach(), r.detach(), gamma, beta, 1e-5, p, seed)          # LN(x + dropout(r)), same mask
    assert_close(ln, torch.nn.functional.layer_norm(y.detach().double(), (D,)).float(), rel=2e-5, what="same mask as add_layernorm")
    keep = ((y.detach() - x.detach()).abs() > 0) | (r.detach() == 0)
    if p == 0.0:
        assert torch.equal(y.detach(), x.detach() + r.detach())
    else:
        frac = 1.0 - float(keep.float().mean())
        assert abs(frac - p) < 0.03, frac
        assert_close((y.detach() - x.detach())[keep], r.detach()[keep] / (1.0 - p), rel=2e-6, what="kept elements scaled")
    go = torch.randn(M, D, generator=g).to(DEV)
    y.backward(go)
    assert torch.equal(x.grad, go)
    assert_close(r.grad, torch.where(keep, go / (1.0 - p), torch.zeros_like(go)) if p > 0 else go, rel=2e-6, what="dr")
