"""-m gpu: (f2) device-side SDF sample selection against the semantics of data/dexycb.py:514-546
(uniform draws without replacement per region, |sdf| < dist pre-filter in training, output row order)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_frames(n_frames=5, seed=0):
    rng = np.random.default_rng(seed)
    frames, index = [], []
    for f in range(n_frames):
        nh, no = int(rng.integers(300, 500)), int(rng.integers(200, 400))
        a = rng.standard_normal((nh + no, 6)).astype(np.float32)
        a[:, 3] = rng.uniform(-0.2, 0.2, nh + no)          # sdf_hand
        a[:, 4] = rng.uniform(-0.2, 0.2, nh + no)          # sdf_obj
        a[:, 5] = rng.integers(0, 6, nh + no)
        frames.append(a)
        index.append([nh, no])
    return frames, np.array(index)


def test_store_roundtrip_from_directory(tmp_path):
    from hoisdf_amd.sdf_data import SdfStore
    frames, index = make_frames(3)
    os.makedirs(tmp_path / "sdf_processed")
    for i, a in enumerate(frames):
        np.save(tmp_path / "sdf_processed" / f"f{i:03d}.npy", a)
    np.save(tmp_path / "sdf_index.npy", index)
    st = SdfStore.from_directory(str(tmp_path))
    assert st.n_frames == 3 and st.rows.shape == (index.sum(), 6)
    assert torch.equal(st.rows.cpu(), torch.from_numpy(np.concatenate(frames)))


@pytest.mark.parametrize("train", [False, True])
def test_draws_respect_regions_filters_and_order(train):
    from hoisdf_amd.sdf_data import SdfStore
    frames, index = make_frames(5, seed=1)
    st = SdfStore(frames, index)
    fid = [4, 0, 2, 2]
    nh, no, dist = 64, 48, 0.05
    out = st.sample(fid, nh, no, dist, train, seed=7)
    n = (nh + no) * (2 if train else 1)
    assert out["sdf_points"].shape == (4, n, 5) and out["sdf_raw_label"].shape == (4, n)
    rows = out["rows"].cpu().numpy()
    row0 = np.concatenate([[0], np.cumsum(index.sum(1))])
    for b, f in enumerate(fid):
        local = rows[b] - row0[f]
        h, o = local[:nh], local[nh:nh + no]
        assert len(set(h)) == nh and h.min() >= 0 and h.max() < index[f, 0]                 # hand region, no repeats
        assert len(set(o)) == no and o.min() >= index[f, 0] and o.max() < index[f].sum()    # obj region
        if train:
            hp, op = local[nh + no:2 * nh + no], local[2 * nh + no:]
            assert len(set(hp)) == nh and np.all(np.abs(frames[f][hp, 3]) < dist) and hp.max() < index[f, 0]
            assert len(set(op)) == no and np.all(np.abs(frames[f][op, 4]) < dist) and op.min() >= index[f, 0]
        np.testing.assert_array_equal(out["sdf_points"][b].cpu().numpy(), frames[f][local, :5])
        np.testing.assert_array_equal(out["sdf_raw_label"][b].cpu().numpy(), frames[f][local, 5])
    # the same frame twice in a batch gets independent draws; another seed gives another draw
    assert not np.array_equal(rows[2], rows[3])
    assert not np.array_equal(rows, st.sample(fid, nh, no, dist, train, seed=8)["rows"].cpu().numpy())
    assert np.array_equal(rows, st.sample(fid, nh, no, dist, train, seed=7)["rows"].cpu().numpy())


def test_draws_are_uniform():
    """each of N rows is drawn with probability k/N: chi-square over 400 seeds, p > 1e-4"""
    from scipy import stats
    from hoisdf_amd.sdf_data import SdfStore
    rng = np.random.default_rng(3)
    a = rng.standard_normal((100, 6)).astype(np.float32)
    st = SdfStore([a], np.array([[60, 40]]))
    counts = np.zeros(60)
    first = np.zeros(60)
    for seed in range(400):
        r = st.sample([0], 15, 10, 0.05, False, seed=seed)["rows"][0, :15].cpu().numpy()
        counts[r] += 1
        first[r[0]] += 1
    assert stats.chisquare(counts).pvalue > 1e-4          # membership uniform
    assert stats.chisquare(first).pvalue > 1e-4           # order uniform too (first element)


def test_too_few_eligible_rows_raises():
    from hoisdf_amd.sdf_data import SdfStore
    a = np.ones((50, 6), np.float32)                       # |sdf| = 1 everywhere: nothing passes the 0.05 filter
    st = SdfStore([a], np.array([[30, 20]]))
    st.sample([0], 8, 8, 0.05, False, seed=0)
    with pytest.raises(ValueError):
        st.sample([0], 8, 8, 0.05, True, seed=0)
    with pytest.raises(ValueError):
        st.sample([0], 31, 8, 0.05, False, seed=0)


def test_make_inputs_applies_flip_rotation_centre_and_scale():
    """(f2 hand-off) SdfStore.make_inputs vs a numpy restatement of data/dexycb.py:548-549 (flip), :288 (rotation),
    :593-620 (centre, scale) applied to the very rows the draw selected."""
    from hoisdf_amd.sdf_data import SdfStore
    frames, index = make_frames(4, seed=3)
    st = SdfStore(frames, index)
    allrows = np.concatenate(frames)
    B, nh, no = 4, 64, 32
    g = torch.Generator().manual_seed(0)
    root, oc = torch.randn(B, 3, generator=g) * 0.1, torch.randn(B, 3, generator=g) * 0.1
    flip = torch.tensor([True, False, True, False])
    ang = torch.tensor([0.3, -0.2, 0.0, 0.5])
    rot = torch.zeros(B, 3, 3)
    rot[:, 0, 0], rot[:, 0, 1], rot[:, 1, 0], rot[:, 1, 1], rot[:, 2, 2] = ang.cos(), -ang.sin(), ang.sin(), ang.cos(), 1.0
    out = st.make_inputs([0, 1, 2, 3], root, oc, nh, no, 0.15, 3.1, 2.9, train=True, seed=5, do_flip=flip, rot_mat=rot)
    rows = out["rows"].cpu().numpy()
    assert rows.shape == (B, 2 * (nh + no))
    for b in range(B):
        d = allrows[rows[b]].copy()
        if flip[b]:
            d[:, 0] *= -1
        d[:, :3] = d[:, :3].dot(rot[b].numpy().T)
        hand, obj = d[:nh].copy(), d[nh:nh + no].copy()
        hand[:, :3] -= root[b].numpy(); obj[:, :3] -= oc[b].numpy()
        hand *= 3.1; obj *= 2.9
        np.testing.assert_allclose(out["hand_sdf_points"][b].cpu().numpy(), hand[:, :3], atol=2e-6)
        np.testing.assert_allclose(out["obj_sdf_points"][b].cpu().numpy(), obj[:, :3], atol=2e-6)
        np.testing.assert_allclose(out["hand_sdf"][b].cpu().numpy(), hand[:, 3], atol=2e-6)
        np.testing.assert_allclose(out["obj_sdf"][b].cpu().numpy(), obj[:, 4], atol=2e-6)
        pre_h = (d[nh + no:2 * nh + no, :3] - root[b].numpy()) * 3.1
        pre_o = (d[2 * nh + no:, :3] - oc[b].numpy()) * 2.9
        np.testing.assert_allclose(out["hand_pre_points"][b].cpu().numpy(), pre_h, atol=2e-6)
        np.testing.assert_allclose(out["obj_pre_points"][b].cpu().numpy(), pre_o, atol=2e-6)
        assert (np.abs(allrows[rows[b, nh + no:2 * nh + no], 3]) < 0.15).all()      # the |sdf_hand| < dist pre-filter


def test_trainer_draws_its_batches_from_the_store():
    """Trainer(sdf_store=...): the dataset yields frame ids + augmentation, the query points come from HBM."""
    from hoisdf_amd.config import Config
    from hoisdf_amd.engine import Trainer
    from hoisdf_amd.sdf_data import synthetic_store
    c = Config()
    c.resnet_type = 18
    c.apply_setting("dexycb")
    c.num_samp_hand, c.num_samp_obj = 96, 32
    dev = torch.device("cuda", 0)
    store = synthetic_store(6, rows_hand=800, rows_obj=500)
    tr = Trainer(c, dev, batch_size=2, tune_encoder=False, sdf_store=store)
    it = iter(tr.batch_generator)
    losses = []
    for _ in range(2):
        inputs, targets, meta = next(it)
        assert "hand_sdf_points" not in inputs and "sdf_frame" in meta
        total, loss = tr.train_step(inputs, targets, meta, 0, 0.0)
        losses.append(float(total))
    assert all(np.isfinite(losses)) and "sdfhand_loss" in loss and float(loss["sdfhand_loss"]) > 0


def test_sampler_against_the_references_own_selection_code():
    """(f2) tests/golden/g14_sampler.npz = data/dexycb.py:514-549,:288,:596-617 executed on the synthetic frames.  The device
    sampler draws from another random stream, so it is held to what IS comparable: the eligibility COUNTS of the |sdf| < dist
    pre-filter (hoisdf_sdf_sample_keys' `elig`) and the sets its draws may come from; and numpy's own draws, fed through the
    device hand-off (flip, in-plane rotation, centre, scale), must reproduce the reference's outputs (1e-6: the reference
    rotates in float64 and rounds once)."""
    from conftest import load_golden
    from hoisdf_amd import testing as T
    from hoisdf_amd.sdf_data import SdfStore
    g = load_golden("g14_sampler")
    frames, index = T.synthetic_sdf_frames(4, seed=14)
    st = SdfStore(frames, index)
    row0 = np.concatenate([[0], np.cumsum(index.sum(1))])
    nh, no, dist, sc = 64, 48, 0.05, 3.1
    B = len(frames)
    for mode in ("train", "test"):
        train = mode == "train"
        key = lambda i, n: g[f"{mode}{i}.{n}"]
        rows = np.stack([np.asarray(key(i, "all_idx")) + row0[i] for i in range(B)])
        root = torch.stack([key(i, "hand_root") for i in range(B)]).float()
        oc = torch.stack([key(i, "obj_center_cam") for i in range(B)]).float()
        flip = torch.tensor([i % 2 == 1 for i in range(B)])
        rot = torch.stack([key(i, "rot_mat").float() for i in range(B)]) if train else None
        out = st.make_inputs(list(range(B)), root, oc, nh, no, dist, sc, sc, train=train, seed=0, do_flip=flip, rot_mat=rot,
                             rows=torch.from_numpy(rows))
        for i in range(B):
            hp, op = key(i, "hand_sdf_points").numpy(), key(i, "obj_sdf_points").numpy()        # (n, 5) scaled rows
            np.testing.assert_allclose(out["hand_sdf_points"][i].cpu().numpy(), hp[:, :3], atol=1e-6)
            np.testing.assert_allclose(out["obj_sdf_points"][i].cpu().numpy(), op[:, :3], atol=1e-6)
            np.testing.assert_allclose(out["hand_sdf"][i].cpu().numpy(), hp[:, 3], atol=1e-7)
            np.testing.assert_allclose(out["obj_sdf"][i].cpu().numpy(), op[:, 4], atol=1e-7)
            if train:
                np.testing.assert_allclose(out["hand_pre_points"][i].cpu().numpy(), key(i, "hand_pre_points").numpy(), atol=1e-6)
                np.testing.assert_allclose(out["obj_pre_points"][i].cpu().numpy(), key(i, "obj_pre_points").numpy(), atol=1e-6)
        # the device draw: its pre points come from the reference's eligibility sets, and a k one above the set size is refused
        # exactly where np.random.choice(replace=False) would raise
        if train:
            smp = st.sample(list(range(B)), nh, no, dist, True, seed=3)["rows"].cpu().numpy()
            for i in range(B):
                eh, eo = set(np.asarray(key(i, "elig_hand")) + row0[i]), set(np.asarray(key(i, "elig_obj")) + row0[i])
                assert set(smp[i, nh + no:2 * nh + no]) <= eh and set(smp[i, 2 * nh + no:]) <= eo
            n_min_h = min(len(np.asarray(key(i, "elig_hand"))) for i in range(B))
            st.sample(list(range(B)), n_min_h, no, dist, True, seed=3)
            with pytest.raises(ValueError):
                st.sample(list(range(B)), n_min_h + 1, no, dist, True, seed=3)
