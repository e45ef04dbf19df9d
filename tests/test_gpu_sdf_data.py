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
