"""hoisdf_amd/metrics.py (batched, device-side) against g11_metrics.npz = the REFERENCE's own metric functions run on
seeded inputs (tests/golden/make_golden.py metrics_golden: common/metrics.py, common/eval_util.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from hoisdf_amd import metrics as M


def _run(dev):
    g = load_golden("g11_metrics")
    t = lambda k: g[k].to(dev) if torch.is_tensor(g[k]) else torch.from_numpy(np.asarray(g[k])).to(dev)
    mje, pamje = M.eval_hand_joint(t("pred_j"), t("gt_j"))
    assert abs(mje - float(g["mje"])) < 1e-7 and abs(pamje - float(g["pamje"])) < 1e-6
    assert (M.rigid_align(t("pred_j"), t("gt_j")).cpu() - g["aligned"]).abs().max().item() < 1e-5
    tv = t("templates")[torch.from_numpy(g["obj_cls"]).long() - 1]
    o = M.obj_metrics(t("obj_rot"), t("obj_trans"), t("obj_rot_gt"), t("obj_trans_gt"), tv, ho3d=False)
    for k, ref in (("ADDS", "adds"), ("MCE", "mce"), ("OCE", "oce")):
        assert abs(o[k] - float(g[ref])) < 2e-6, (k, o[k], float(g[ref]))
    o = M.obj_metrics(t("obj_rot"), t("obj_trans"), t("obj_rot_gt"), t("obj_trans_gt"), tv, ho3d=True)
    assert abs(o["ADDS"] - float(g["adds_ho3d"])) < 2e-6 and abs(o["MME"] - float(g["mme_ho3d"])) < 2e-6
    ev = M.MeshEval()
    ev.feed(t("gt_v")[:3], t("pr_v")[:3])
    ev.feed(t("gt_v")[3:], t("pr_v")[3:])
    m3d, med, auc, pck, th = ev.get_measures(0.0, 0.05, 100)
    assert abs(m3d - float(g["mesh_mean"])) < 1e-8 and abs(med - float(g["mesh_median"])) < 1e-8
    assert abs(auc - float(g["mesh_auc"])) < 1e-6 and np.abs(pck - np.asarray(g["mesh_pck"])).max() < 1e-6
    fs = torch.stack([M.fscore(t("gt_v"), t("pr_v"), th_) for th_ in (0.005, 0.015)], 1).cpu().numpy()
    assert np.abs(fs - np.asarray(g["fscore_bruteforce"])).max() < 1e-6
    return ev, fs


def test_metrics_match_reference_cpu(tmp_path):
    ev, fs = _run("cpu")
    # results.txt in the reference's "key :  value" layout + the mesh / F-score blocks (main/test.py:229-261)
    p = os.path.join(tmp_path, "results.txt")
    M.write_results(p, {"ADDS_error": 12.0, "mano_mje": 6.0}, 6, mesh=(ev, ev), fscores=(fs.T, fs.T, [0.005, 0.015]))
    lines = open(p).read().splitlines()
    assert lines[0] == "ADDS_error :  2.0" and lines[1] == "mano_mje :  1.0"
    assert lines[2] == "Evaluation 3D MESH results:" and lines[3].startswith("auc=0.809, mean_vert3d_avg=0.96 cm")
    assert "F-scores" in lines and any(l.startswith("F@5.0mm = ") and "\tF_aligned@5.0mm = " in l for l in lines)
    q = os.path.join(tmp_path, "pred_mano.json")
    M.dump_pred_mano(q, [np.zeros((21, 3))] * 2, [np.ones((778, 3))] * 2)
    d = json.load(open(q))
    assert len(d) == 2 and len(d[0]) == 2 and len(d[0][0]) == 21 and len(d[1][1]) == 778


@pytest.mark.gpu
def test_metrics_match_reference_on_device():
    _run("cuda")
