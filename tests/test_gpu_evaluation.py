"""§8(f) rank 1 — evaluation-harness metrics on the device path (xpoint_amd/evaluation.py) against goldens produced by the
REFERENCE's own benchmark_evaluation.py functions (oracle/refharness/make_golden.py: gen_g13) on the same synthetic heat
maps / descriptor maps / homographies (xpoint_amd.synth.make_eval_case).  Counts are exact; float lists to 1e-5."""
import numpy as np
import pytest
import torch

from xpoint_amd import synth

pytestmark = pytest.mark.gpu

CONFIG = {"prediction": {"matching": {"method": "bfmatcher", "knn_matches": False, "method_kwargs": {"crossCheck": True}}}}


def _case(seed):
    c = synth.make_eval_case(seed)
    t = {k: torch.from_numpy(v).cuda() for k, v in c.items()}
    data = {"optical": {"image": torch.zeros(t["prob_optical"].shape), "valid_mask": t["mask_optical"], "homography": t["H_optical"]},
            "thermal": {"image": torch.zeros(t["prob_thermal"].shape), "valid_mask": t["mask_thermal"], "homography": t["H_thermal"]}}
    return t, data


def test_points_min_dist_kernel(gpu_lib):
    from xpoint_amd import evaluation as ev
    rng = np.random.default_rng(5)
    a = rng.uniform(-5, 130, (777, 2)); b = rng.integers(0, 128, (1301, 2)).astype(np.float32)
    got = ev._min_dist(a, torch.from_numpy(b).cuda())
    diff = (a[:, None, :] - b[None].astype(np.float64)).astype(np.float32)
    ref = np.sqrt(diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]).min(1)
    assert np.array_equal(got, ref)
    assert ev._min_dist(a[:3], torch.zeros((0, 2)).cuda()).tolist() == [float("inf")] * 3
    assert ev._min_dist(a[:0], torch.from_numpy(b).cuda()).shape == (0,)


@pytest.mark.parametrize("seed", [0, 1])
def test_repeatability_vs_reference(gpu_lib, golden, seed):
    from xpoint_amd import evaluation as ev
    g = golden("g13_eval_metrics.npz")
    t, data = _case(seed)
    rep, nko, nkt = ev.compute_repeatability_for_sample({"prob": t["prob_optical"]}, {"prob": t["prob_thermal"]}, data,
                                                        t["H_optical"], t["H_thermal"], 0.015, [1, 3, 5])
    assert nko == g[f"s{seed}/rep/n_kp_optical"].tolist() and nkt == g[f"s{seed}/rep/n_kp_thermal"].tolist()
    for th in (1, 3, 5):
        assert np.allclose(np.array(rep[th]), g[f"s{seed}/rep/{th}"], rtol=0, atol=1e-12), (th, rep[th], g[f"s{seed}/rep/{th}"])
    one = ev.compute_repeatability_for_sample({"prob": t["prob_optical"]}, {"prob": t["prob_thermal"]}, data,
                                              t["H_optical"], t["H_thermal"], 0.015, 3)[0]
    assert list(one.keys()) == [3] and np.allclose(one[3], g[f"s{seed}/rep/3"], atol=1e-12)


@pytest.mark.parametrize("seed", [0, 1])
def test_descriptor_metrics_vs_reference(gpu_lib, golden, seed):
    from xpoint_amd import evaluation as ev
    g = golden("g13_eval_metrics.npz")
    t, data = _case(seed)
    po, pt = t["prob_optical"] * t["mask_optical"], t["prob_thermal"] * t["mask_thermal"]
    dd = ev.compute_descriptor_for_sample(po, pt, t["desc_optical"], t["desc_thermal"], data, CONFIG, 0.015, [2, 4])
    for th in (2, 4):
        for k in ("n_gt_optical", "n_gt_thermal"):
            assert dd[th][k] == int(g[f"s{seed}/desc/{th}/{k}"]), (th, k)
        for k in ("tp_optical", "tp_thermal", "matching_kp_numbers"):
            assert [float(v) for v in dd[th][k]] == g[f"s{seed}/desc/{th}/{k}"].tolist(), (th, k)
        for k in ("distance_optical", "distance_thermal", "m_score_optical", "m_score_thermal"):
            assert np.allclose(np.array(dd[th][k], np.float64), g[f"s{seed}/desc/{th}/{k}"], rtol=0, atol=1e-5), (th, k)
    res = ev.compute_desc_dict(dd)
    for th in (2, 4):
        for k in ("nn_map_optical", "nn_map_thermal", "nn_map", "m_score"):
            assert abs(float(res[th][k]) - float(g[f"s{seed}/res/{th}/{k}"])) < 1e-5, (th, k)
        for k in ("precision_optical", "recall_thermal"):
            assert np.allclose(res[th][k], g[f"s{seed}/res/{th}/{k}"], atol=1e-6)


def test_compute_metrics_runs_end_to_end(gpu_lib):
    """The full loop (network forward -> NMS -> repeatability + descriptor metrics) on identity homographies: identical
    inputs for both spectra give repeatability 1 and an M-score of 1."""
    from xpoint_amd import evaluation as ev, models
    H, W = 64, 96
    cfg = synth.xpoint_exp1_config(H, W)
    net = models.XPoint(cfg)
    net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True)
    net = net.to("cuda").eval()
    d = synth.to_torch(synth.make_pair_batch(0, 2, H, W))
    d["thermal"]["image"] = d["optical"]["image"].clone()
    config = dict(CONFIG); config["prediction"] = dict(CONFIG["prediction"], nms=4, topk=0, cpu_nms=False)
    with torch.no_grad():
        out = ev.compute_metrics(net, [d], "cuda", config, 0.015, [1, 3], [2])
    assert out["repeatability"]["repeatability_mean"][1] == 1.0 and out["repeatability"]["n_kp_avg"] > 10
    assert abs(out["descriptor"][2]["m_score"] - 1.0) < 1e-9 and out["descriptor"][2]["nn_map"] > 0.99
