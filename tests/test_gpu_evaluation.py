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
    hd = out["homography"][3]          # identical images: the estimated homography is the identity
    assert hd["average_h_error"] < 1e-6 and hd["h_correctness"]["epsilon_warp_th2"] == 1.0


def _project(H, pts):
    hom = np.concatenate([pts, np.ones((pts.shape[0], 1))], 1) @ H.T
    return hom[:, :2] / hom[:, 2:3]


def _corr_case(seed, n, inlier_frac, noise, W=640, H=480):
    rng = np.random.default_rng(seed)
    r = rng.uniform(-1, 1, 8)
    Ht = np.array([[1 + 0.08 * r[0], 0.06 * r[1], 25 * r[2]], [0.06 * r[3], 1 + 0.08 * r[4], 25 * r[5]], [1e-4 * r[6], 1e-4 * r[7], 1.0]])
    src = np.stack([rng.uniform(0, W, n), rng.uniform(0, H, n)], 1)
    dst = _project(Ht, src) + rng.normal(0, noise, (n, 2))
    out = rng.random(n) > inlier_frac
    dst[out] = np.stack([rng.uniform(0, W, out.sum()), rng.uniform(0, H, out.sum())], 1)
    return Ht, src.astype(np.float32), dst.astype(np.float32), ~out


def test_find_homography_recovers_model_under_outliers(gpu_lib):
    """the device estimator (stand-in for cv2.findHomography MAGSAC; unpinnable against OpenCV here)
    recovers a known homography from noisy correspondences with 45 % outliers, marks the inliers, is deterministic, handles
    batches with different counts and reports 'no model' below four correspondences — the contract the reference's callers
    rely on (predict_align_image_pair.py:291-303, benchmark_evaluation.py:796-824)."""
    from xpoint_amd import utils
    corners = np.array([[0, 0], [640, 0], [0, 480], [640, 480]], np.float64)
    Ht, src, dst, inl = _corr_case(1, 900, 0.55, 0.5)
    H1, m1 = utils.find_homography(src.reshape(-1, 1, 2), dst.reshape(-1, 1, 2), 3.0)
    H2, m2 = utils.find_homography(src, dst, 3.0)
    assert np.array_equal(H1, H2) and np.array_equal(m1, m2)                      # deterministic
    assert np.abs(_project(H1, corners) - _project(Ht, corners)).max() < 1.0       # corner error below a pixel
    err = np.linalg.norm(_project(H1, src.astype(np.float64)) - dst, axis=1)
    assert np.array_equal(m1.ravel().astype(bool), err <= 3.0) or (np.abs(err - 3.0) < 1e-3).any()
    assert (m1.ravel().astype(bool) & inl).sum() > 0.97 * inl.sum() and m1.shape == (900, 1) and abs(H1[2, 2] - 1.0) < 1e-12
    # exact data: the model is recovered to rounding
    Ht, src, dst, _ = _corr_case(2, 200, 1.0, 0.0)
    H3, m3 = utils.find_homography(src, dst, 1.0)
    assert np.abs(_project(H3, corners) - _project(Ht, corners)).max() < 2e-2 and int(m3.sum()) == 200
    # below four correspondences / nothing: no model, empty mask (the reference then uses identity / 999.0)
    Hn, mn = utils.find_homography(src[:3], dst[:3], 3.0)
    assert Hn is None and mn.shape == (3, 1) and int(mn.sum()) == 0
    # batch with different counts, incl. an empty pair
    cases = [_corr_case(s, 600, 0.6, 0.4) for s in (3, 4)]
    cap = 700
    S = torch.zeros((3, cap, 2)); D = torch.zeros((3, cap, 2))
    S[0, :600] = torch.from_numpy(cases[0][1]); D[0, :600] = torch.from_numpy(cases[0][2])
    S[2, :450] = torch.from_numpy(cases[1][1][:450]); D[2, :450] = torch.from_numpy(cases[1][2][:450])
    Hb, mb, nb = utils.find_homography_batched(S.cuda(), D.cuda(), torch.tensor([600, 0, 450], dtype=torch.int32).cuda(), 3.0)
    assert int(nb[1]) == 0 and torch.equal(Hb[1].cpu(), torch.eye(3, dtype=torch.float64)) and int(mb[1].sum()) == 0
    for b, c in ((0, cases[0]), (2, cases[1])):
        assert np.abs(_project(Hb[b].cpu().numpy(), corners) - _project(c[0], corners)).max() < 1.0
        assert int(nb[b]) == int(mb[b].sum()) and int(mb[b, (600 if b == 0 else 450):].sum()) == 0


@pytest.mark.parametrize("seed_case,n,inl,noise,thr,iters,hseed", [(1, 900, 0.55, 0.5, 3.0, 10000, 0), (5, 300, 0.8, 0.3, 2.0, 2000, 7), (6, 40, 0.7, 0.2, 3.0, 500, 123),
                                                                  (2, 200, 1.0, 0.0, 1.0, 1000, 0)])
def test_find_homography_equals_oracle(gpu_lib, seed_case, n, inl, noise, thr, iters, hseed):
    """The shipped estimator against its independent numpy restatement (oracle/homography_oracle.py: same counter-hashed samples,
    same normalised DLT, same f32 MSAC score in point order, same three least-squares rounds): SAME winning hypothesis,
    identical inlier mask and count, H to 1e-9.  (cv2.findHomography itself stays unpinned: OpenCV is absent)"""
    from oracle import homography_oracle as ho
    from xpoint_amd import utils
    _, src, dst, _ = _corr_case(seed_case, n, inl, noise)
    H, mask, n_inl = utils.find_homography_batched(torch.from_numpy(src).cuda()[None], torch.from_numpy(dst).cuda()[None], None, thr, iters, hseed)
    Ho, mo, no, best = ho.find_homography(src, dst, thr, iters, hseed, pair=0)
    assert best >= 0 and int(n_inl[0]) == no
    assert np.array_equal(mask[0].cpu().numpy(), mo)
    np.testing.assert_allclose(H[0].cpu().numpy(), Ho, rtol=1e-9, atol=1e-9)
    # second pair of a batch: the pair index enters the hypothesis hash
    S = torch.from_numpy(np.stack([src, src])).cuda(); D = torch.from_numpy(np.stack([dst, dst])).cuda()
    Hb, mb, nb = utils.find_homography_batched(S, D, None, thr, iters, hseed)
    H1, m1, n1, _ = ho.find_homography(src, dst, thr, iters, hseed, pair=1)
    assert int(nb[1]) == n1 and np.array_equal(mb[1].cpu().numpy(), m1)
    np.testing.assert_allclose(Hb[1].cpu().numpy(), H1, rtol=1e-9, atol=1e-9)


def test_predict_align_image_pair_with_registration(gpu_lib):
    """predict_align_image_pair.py:287-303 end to end: thermal = optical shifted by (dx, dy) pixels -> the estimated
    homography is that translation (keypoints are integer pixels, so to a fraction of a pixel)."""
    from xpoint_amd import models
    from xpoint_amd.predict import predict_align_image_pair
    H, W = 96, 128
    cfg = synth.xpoint_exp1_config(H, W)
    net = models.XPoint(cfg)
    net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True)
    net = net.to("cuda").eval()
    d = synth.to_torch(synth.make_pair_batch(4, 1, H, W), "cuda")
    with torch.no_grad():
        _, _, res = predict_align_image_pair(net, d, estimate_homography=True)
    r = res[0]
    assert r["H_est"].shape == (3, 3) and len(r["matchesMask"]) in (0, len(r["matches"]))
    d["thermal"]["image"] = d["optical"]["image"].clone()
    with torch.no_grad():
        _, _, res = predict_align_image_pair(net, d, estimate_homography=True)
    r = res[0]
    assert len(r["matches"]) >= 4 and np.abs(r["H_est"] - np.eye(3)).max() < 1e-6 and sum(r["matchesMask"]) == len(r["matches"])


def test_pairpipeline_device_registration(gpu_lib):
    """PairPipeline(estimate_homography=True): the registration step runs on the device from the match lists (no host
    round trip) and agrees with the per-pair host-facing call on the same matches."""
    from xpoint_amd import models, utils
    from xpoint_amd.predict import PairPipeline
    H, W, B = 96, 128, 2
    cfg = synth.xpoint_exp1_config(H, W)
    net = models.XPoint(cfg)
    net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True)
    net = net.to("cuda").eval()
    d = synth.to_torch(synth.make_pair_batch(0, B, H, W), "cuda")
    d["thermal"]["image"] = torch.roll(d["optical"]["image"], shifts=(0, 0), dims=(2, 3)).clone()       # identical spectra
    with torch.no_grad():
        out = PairPipeline(net, B, H, W, cap=2048, estimate_homography=True).run(d["optical"]["image"], d["thermal"]["image"]).fetch()
    for r in out:
        assert r["n_inliers"] == len(r["match_q"]) == int(r["matchesMask"].sum()) and r["n_inliers"] >= 4
        assert np.abs(r["H_est"] - np.eye(3)).max() < 1e-6
        src = r["kp_optical"][r["match_q"]].flip(-1).float(); dst = r["kp_thermal"][r["match_t"]].flip(-1).float()
        H2, m2 = utils.find_homography(src, dst, 3.0)
        assert np.array_equal(H2, r["H_est"]) and np.array_equal(m2.ravel(), r["matchesMask"])


def test_evaluation_and_homography_edge_cases(gpu_lib):
    """Empty keypoint sets, collinear correspondences and duplicate points: no crash, no NaN, the documented 'no model' result."""
    from xpoint_amd import evaluation as ev, utils
    B, H, W = 1, 64, 96
    zero = torch.zeros((B, 1, H, W)).cuda()
    one = zero.clone(); one[0, 0, 10, 20] = 0.9; one[0, 0, 30, 40] = 0.5
    eye = torch.eye(3).repeat(B, 1, 1)
    data = {"optical": {"image": torch.zeros((B, 1, H, W)), "valid_mask": torch.ones((B, 1, H, W)).cuda(), "homography": eye},
            "thermal": {"image": torch.zeros((B, 1, H, W)), "valid_mask": torch.ones((B, 1, H, W)).cuda(), "homography": eye}}
    rep, nko, nkt = ev.compute_repeatability_for_sample({"prob": zero}, {"prob": zero}, data, eye, eye, 0.015, [3])
    assert rep[3] == [] and nko == [0] and nkt == [0]                      # reference: nothing appended when both sets are empty
    rep, nko, nkt = ev.compute_repeatability_for_sample({"prob": one}, {"prob": zero}, data, eye, eye, 0.015, [3])
    assert rep[3] == [0.0] and nko == [2] and nkt == [0]
    desc = torch.nn.functional.normalize(torch.randn((B, 256, H // 8, W // 8)), dim=1).cuda()
    dd = ev.compute_descriptor_for_sample(one, zero, desc, desc, data, {"prediction": {"matching": {"method": "bfmatcher", "knn_matches": False,
                                          "method_kwargs": {"crossCheck": True}}}}, 0.015, [2])
    assert dd[2]["tp_optical"] == [] and dd[2]["m_score_optical"] == [0.0] and dd[2]["n_gt_optical"] == 0
    # collinear correspondences: every 4-point sample is degenerate -> no model
    x = np.linspace(0, 100, 50, dtype=np.float32)
    src = np.stack([x, 2 * x + 1], 1); dst = np.stack([x + 3, 2 * x + 5], 1)
    Hc, mc = utils.find_homography(src, dst, 3.0, max_iters=512)
    assert Hc is None and int(mc.sum()) == 0
    # all correspondences identical points
    src = np.full((20, 2), 5.0, np.float32)
    Hd, md = utils.find_homography(src, src, 3.0, max_iters=256)
    assert Hd is None or np.isfinite(Hd).all()


def test_desc_process_and_display_sample_time_dict(gpu_lib, tmp_path):
    """The reference's timing harness (benchmark_evaluation.py:16-227) as a callable: same signature and `time_dict_seconds` keys
    (two_forward, nms, interpolate), one forward / NMS entry per sample and one interpolate entry per pair, positive HIP-event times;
    with args.plot the match / registration quantities the reference draws are written next to where its PNG would go."""
    import types
    from xpoint_amd import evaluation as ev, models
    H, W = 64, 96
    cfg = synth.xpoint_exp1_config(H, W)
    net = models.XPoint(cfg)
    net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True)
    net = net.to("cuda").eval()

    class DS:               # dataset[index] -> un-batched CPU sample, like ImagePairDataset.__getitem__
        def __getitem__(self, i):
            d = synth.to_torch(synth.make_pair_batch(i, 1, H, W))
            return {s: {k: v[0] for k, v in d[s].items()} for s in d}
    config = dict(CONFIG); config["prediction"] = dict(CONFIG["prediction"], detection_threshold=0.015, nms=8, topk=0, cpu_nms=True, reprojection_threshold=3)
    args = types.SimpleNamespace(index=[0, 3, 5], plot=False, radius=4, output_dir=str(tmp_path), model_dir="model_weights/xpoint", version="synth", seed=0)
    with torch.no_grad():
        td = ev.desc_process_and_display_sample(net, DS(), "cuda", config, args)
    assert sorted(td) == ["interpolate", "nms", "two_forward"]
    assert len(td["two_forward"]) == 3 and len(td["nms"]) == 3 and len(td["interpolate"]) == 3
    assert all(0.0 < t < 5.0 for k in td for t in td[k])
    args.plot = True; args.index = [3]
    with torch.no_grad():
        ev.desc_process_and_display_sample(net, DS(), "cuda", config, args)
    import glob
    files = glob.glob(str(tmp_path / "images" / "supp" / "i3" / "*.npz"))
    assert len(files) == 1
    z = np.load(files[0])
    assert z["matches"].shape[1] == 2 and z["H_est"].shape == (3, 3) and len(z["matches_mask"]) == len(z["matches"])
