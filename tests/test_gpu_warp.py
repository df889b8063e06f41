"""§8(f) rank 2, last step — the aligned image: xp_warp_perspective / utils.warp_perspective / predict_align_image_pair(..., estimate_homography=True)
against the oracle's plain-C restatement of OpenCV's documented INTER_LINEAR fixed-point scheme (oracle/csrc/oracle_kernels.c:
xo_warp_perspective_u8 / _f32; reference call predict_align_image_pair.py:308).  PARITY UNPINNED vs OpenCV itself (absent); the bar here is
bit-equality HIP == oracle for uint8 and for float32 (tolerance written: 0 for u8, 1e-6 for f32 — measured 0)."""
import numpy as np
import pytest
import torch

from xpoint_amd import synth

pytestmark = pytest.mark.gpu


def _img(tag, shape, dtype):
    u = synth.uniform(tag, shape, 0.0, 1.0)
    return (u * 255.999).astype(np.uint8) if dtype == np.uint8 else u.astype(np.float32)


def _homography(kind, H, W):
    if kind == "identity":
        return np.eye(3)
    if kind == "shift_int":
        return np.array([[1, 0, 7], [0, 1, -5], [0, 0, 1.0]])
    if kind == "shift_frac":
        return np.array([[1, 0, 3.3], [0, 1, 2.71], [0, 0, 1.0]])
    if kind == "affine":
        a = 0.21
        return np.array([[1.1 * np.cos(a), -np.sin(a), 0.12 * W], [np.sin(a), 0.93 * np.cos(a), -0.08 * H], [0, 0, 1.0]])
    if kind == "projective":
        return np.array([[0.94, 0.07, 11.3], [-0.05, 1.06, 6.9], [1.7e-4, -2.3e-4, 1.0]])
    if kind == "strong":          # the horizon (w = 0) crosses the destination: W == 0 / huge coordinates / saturation paths
        return np.array([[1.3, 0.2, -20.0], [0.1, 0.8, 14.0], [4.0e-3, 2.5e-3, 1.0]])
    if kind == "far":             # the whole source lands outside the destination
        return np.array([[1, 0, 10.0 * W], [0, 1, 0], [0, 0, 1.0]])
    if kind == "singular":        # det = 0: OpenCV's invert gives the zero matrix -> every pixel samples source (0, 0)
        return np.array([[1, 2, 3], [2, 4, 6], [0, 0, 1.0]])
    raise KeyError(kind)


KINDS = ["identity", "shift_int", "shift_frac", "affine", "projective", "strong", "far", "singular"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_warp_perspective_equals_oracle(gpu_lib, kind, dtype):
    from oracle import xpoint_oracle as xo
    from xpoint_amd import utils
    H, W = 120, 200                      # 200 = three 64-wide blocks + a ragged one: the block base of OpenCV's coordinate arithmetic matters
    img = _img(f"warp{kind}", (H, W), dtype)
    M = _homography(kind, H, W)
    ref = xo.warp_perspective(img, M)
    got = utils.warp_perspective(torch.from_numpy(img).cuda(), M).cpu().numpy()
    assert got.dtype == ref.dtype and got.shape == ref.shape
    if dtype == np.uint8:
        assert np.array_equal(got, ref), int((got != ref).sum())
    else:
        assert float(np.abs(got - ref).max()) <= 1e-6
        assert np.array_equal(got, ref)          # same operation order, no contraction: the same bits
    if kind == "identity":
        assert np.array_equal(got, img)
    if kind == "shift_int":
        assert np.array_equal(got[:-5, 7:], img[5:, :-7]) and not got[:, :7].any() and not got[-5:].any()
    if kind == "far":
        assert not got.any()
    if kind == "singular":
        assert np.all(got == img[0, 0])


def test_warp_perspective_shapes_channels_batch_dsize_inverse(gpu_lib):
    from oracle import xpoint_oracle as xo
    from xpoint_amd import utils
    H, W = 57, 91
    rgb = _img("warprgb", (H, W, 3), np.uint8)
    Ms = np.stack([_homography(k, H, W) for k in ("projective", "affine", "shift_frac")])
    # (H, W, C) image, dsize different from the source size (width, height as in cv2), narrow destination (block width = the width)
    for dsize in ((140, 33), (40, 70), (64, 8)):
        ref = xo.warp_perspective(rgb, Ms[0], dsize)
        got = utils.warp_perspective(torch.from_numpy(rgb).cuda(), Ms[0], dsize).cpu().numpy()
        assert got.shape == (dsize[1], dsize[0], 3) and np.array_equal(got, ref), dsize
    # batch with one matrix per image, matrices as a float64 DEVICE tensor (what find_homography_batched returns)
    batch = np.stack([_img(f"warpb{i}", (H, W, 1), np.float32) for i in range(3)])
    got = utils.warp_perspective(torch.from_numpy(batch).cuda(), torch.from_numpy(Ms).cuda()).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], xo.warp_perspective(batch[i], Ms[i]))
    # one matrix broadcast over the batch; the network's (B, 1, H, W) layout comes back as (B, 1, H, W)
    nchw = torch.from_numpy(batch[..., 0][:, None]).cuda()
    got = utils.warp_perspective(nchw, Ms[1])
    assert tuple(got.shape) == (3, 1, H, W)
    assert np.array_equal(got[2, 0].cpu().numpy(), xo.warp_perspective(batch[2, ..., 0], Ms[1]))
    # cv2.WARP_INVERSE_MAP: M is used as given; equals warping by inv(M) up to the inversion's rounding (here: against the oracle's own flag)
    ref = xo.warp_perspective(rgb, Ms[0], inverse_map=True)
    got = utils.warp_perspective(torch.from_numpy(rgb).cuda(), Ms[0], inverse_map=True).cpu().numpy()
    assert np.array_equal(got, ref)
    # gray -> 3 identical channels (cv2.COLOR_GRAY2RGB ahead of the warp), and the reference's quantisation of a [0, 1] float image on load
    gray = synth.uniform("warpq", (H, W), -0.2, 1.2).astype(np.float32)
    ref = xo.warp_perspective(np.repeat(xo.to_u8_image(gray)[..., None], 3, axis=2), Ms[0])
    got = utils.warp_perspective(torch.from_numpy(gray).cuda(), Ms[0], quantise_u8=True, dst_channels=3).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, ref)


def test_checkerboard_visualization_matches_the_reference_recipe(gpu_lib):
    """demo.py:222-234: warp the visible image by H into the other image's frame, composite in 50-pixel checker cells — against the same recipe evaluated
    with the oracle's warp in numpy; other image of a different size than the visible one."""
    from oracle import xpoint_oracle as xo
    from xpoint_amd import utils
    vis = _img("cbv", (120, 170), np.uint8); oth = _img("cbo", (130, 160), np.uint8)
    M = _homography("projective", 120, 170)
    got = utils.checkerboard_visualization(torch.from_numpy(vis).cuda(), torch.from_numpy(oth).cuda(), M).cpu().numpy()
    warped = xo.warp_perspective(vis, M, (160, 130))
    x, y = np.meshgrid(np.arange(160), np.arange(130))
    ref = np.where(((x // 50) + (y // 50)) % 2, warped, oth)
    assert got.shape == oth.shape and np.array_equal(got, ref)
    with pytest.raises(ValueError):
        utils.checkerboard_visualization(torch.from_numpy(vis).cuda().float(), torch.from_numpy(oth).cuda(), M)


def test_warp_perspective_degenerate_sizes(gpu_lib):
    """1 x 1 images, one-row / one-column destinations (OpenCV's block width then follows the height), a destination far larger than the source."""
    from oracle import xpoint_oracle as xo
    from xpoint_amd import utils
    M = np.array([[1.02, 0.01, 0.4], [-0.02, 0.98, 0.3], [1e-4, 2e-4, 1.0]])
    for (hs, ws), dsize in (((1, 1), (1, 1)), ((1, 1), (9, 7)), ((5, 300), (300, 1)), ((300, 5), (1, 300)), ((17, 23), (700, 3)), ((9, 9), (130, 15)), ((40, 30), (900, 600))):
        for dtype in (np.uint8, np.float32):
            img = _img(f"deg{hs}{ws}{dsize}", (hs, ws), dtype)
            ref = xo.warp_perspective(img, M, dsize)
            got = utils.warp_perspective(torch.from_numpy(img).cuda(), M, dsize).cpu().numpy()
            assert got.shape == (dsize[1], dsize[0]) and np.array_equal(got, ref), ((hs, ws), dsize, dtype)


def test_warp_perspective_argument_errors(gpu_lib):
    import ctypes
    from xpoint_amd import _lib as L, utils
    img = torch.zeros((8, 8), dtype=torch.uint8)
    with pytest.raises(L.XPointHipError):                      # no CPU fallback
        utils.warp_perspective(img, np.eye(3))
    with pytest.raises(ValueError):
        utils.warp_perspective(img.cuda().to(torch.int32), np.eye(3))
    with pytest.raises(ValueError):
        utils.warp_perspective(img.cuda(), np.eye(3), quantise_u8=True)
    with pytest.raises(ValueError):
        utils.warp_perspective(torch.zeros((2, 8, 8, 1), dtype=torch.uint8).cuda(), np.zeros((3, 3, 3)))
    d = img.cuda(); o = torch.empty_like(d); M = torch.eye(3, dtype=torch.float64).cuda()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = L.current_stream()
    for args in ((None, vp(o), vp(M), 1, 8, 8, 8, 8, 1, 1, 0, 0), (vp(d), vp(o), vp(M), 1, 8, 8, 8, 8, 5, 5, 0, 0),
                 (vp(d), vp(o), vp(M), 1, 8, 8, 8, 8, 3, 1, 0, 0), (vp(d), vp(o), vp(M), 1, 8, 8, 8, 8, 1, 1, 7, 0),
                 (vp(d), vp(d), vp(M), 1, 8, 8, 8, 8, 1, 1, 0, 0), (vp(d), vp(o), vp(M), 1, 40000, 8, 8, 8, 1, 1, 0, 0)):
        with pytest.raises(L.XPointHipError):
            L.call("xp_warp_perspective", *args, st)


def test_predict_align_image_pair_returns_the_aligned_image(gpu_lib):
    """predict_align_image_pair.py:271, 283-308 end to end: the flow returns `warped_optical` = the uint8 RGB optical image warped by H_est.
    thermal == optical -> H_est = identity -> the warp is the quantised image itself; fewer than 4 matches -> identity too (the reference's
    `H_est = np.eye(3,3)` branch); the batched PairPipeline's device-resident warp equals the per-pair flow."""
    from oracle import xpoint_oracle as xo
    from xpoint_amd import models
    from xpoint_amd.predict import PairPipeline, predict_align_image_pair
    H, W, B = 96, 128, 2
    cfg = synth.xpoint_exp1_config(H, W)
    net = models.XPoint(cfg)
    net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True)
    net = net.to("cuda").eval()
    d = synth.to_torch(synth.make_pair_batch(4, B, H, W), "cuda")
    with torch.no_grad():
        _, _, res = predict_align_image_pair(net, d, estimate_homography=True)
    for i, r in enumerate(res):
        img = d["optical"]["image"][i, 0].cpu().numpy()
        ref = xo.warp_perspective(np.repeat(xo.to_u8_image(img)[..., None], 3, axis=2), r["H_est"])
        got = r["warped_optical"].cpu().numpy()
        assert got.shape == (H, W, 3) and got.dtype == np.uint8 and np.array_equal(got, ref)
    d["thermal"]["image"] = d["optical"]["image"].clone()
    with torch.no_grad():
        _, _, res = predict_align_image_pair(net, d, estimate_homography=True)
        pipe = PairPipeline(net, B, H, W, cap=2048, estimate_homography=True, warp_optical=True)
        out = pipe.run(d["optical"]["image"], d["thermal"]["image"]).fetch()
    for i, r in enumerate(res):
        q = xo.to_u8_image(d["optical"]["image"][i, 0].cpu().numpy())
        assert np.abs(r["H_est"] - np.eye(3)).max() < 1e-6
        assert np.array_equal(r["warped_optical"].cpu().numpy(), np.repeat(q[..., None], 3, axis=2))
        assert np.array_equal(out[i]["warped_optical"].numpy(), xo.warp_perspective(q, out[i]["H_est"]))
    # fewer than four matches: detection threshold no pixel reaches -> no keypoints, no matches, identity, the quantised image back
    with torch.no_grad():
        _, _, res = predict_align_image_pair(net, d, dict(detection_threshold=2.0), estimate_homography=True)
    assert len(res[0]["matches"]) == 0 and np.array_equal(res[0]["H_est"], np.eye(3)) and res[0]["matchesMask"] == []
    assert np.array_equal(res[0]["warped_optical"].cpu().numpy()[..., 0], xo.to_u8_image(d["optical"]["image"][0, 0].cpu().numpy()))
    with pytest.raises(ValueError):
        predict_align_image_pair(net, d, warp_optical=True)
    with torch.no_grad():
        _, _, res = predict_align_image_pair(net, d, estimate_homography=True, warp_optical=False)
    assert "warped_optical" not in res[0] and "H_est" in res[0]


def test_h_correctness_on_synthetic_ground_truth_homographies(gpu_lib):
    """benchmark_evaluation.py:560-586, 755-830 on pairs with KNOWN homographies (synth.make_eval_case: heat maps with peaks at the images of common scene
    points under H_optical / H_thermal, descriptor maps sampled from one field in the scene frame): the estimated optical -> thermal model against
    gt = H_thermal @ inv(H_optical) through the reference's own quality metric (mean distance of its four "corner" points; h_correctness at epsilon).
    And the warp closes the loop in the SAME (x, y) convention: the optical heat map warped by gt puts mass where the thermal keypoints are — which the
    identity (no registration) does not."""
    from xpoint_amd import evaluation as ev, utils
    config = {"prediction": {"matching": {"method": "bfmatcher", "knn_matches": False, "method_kwargs": {"crossCheck": True}}}}
    dists = []
    for seed in (0, 1, 2):
        c = synth.make_eval_case(seed)
        t = {k: torch.from_numpy(v).cuda() for k, v in c.items()}
        data = {"optical": {"image": torch.zeros(t["prob_optical"].shape), "valid_mask": t["mask_optical"], "homography": t["H_optical"]},
                "thermal": {"image": torch.zeros(t["prob_thermal"].shape), "valid_mask": t["mask_thermal"], "homography": t["H_thermal"]}}
        pd = ev.compute_pts_dist_for_sample(t["prob_optical"] * t["mask_optical"], t["prob_thermal"] * t["mask_thermal"], t["desc_optical"], t["desc_thermal"],
                                            data, config, 0.015, [3])
        assert list(pd) == [3] and len(pd[3]) == t["prob_optical"].shape[0]
        dists += pd[3]
        # the aligned heat map: warp by the ground truth vs by the identity
        for b in range(t["prob_optical"].shape[0]):
            gt = c["H_thermal"][b].astype(np.float64) @ np.linalg.inv(c["H_optical"][b].astype(np.float64))
            po = t["prob_optical"][b, 0].contiguous()
            kp_t = torch.nonzero(t["prob_thermal"][b, 0] > 0.015).cpu().numpy()
            kp_t = kp_t[(kp_t[:, 0] >= 1) & (kp_t[:, 0] < po.shape[0] - 1) & (kp_t[:, 1] >= 1) & (kp_t[:, 1] < po.shape[1] - 1)]
            rates = []
            for M in (gt, np.eye(3)):
                w = utils.warp_perspective(po, M).cpu().numpy()
                rates.append(np.mean([w[y - 1:y + 2, x - 1:x + 2].max() > 0 for y, x in kp_t]))
            assert rates[0] > 0.6 and rates[0] > rates[1] + 0.25, rates       # 260 of 350 peaks are common scene points
    hd = ev.compute_homography_dict({3: dists}, [1, 3, 5])[3]
    assert set(hd) == {"average_h_error", "h_correctness"} and set(hd["h_correctness"]) == {"epsilon_warp_th1", "epsilon_warp_th3", "epsilon_warp_th5"}
    hc = hd["h_correctness"]
    assert 0.0 <= hc["epsilon_warp_th1"] <= hc["epsilon_warp_th3"] <= hc["epsilon_warp_th5"] <= 1.0
    # keypoints are integer pixels (+- 0.5 px of rounding on both sides), so the model is recovered to about a pixel at the image "corners"
    assert max(dists) < 5.0 and hd["average_h_error"] < 3.0 and hc["epsilon_warp_th5"] == 1.0, (dists, hd)
