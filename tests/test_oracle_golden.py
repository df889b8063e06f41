"""CPU (-m "not gpu"): pins the oracle restatement against the golden vectors that were produced by
the REAL reference (oracle/refharness/make_golden.py).  If these fail the oracle cannot be trusted
as the checker for the HIP path."""
import numpy as np
import pytest
import torch

from oracle import xpoint_oracle as xo
from oracle.refharness.make_golden import SCAN_CASES, scan_inputs
from xpoint_amd import synth

torch.set_num_threads(1)


def _sd(cfg):
    return {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}


@pytest.mark.parametrize("case", SCAN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_g1_selective_scan(golden, case):
    g = golden("g1_selective_scan.npz")
    name = "scan/%d_%d_%d_%d_%d" % case
    u, delta, A, Bm, Cm, Dv, bias = [torch.from_numpy(x) for x in scan_inputs(name, *case)]
    out = xo.selective_scan(u, delta, A, Bm, Cm, Dv, bias, True)
    ref = g[name + "/out"]
    # reference tolerance (test_selective_scan.py:401): rtol 6e-4 / atol 2e-3; the oracle is far tighter
    np.testing.assert_allclose(out.numpy(), ref, rtol=0, atol=2e-6 * max(1.0, np.abs(ref).max()))
    if name + "/out_plain" in g.files:
        out2 = xo.selective_scan(u, delta, A, Bm, Cm, None, None, False)
        np.testing.assert_allclose(out2.numpy(), g[name + "/out_plain"], rtol=0, atol=2e-6 * max(1.0, np.abs(ref).max()))


def test_g1_last_state_consistency():
    # last state = h_L; check it against a direct fp64 recurrence on a small case
    case = (1, 2, 3, 4, 37)
    u, delta, A, Bm, Cm, Dv, bias = [torch.from_numpy(x) for x in scan_inputs("scan/ls", *case)]
    out, last = xo.selective_scan(u, delta, A, Bm, Cm, Dv, bias, True, return_last_state=True)
    B, K, C, N, L = case
    d = torch.nn.functional.softplus(delta.double() + bias.double()[None, :, None])
    h = torch.zeros(B, K * C, N, dtype=torch.float64)
    for l in range(L):
        Bl = Bm[:, :, :, l].double().repeat_interleave(C, dim=1)
        h = torch.exp(d[:, :, l, None] * A.double()[None]) * h + d[:, :, l, None] * Bl * u.double()[:, :, l, None]
    np.testing.assert_allclose(last.numpy(), h.numpy(), atol=1e-5)


@pytest.mark.parametrize("shp", [(2, 3, 5, 7), (1, 2, 33, 58)])
def test_g2_cross_scan_merge_exact(golden, shp):
    g = golden("g2_cross_scan.npz")
    x = torch.arange(int(np.prod(shp)), dtype=torch.float32).view(shp)
    xs = xo.cross_scan(x)
    assert np.array_equal(xs.numpy(), g["scan/%dx%dx%dx%d" % shp])
    ys = (xs * torch.tensor([1.0, 2.0, 3.0, 5.0]).view(1, 4, 1, 1)).view(shp[0], 4, shp[1], shp[2], shp[3])
    assert np.array_equal(xo.cross_merge(ys).numpy(), g["merge/%dx%dx%dx%d" % shp])


def test_g3_block_and_ss2d(golden):
    g = golden("g345_model.npz")
    cfg = synth.xpoint_exp1_config(64, 96)
    sd = _sd(cfg)
    pre = "encoder.layers.0.blocks.0."
    x = torch.from_numpy(g["full_64x96/blk_in"])
    np.testing.assert_allclose(xo.vss_block(x, sd, pre).numpy(), g["full_64x96/blk_out"], atol=2e-6)
    xin = torch.from_numpy(g["full_64x96/ss2d_in"])
    np.testing.assert_allclose(xo.ss2d(xin, sd, pre + "op.").numpy(), g["full_64x96/ss2d_out"], atol=2e-6)


@pytest.mark.parametrize("tag,H,W,B,vssm", [("tiny32_64x96", 64, 96, 1, {"EMBED_DIM": 32}),
                                            ("full_64x96", 64, 96, 2, None)])
def test_g4_g5_forward(golden, tag, H, W, B, vssm):
    g = golden("g345_model.npz")
    cfg = synth.xpoint_exp1_config(H, W, vssm=vssm)
    sd = _sd(cfg)
    data = synth.to_torch(synth.make_pair_batch(0, B, H, W))
    with torch.no_grad():
        o, t, _ = xo.xpoint_forward(data, sd)
    for spec, r in (("optical", o), ("thermal", t)):
        for k, tol in (("prob", 1e-6), ("desc", 1e-6), ("encoder_output", 2e-5)):
            np.testing.assert_allclose(r[k].numpy(), g[f"{tag}/{spec}/{k}"], rtol=0, atol=tol, err_msg=f"{spec}/{k}")


def test_g5_g10_forward_224x320_end_to_end(golden):
    g = golden("g345_model.npz")
    tag, H, W = "full_224x320", 224, 320
    sd = _sd(synth.xpoint_exp1_config(H, W))
    data = synth.to_torch(synth.make_pair_batch(0, 1, H, W))
    with torch.no_grad():
        res, (o, t, _), (po, pt) = xo.predict_align_image_pair(data, sd)
    np.testing.assert_allclose(o["prob"].numpy(), g[f"{tag}/optical/prob"], atol=1e-6)
    np.testing.assert_allclose(t["prob"].numpy(), g[f"{tag}/thermal/prob"], atol=1e-6)
    np.testing.assert_allclose(o["desc"].numpy(), g[f"{tag}/optical/desc"], atol=1e-6)
    r = res[0]
    assert np.array_equal(r["kp_optical"].numpy(), g[f"{tag}/kp_optical"])
    assert np.array_equal(r["kp_thermal"].numpy(), g[f"{tag}/kp_thermal"])
    np.testing.assert_allclose(r["desc_optical"].numpy(), g[f"{tag}/desc_optical_sampled"], atol=1e-6)
    # strict mutual-NN (fp64 direct form) vs the reference's NNMatcher (fp32 Gram form): identical pairs
    # except where NNMatcher's own fp32 rounding decides a near-tie; report and bound those.
    mine = {(m.queryIdx, m.trainIdx) for m in r["matches"]}
    ref = {tuple(x) for x in g[f"{tag}/matches_nnmatcher"].tolist()}
    assert len(mine ^ ref) <= 2, (len(mine), len(ref), sorted(mine ^ ref)[:10])
    # keep_top_k on the unmasked prob (reference predict_keypoints flow)
    pk = xo.box_nms(o["prob"], 8, 0.015, keep_top_k=100)
    assert np.array_equal(torch.nonzero(pk[0].squeeze() > 0.015).numpy(), g[f"{tag}/kp_optical_top100"])


@pytest.mark.parametrize("name,shape,size,levels", [("ties", (1, 1, 40, 56), 8, 16), ("size4", (1, 1, 33, 47), 4, 0),
                                                    ("batch", (3, 1, 32, 48), 8, 64), ("size3", (1, 1, 24, 24), 3, 0)])
def test_g6_box_nms(golden, name, shape, size, levels):
    g = golden("g6_box_nms.npz")
    p = synth.uniform("nms/" + name, shape, 0.0, 1.0)
    if levels:
        p = (np.floor(p * levels) / levels).astype(np.float32)
    assert np.array_equal(xo.box_nms(torch.from_numpy(p), size, 0.3).numpy(), g[name + "/out"])
    assert np.array_equal(xo.box_nms(torch.from_numpy(p), size, 0.3, keep_top_k=5).numpy(), g[name + "/out_top5"])


def test_g6_box_nms_2d_and_errors(golden):
    g = golden("g6_box_nms.npz")
    p2 = synth.uniform("nms/2d", (30, 44), 0.0, 1.0)
    assert np.array_equal(xo.box_nms(torch.from_numpy(p2), 8, 0.5).numpy(), g["2d/out"])
    with pytest.raises(ValueError):
        xo.box_nms(torch.zeros(3, 4, 5), 8, 0.5)
    assert float(xo.box_nms(torch.zeros(1, 1, 16, 16), 8, 0.5).abs().sum()) == 0.0   # no candidates


def test_g6_integer_predicate():
    """for size 8 the IoU>0.1 test is (8-|dx|)(8-|dy|) >= 12; size 4: >= 3."""
    for size, thr in ((8, 12), (4, 3)):
        for dy in range(0, size + 1):
            for dx in range(0, size + 1):
                inter = max(0, size - dx) * max(0, size - dy)
                iou = np.float32(inter) / np.float32(2 * size * size - inter)
                assert (iou > np.float32(0.1)) == (inter >= thr and dx < size and dy < size)


def test_g7_interpolate(golden):
    g = golden("g7_interpolate.npz")
    desc = torch.from_numpy(synth.uniform("interp/desc", (16, 6, 9), -1, 1))
    out = xo.interpolate_descriptors(torch.from_numpy(g["kp"]), desc, 48, 72)
    np.testing.assert_allclose(out.numpy(), g["out"], atol=1e-7)


def test_g8_match(golden):
    g = golden("g8_match.npz")
    ref = {tuple(x) for x in g["matches"].tolist()}
    ms = xo.get_matches(g["d1"], g["d2"], "strict_mnn")
    assert {(m.queryIdx, m.trainIdx) for m in ms} == ref
    assert [m.queryIdx for m in ms] == sorted(m.queryIdx for m in ms)      # ascending queryIdx
    assert {(m.queryIdx, m.trainIdx) for m in xo.nnmatcher(g["d1"], g["d2"], 10.0)} == ref
    # legacy cross-check is a superset of strict mutual-NN
    leg = {(m.queryIdx, m.trainIdx) for m in xo.get_matches(g["d1"], g["d2"], "legacy_crosscheck")}
    assert ref <= leg
    assert xo.get_matches(g["d1"][:0], g["d2"]) == [] and xo.get_matches(g["d1"], g["d2"][:0]) == []


def test_g9_regnet(golden):
    g = golden("g9_regnet.npz")
    sd = _sd(synth.xpoint_exp1_config(256, 256, hm_head=True))
    hm = xo.regnet_forward(torch.from_numpy(g["enc_optical"]), torch.from_numpy(g["enc_thermal"]), sd)
    np.testing.assert_allclose(hm.numpy(), g["hm"], atol=1e-6)


def test_g11_superpoint(golden):
    g = golden("g11_superpoint.npz")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_superpoint_state_dict().items()}
    img = torch.from_numpy(synth.make_image(0, "optical", 64, 96)[None])
    with torch.no_grad():
        r = xo.superpoint_forward(img, sd)
    for k in ("logits", "desc", "prob"):
        np.testing.assert_allclose(r[k].numpy(), g[f"64x96/{k}"], atol=2e-6, err_msg=k)
    img = torch.from_numpy(synth.make_image(0, "optical", 240, 320)[None])
    with torch.no_grad():
        r = xo.superpoint_forward(img, sd)
    np.testing.assert_allclose(r["prob"].numpy(), g["240x320/prob"], atol=2e-6)


def test_state_spec_matches_reference_counts():
    cfg = synth.xpoint_exp1_config(480, 640)
    spec = synth.xpoint_state_spec(cfg)
    assert sum(int(np.prod(s)) if len(s) else 1 for s, _ in spec.values()) == 20404169
    cfg = synth.xpoint_exp1_config(256, 256, hm_head=True)
    spec = synth.xpoint_state_spec(cfg)
    assert len(spec) == 208 and sum(int(np.prod(s)) if len(s) else 1 for s, _ in spec.values()) == 20629651


def test_synth_is_bit_reproducible():
    u = synth.hash_uniform("probe", 5)
    assert u.dtype == np.float32
    assert [int(x * 16777216) for x in u] == [int(x * 16777216) for x in synth.hash_uniform("probe", 5)]
    # frozen known answers: any change to the generator invalidates every golden fixture
    assert [int(x * 16777216) for x in synth.hash_uniform("probe", 3)] == KNOWN_PROBE


KNOWN_PROBE = None  # filled below at import from the committed constant


def _known():
    return [int(x) for x in np.load(__file__.replace("test_oracle_golden.py", "golden/probe.npy"))]


KNOWN_PROBE = _known()


def test_g12_conv_xpoint(golden):
    g = golden("g12_conv_xpoint.npz")
    cfg = synth.multipoint_config()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_conv_xpoint_state_dict(cfg).items()}
    img = torch.from_numpy(synth.make_image(0, "optical", 64, 96)[None])
    with torch.no_grad():
        r = xo.forward_impl(img, sd)
    for k in ("prob", "desc", "encoder_output"):
        np.testing.assert_allclose(r[k].numpy(), g[f"64x96/{k}"], atol=2e-6, err_msg=k)


def test_g22_cross_scan_ops_all_layouts(golden):
    """The oracle's index-table restatement of cross_scan_fn / cross_merge_fn == the real reference (fixture g22) bit for bit: four layouts x
    scans {0, 1, 2} x one_by_one x {f32, f16, bf16} on random data, incl. the association of the merge adds and the reference's own odd check
    shape (27, 253, 57, 58) (csm_triton.py:670) by checksum."""
    from tests import csm_cases
    n = csm_cases.check(golden("g22_cross_scan_ops.npz"), xo.cross_scan_op, xo.cross_merge_op)
    assert n >= 150, n


def test_g23_threshold_matcher_and_knn_restatements(golden):
    """The oracle's ThresholdMatcher restatement (exact arithmetic) == the REAL reference class on the G8 descriptors (fixture g23; thresholds with a
    margin >> float32 BLAS noise), NNMatcher at its default threshold too; the k = 2 restatement agrees with a brute-force numpy fp64 ranking."""
    g = golden("g23_threshold_matcher.npz"); g8 = golden("g8_match.npz")
    d1, d2 = g8["d1"], g8["d2"]
    for thr in (1.25, 1.3):
        assert float(g[f"thr{thr}/margin"][0]) > 1e-5
        ms = xo.thresholdmatcher(d1, d2, thr)
        assert np.array_equal(np.array([[m.queryIdx, m.trainIdx] for m in ms], np.int32).reshape(-1, 2), g[f"thr{thr}/pairs"])
        np.testing.assert_allclose([m.distance for m in ms], g[f"thr{thr}/dist"], atol=2e-6)
    nn = xo.nnmatcher(d1, d2)
    assert np.array_equal(np.array([[m.queryIdx, m.trainIdx] for m in nn], np.int32).reshape(-1, 2), g["nn0.7/pairs"])
    idx, dist = xo.knn2(d1[:40], d2)
    dm = ((d1[:40, None, :].astype(np.float64) - d2[None].astype(np.float64)) ** 2).sum(-1)
    order = np.argsort(dm, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, order.astype(np.int32))
    np.testing.assert_allclose(dist, np.sqrt(np.take_along_axis(dm, order, 1)), rtol=1e-12)
    with pytest.raises(ValueError):
        xo.knn_ratio_matches(d1, d2[:1])
