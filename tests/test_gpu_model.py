"""GPU parity of the model-level path (xpoint_amd.models.XPoint / predict flows) against the golden vectors
produced by the REAL reference and against the oracle.  Tolerance from BASELINE.json north_star: keypoint
scores / descriptors within 1e-4; index results identical (stage-wise, see test_pipeline_stagewise)."""
import numpy as np
import pytest
import torch

from oracle import xpoint_oracle as xo
from xpoint_amd import synth

pytestmark = pytest.mark.gpu


def _lib():
    from xpoint_amd import _lib as L
    return L

TOL = 1e-4


def _net(cfg, sd=None):
    from xpoint_amd import models
    net = models.XPoint(cfg)
    net.load_state_dict(synth.make_torch_state_dict(cfg) if sd is None else sd, strict=True)
    return net.to("cuda").eval()


def _data(first, B, H, W):
    return synth.to_torch(synth.make_pair_batch(first, B, H, W), "cuda")


@pytest.mark.parametrize("gemm_mode", ["h2", "x3", "f32"])
@pytest.mark.parametrize("tag,H,W,B,vssm", [("tiny32_64x96", 64, 96, 1, {"EMBED_DIM": 32}), ("full_64x96", 64, 96, 2, None)])
def test_forward_vs_reference_golden(gpu_lib, golden, tag, H, W, B, vssm, gemm_mode):
    """All three f32-grade dense-layer back ends (split-fp16 on the f16 matrix pipe = default, split-bf16, exact-f32 MFMA)
    against the reference."""
    g = golden("g345_model.npz")
    cfg = synth.xpoint_exp1_config(H, W, vssm=vssm)
    net = _net(cfg)
    net.gemm_mode = gemm_mode
    with torch.no_grad():
        o, t, hm = net(_data(0, B, H, W))
    assert hm is None and o["logits"] is None
    for spec, r in (("optical", o), ("thermal", t)):
        for k in ("prob", "desc", "encoder_output"):
            ref = g[f"{tag}/{spec}/{k}"]
            got = r[k].cpu().numpy()
            assert got.shape == ref.shape, (k, got.shape, ref.shape)
            err = float(np.abs(got - ref).max())
            assert err < TOL, (spec, k, err)


def test_gemm_modes_agree_480x640(gpu_lib):
    """Full-size forward: the split-bf16 and the exact-f32 dense layers give the same network outputs to f32 rounding
    (both are f32-accurate; they differ only in summation order / sub-ulp truncation)."""
    H, W = 480, 640
    net = _net(synth.xpoint_exp1_config(H, W))
    img = _data(0, 1, H, W)["optical"]["image"]
    with torch.no_grad():
        net.gemm_mode = "f32"
        b = net.forward_raw(img, want_logits=True)
        b = {k: v.clone() for k, v in b.items() if v is not None}
        for mode in ("h2", "x3"):
            net.gemm_mode = mode
            a = net.forward_raw(img, want_logits=True)
            for k in ("prob", "desc_nhwc", "enc_nhwc", "logits_nhwc"):
                err = float((a[k] - b[k]).abs().max())
                assert err < 2e-5 * max(1.0, float(b[k].abs().max())), (mode, k, err)


def test_forward_224x320_and_end_to_end(gpu_lib, golden):
    from xpoint_amd.predict import predict_align_image_pair, predict_keypoints
    g = golden("g345_model.npz")
    tag, H, W = "full_224x320", 224, 320
    net = _net(synth.xpoint_exp1_config(H, W))
    data = _data(0, 1, H, W)
    with torch.no_grad():
        o, t, res = predict_align_image_pair(net, data)
    raw_o, raw_t, _ = net(data)
    assert float(np.abs(raw_o["prob"].cpu().numpy() - g[f"{tag}/optical/prob"]).max()) < TOL
    assert float(np.abs(raw_t["prob"].cpu().numpy() - g[f"{tag}/thermal/prob"]).max()) < TOL
    assert float(np.abs(raw_o["desc"].cpu().numpy() - g[f"{tag}/optical/desc"]).max()) < TOL
    # end to end: keypoint sets and mutual-NN index pairs equal the reference's, except where a decision sat inside the
    # 1e-4 parity budget — every differing element must be attributed to such a near-tie (tests/parity.py)
    from tests import parity
    from xpoint_amd import utils
    r = res[0]
    kp_m = {"optical": r["kp_optical"].cpu().numpy(), "thermal": r["kp_thermal"].cpu().numpy()}
    for spec, raw in (("optical", raw_o), ("thermal", raw_t)):
        rep, bad = parity.explain_keypoint_diff(kp_m[spec], g[f"{tag}/kp_{spec}"], raw["prob"][0, 0].cpu().numpy(), 0.015, 8, tol=TOL)
        print(parity.format_report(f"224x320 {spec} keypoints vs reference", rep))
        assert not bad, parity.format_report(spec, bad)
    # the reference's NNMatcher index pairs (strict mutual NN) against the HIP matcher on the HIP pipeline's own keypoints
    ms = utils.get_matches(r["desc_optical"], r["desc_thermal"], "nnmatcher", False, threshold=10.0)
    mine = np.array([[m.queryIdx, m.trainIdx] for m in ms]).reshape(-1, 2)
    dvol = {"optical": raw_o["desc_nhwc"][0], "thermal": raw_t["desc_nhwc"][0]}
    desc_of = lambda side, pts: utils.interpolate_descriptors_nhwc(torch.from_numpy(pts), dvol[side], H, W).cpu().numpy()
    rep, bad = parity.explain_match_diff(kp_m["optical"], kp_m["thermal"], g[f"{tag}/kp_optical"], g[f"{tag}/kp_thermal"], mine,
                                         g[f"{tag}/matches_nnmatcher"], desc_of, tol=TOL)
    print(parity.format_report("224x320 mutual-NN pairs vs reference NNMatcher", rep))
    assert not bad, parity.format_report("matches", bad)
    assert len(rep) <= 2 * max(2, len(mine) // 50)          # explained differences stay rare
    kpo, kpt = predict_keypoints(net, data)
    rep, bad = parity.explain_keypoint_diff(kpo[0].cpu().numpy(), g[f"{tag}/kp_optical"], raw_o["prob"][0, 0].cpu().numpy(), 0.015, 8, tol=TOL)
    assert not bad, parity.format_report("predict_keypoints", bad)


@pytest.mark.parametrize("schedule", ["alternate", "split2"])
def test_c2_batch8_end_to_end_indices_vs_reference(gpu_lib, golden, capsys, schedule):
    """BASELINE config C2 on the GPU, end to end, against the REAL reference (tests/golden/g15_c2_batch8.npz, flow of
    predict_align_image_pair.py:185-260): 8 pairs of 480x640 through the overlapped PairPipeline — "alternate" is the bench
    configuration (whole-batch encoders of consecutive steps on two streams), "split2" the image-group schedule.
    Keypoint lists and mutual-NN index pairs must be IDENTICAL to the reference's, except elements attributed, one by one, to a
    decision inside the 1e-4 parity budget (score vs threshold / vs an overlapping competitor, descriptor-distance gap);
    the near-tie report is printed.  Any unexplained difference fails."""
    from tests import parity
    from xpoint_amd import utils
    from xpoint_amd.predict import PairPipeline
    g = golden("g15_c2_batch8.npz")
    B, H, W = [int(v) for v in g["meta"]]
    assert (B, H, W) == (8, 480, 640)
    net = _net(synth.xpoint_exp1_config(H, W))
    data = _data(0, B, H, W)
    pipe = PairPipeline(net, B, H, W, cap=8192, overlap=True, split_encoder=2 if schedule == "split2" else 0, alternate_encoders=schedule == "alternate")
    with torch.no_grad():
        for _ in range(2):      # second call: the other half of the double-buffered outputs, steady-state schedule
            pipe.run(data["optical"]["image"], data["thermal"]["image"], data["optical"]["valid_mask"], data["thermal"]["valid_mask"])
        out = pipe.fetch()
    prob = pipe.raw["prob"].cpu().numpy()
    dvol = pipe.raw["desc_nhwc"]
    n_kp_diff = n_m_diff = n_kp = n_m = 0
    lines = []
    for i in range(B):
        kp_m = {"optical": out[i]["kp_optical"].numpy(), "thermal": out[i]["kp_thermal"].numpy()}
        for spec, img in (("optical", i), ("thermal", B + i)):
            ref = g[f"p{i}/kp_{spec}"].astype(np.int64)
            n_kp += len(ref)
            # the reference's own scores at its keypoints agree with this forward to the parity bar
            sc = prob[img][ref[:, 0], ref[:, 1]]
            assert float(np.abs(sc - g[f"p{i}/score_{spec}"]).max()) < TOL
            if np.array_equal(kp_m[spec], ref):
                continue
            rep, bad = parity.explain_keypoint_diff(kp_m[spec], ref, prob[img], 0.015, 8, tol=TOL)
            n_kp_diff += len(rep)
            lines.append(parity.format_report(f"pair {i} {spec} keypoints", rep))
            assert not bad, parity.format_report(f"pair {i} {spec}: UNEXPLAINED keypoint differences", bad)
        mine = np.stack([out[i]["match_q"], out[i]["match_t"]], 1).astype(np.int64)
        ref_m = g[f"p{i}/matches"].astype(np.int64)
        n_m += len(ref_m)
        same_kp = all(np.array_equal(kp_m[s], g[f"p{i}/kp_{s}"]) for s in ("optical", "thermal"))
        if same_kp and np.array_equal(mine, ref_m):
            continue
        vol = {"optical": dvol[i], "thermal": dvol[B + i]}
        desc_of = lambda side, pts: utils.interpolate_descriptors_nhwc(torch.from_numpy(pts), vol[side], H, W).cpu().numpy()
        rep, bad = parity.explain_match_diff(kp_m["optical"], kp_m["thermal"], g[f"p{i}/kp_optical"], g[f"p{i}/kp_thermal"], mine, ref_m,
                                             desc_of, tol=TOL)
        n_m_diff += len(rep)
        lines.append(parity.format_report(f"pair {i} mutual-NN pairs", rep))
        assert not bad, parity.format_report(f"pair {i}: UNEXPLAINED match differences", bad)
    with capsys.disabled():
        print(f"\nC2 batch-8 end-to-end vs reference: {n_kp} keypoints, {n_kp_diff} differ (all near-tie explained); "
              f"{n_m} mutual-NN pairs, {n_m_diff} differ (all explained)")
        print("\n".join(lines))
    assert n_kp_diff <= n_kp // 200 and n_m_diff <= n_m // 50          # attributed differences stay rare (< 0.5 % / 2 %)
    # HARD pins of what the default (split-fp16) back end measures today on this fixture: every one of the 64 876 keypoints identical,
    # 2 differing elements among the 9 934 mutual-NN pairs (one pair of pair 6, distance gap 4.75e-7).  A kernel change that moves a rounding shows up here.
    assert n_kp == 64876 and n_kp_diff == 0, (n_kp, n_kp_diff)
    assert n_m == 9934 and n_m_diff <= 2, (n_m, n_m_diff)


def test_pipeline_stagewise_exact(gpu_lib, golden):
    """Index exactness is defined stage-wise: feed the REFERENCE's stage k-1 output into HIP
    stage k and require identical indices."""
    from xpoint_amd import utils
    g = golden("g345_model.npz")
    tag, H, W = "full_224x320", 224, 320
    prob = torch.from_numpy(g[f"{tag}/optical/prob"])
    nms = utils.box_nms(prob.cuda(), 8, 0.015)
    kp = torch.nonzero((nms[0].squeeze() > 0.015).float()).cpu()
    assert np.array_equal(kp.numpy(), g[f"{tag}/kp_optical"])               # NMS + extraction: identical keypoints
    kpd, cnt = utils.extract_keypoints(nms, 0.015)
    assert int(cnt[0]) == len(kp) and torch.equal(kpd[0, :len(kp)].cpu().long(), kp)
    desc = torch.from_numpy(g[f"{tag}/optical/desc"])[0]
    d = utils.interpolate_descriptors(kp.cuda(), desc.cuda(), H, W)
    np.testing.assert_allclose(d.cpu().numpy(), g[f"{tag}/desc_optical_sampled"], atol=1e-6)
    top = utils.box_nms(prob.cuda(), 8, 0.015, keep_top_k=100)
    assert np.array_equal(torch.nonzero(top[0].squeeze() > 0.015).cpu().numpy(), g[f"{tag}/kp_optical_top100"])


def test_batched_pipeline_matches_per_pair_flow(gpu_lib):
    """PairPipeline (batched, device resident, async NMS) == predict_align_image_pair (per pair, reference call
    sequence) on the same inputs; and its match indices equal the oracle's exact matcher on ITS descriptors."""
    from xpoint_amd.predict import PairPipeline, predict_align_image_pair
    H, W, B = 96, 128, 3
    net = _net(synth.xpoint_exp1_config(H, W))
    data = _data(5, B, H, W)
    data["optical"]["valid_mask"][:, :, :10] = False                         # exercise the mask
    with torch.no_grad():
        _, _, res = predict_align_image_pair(net, data)
        pipe = PairPipeline(net, B, H, W, cap=2048)
        out = pipe.run(data["optical"]["image"], data["thermal"]["image"], data["optical"]["valid_mask"],
                       data["thermal"]["valid_mask"]).fetch()
    for i in range(B):
        assert torch.equal(out[i]["kp_optical"], res[i]["kp_optical"].cpu())
        assert torch.equal(out[i]["kp_thermal"], res[i]["kp_thermal"].cpu())
        np.testing.assert_allclose(out[i]["desc_optical"].numpy(), res[i]["desc_optical"].cpu().numpy(), atol=1e-6)
        assert list(zip(out[i]["match_q"].tolist(), out[i]["match_t"].tolist())) == \
               [(m.queryIdx, m.trainIdx) for m in res[i]["matches"]]
        oms = xo.get_matches(out[i]["desc_optical"].numpy(), out[i]["desc_thermal"].numpy())
        assert [(m.queryIdx, m.trainIdx) for m in oms] == list(zip(out[i]["match_q"].tolist(), out[i]["match_t"].tolist()))
        assert int(out[i]["kp_optical"][:, 0].min()) >= 10


def test_full_size_480x640_vs_reference_summary(gpu_lib, golden):
    """BASELINE config-2 image size against the reference's 480x640 run (strided samples + checksums)."""
    g = golden("g10_full480x640.npz")
    H, W = 480, 640
    net = _net(synth.xpoint_exp1_config(H, W))
    with torch.no_grad():
        o, t, _ = net(_data(0, 1, H, W))
    for spec, r in (("optical", o), ("thermal", t)):
        p = r["prob"][0, 0].cpu().numpy()
        assert float(np.abs(p[::16] - g[f"{spec}/prob_rows"]).max()) < TOL
        d = r["desc"][0, :, ::6, ::8].cpu().numpy()
        assert float(np.abs(d - g[f"{spec}/desc_cols"]).max()) < TOL
        enc = r["encoder_output"].double()
        ref_sum = g[f"{spec}/enc_sum"]
        assert abs(float(enc.abs().sum()) - ref_sum[1]) < 1e-5 * ref_sum[1]
        n_cand = int((r["prob"] > 0.015).sum())
        assert abs(n_cand - int(g[f"{spec}/n_candidates"][0])) <= 20


def test_api_surface_and_errors(gpu_lib):
    from xpoint_amd import models
    cfg = synth.xpoint_exp1_config(64, 96)
    net = _net(cfg)
    assert net.takes_pair() is True and net.get_encoder_downsample_ratio() == 8
    with pytest.raises(ValueError):
        net.set_force_return_logits(1)
    net.set_force_return_logits(True)
    with torch.no_grad():
        o, t, _ = net(_data(0, 1, 64, 96))
    assert o["prob"] is None and tuple(o["logits"].shape) == (1, 65, 8, 12)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}
    with torch.no_grad():
        _, ref_logits = xo.detector_head(xo.vssm_forward(synth.to_torch(synth.make_pair_batch(0, 1, 64, 96))["optical"]["image"], sd), sd, True)
    assert float((o["logits"].cpu() - ref_logits).abs().max()) < 1e-3      # logits are ~8x the prob scale (detector gain)
    net.set_force_return_logits(False)
    with pytest.raises(RuntimeError):                                      # CPU tensors: no CPU fallback
        net({"optical": {"image": torch.zeros(1, 1, 64, 96)}, "thermal": {"image": torch.zeros(1, 1, 64, 96)}})
    with pytest.raises(RuntimeError):                                      # 240x320 is not a valid VMamba size
        net({"optical": {"image": torch.zeros(1, 1, 240, 320, device="cuda")}, "thermal": {"image": torch.zeros(1, 1, 240, 320, device="cuda")}})
    bad = dict(sd); bad.pop("encoder.patch_embed.0.bias")
    with pytest.raises(RuntimeError):
        models.XPoint(cfg).load_state_dict(bad, strict=True)
    r = models.XPoint(cfg).load_state_dict(bad, strict=False)
    assert r.missing_keys == ["encoder.patch_embed.0.bias"]
    with pytest.raises(NotImplementedError):
        models.XPoint({"use_attention": {"check": False}})


def test_config4_1024_topk_4096_pipeline(gpu_lib):
    """BASELINE configs[3]: 1024x1024 pair, keep_top_k = 4096 keypoints per image, dense 4k x 4k x 256 matching.
    Size-independent properties: exactly <= 4096 keypoints sorted row-major, top-k = the k best NMS survivors, match
    indices identical to the exact (fp64) matcher on the pipeline's own descriptors."""
    from xpoint_amd.predict import PairPipeline
    from xpoint_amd import utils
    H = W = 1024
    net = _net(synth.xpoint_exp1_config(H, W))
    data = _data(3, 1, H, W)
    with torch.no_grad():
        pipe = PairPipeline(net, 1, H, W, cap=16384, cfg_prediction={"topk": 4096}, nms_sweeps=8)
        out = pipe.run(data["optical"]["image"], data["thermal"]["image"]).fetch()[0]
        prob = pipe.raw["prob"]
        full = utils.box_nms(prob[0:1].unsqueeze(1), 8, 0.015)                       # all survivors (sync NMS)
    n = len(out["kp_optical"])
    assert 0 < n <= 4096 and len(out["kp_thermal"]) <= 4096
    kp = out["kp_optical"]
    lin = kp[:, 0] * W + kp[:, 1]
    assert bool((lin[1:] > lin[:-1]).all())                                             # row-major order
    surv = full[0, 0][full[0, 0] > 0.015]
    if surv.numel() > 4096:
        kth = torch.sort(surv, descending=True).values[4095]
        kept = full[0, 0][kp[:, 0].cuda(), kp[:, 1].cuda()]
        assert n == 4096 and float(kept.min()) >= float(kth)                            # the k best survivors
    oms = xo.get_matches(out["desc_optical"].numpy(), out["desc_thermal"].numpy())
    assert [(m.queryIdx, m.trainIdx) for m in oms] == list(zip(out["match_q"].tolist(), out["match_t"].tolist()))


@pytest.mark.parametrize("overlap,split", [(False, 0), (True, 0), (True, 2), (True, -1)])
def test_hipgraph_replay_equals_eager(gpu_lib, overlap, split):
    """BASELINE configs[4] asks for a hipGraph-captured forward: a captured PairPipeline step replays to the same results as the eager
    step, with new inputs at every replay — on one stream, and for the overlapped pipeline (per output buffer one graph per encoder
    image group + one for detection / matching, chained by events: the cross-step overlap survives capture)."""
    from xpoint_amd.predict import PairPipeline
    H, W, B = 96, 128, 2
    net = _net(synth.xpoint_exp1_config(H, W))
    seq = [_data(s, B, H, W) for s in (0, 9, 4, 7)]
    with torch.no_grad():
        eager = PairPipeline(net, B, H, W, cap=2048)
        refs = [eager.run(d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"]).fetch() for d in seq]
        pipe = PairPipeline(net, B, H, W, cap=2048, overlap=overlap, split_encoder=max(split, 0), alternate_encoders=split < 0)   # -1: alternating encoder streams
        d0 = seq[0]
        replay = pipe.capture(d0["optical"]["image"], d0["thermal"]["image"], d0["optical"]["valid_mask"], d0["thermal"]["valid_mask"])
        for d, ref in zip(seq[1:], refs[1:]):
            replay(d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"])
            got = pipe.fetch()
            for a, b in zip(got, ref):
                assert torch.equal(a["kp_optical"], b["kp_optical"]) and torch.equal(a["desc_thermal"], b["desc_thermal"])
                assert a["match_q"].tolist() == b["match_q"].tolist() and a["match_t"].tolist() == b["match_t"].tolist()
        # back-to-back replays without a fetch in between (both output buffers in flight), last result checked
        for d in seq:
            replay(d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"])
        got = pipe.fetch()
        for a, b in zip(got, refs[-1]):
            assert torch.equal(a["kp_thermal"], b["kp_thermal"]) and a["match_t"].tolist() == b["match_t"].tolist()


@pytest.mark.parametrize("split", [0, 2, 4, -1])
def test_overlapped_pipeline_equals_single_stream(gpu_lib, split):
    """overlap=True (two streams, encoder of call i+1 behind the detection / matching of call i, double-buffered encoder
    outputs) returns exactly the single-stream results: a sequence of calls with different inputs, the caller reusing
    its input tensors right after run() returns, masks included; also with the encoder split into image groups on
    several streams, or (split = -1) with the whole-batch encoders of consecutive calls alternating between two streams."""
    from xpoint_amd.predict import PairPipeline
    H, W, B = 96, 128, 2
    net = _net(synth.xpoint_exp1_config(H, W))
    seq = [_data(s, B, H, W) for s in (0, 5, 9, 3)]
    with torch.no_grad():
        single = PairPipeline(net, B, H, W, cap=2048)
        refs = []
        for d in seq:
            refs.append(single.run(d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"]).fetch())
        pipe = PairPipeline(net, B, H, W, cap=2048, overlap=True, split_encoder=max(split, 0), alternate_encoders=split < 0)
        io, it = torch.empty_like(seq[0]["optical"]["image"]), torch.empty_like(seq[0]["thermal"]["image"])
        mo, mt = torch.empty_like(seq[0]["optical"]["valid_mask"]), torch.empty_like(seq[0]["thermal"]["valid_mask"])
        for n_calls in (1, 2, 3, 4):
            for d in seq[:n_calls]:          # back-to-back calls, inputs refilled in place without synchronising
                io.copy_(d["optical"]["image"]); it.copy_(d["thermal"]["image"])
                mo.copy_(d["optical"]["valid_mask"]); mt.copy_(d["thermal"]["valid_mask"])
                pipe.run(io, it, mo, mt)
            got = pipe.fetch()               # results of the LAST call
            for a, b in zip(got, refs[n_calls - 1]):
                assert torch.equal(a["kp_optical"], b["kp_optical"]) and torch.equal(a["kp_thermal"], b["kp_thermal"])
                assert torch.equal(a["desc_optical"], b["desc_optical"]) and torch.equal(a["desc_thermal"], b["desc_thermal"])
                assert a["match_q"].tolist() == b["match_q"].tolist() and a["match_t"].tolist() == b["match_t"].tolist()


def test_batch_invariance_480x640(gpu_lib):
    """at the BENCH size an image's result must not depend on which other images share the call.  Pair 0 of the C2 batch
    (i) alone, (ii) inside the batch of 8 on the overlapped alternating-encoder schedule (what bench.py times), (iii) inside the batch of 8 with the
    encoder split into two image groups (`split2`: M = 2400 rows at stage 3) and (iv) through the eager forward: prob, the descriptor volume, keypoints,
    sampled descriptors and match indices bit-identical (torch.equal).  The reference computes every image independently (XPoint.py:283-323).
    Round 3's ping-pong GEMM was picked by a tile count (a batch quantity) and broke exactly this at 480 x 640 while the 96 x 128 tests stayed green."""
    from xpoint_amd.predict import PairPipeline
    H, W = 480, 640
    net = _net(synth.xpoint_exp1_config(H, W))
    d8, d1 = _data(0, 8, H, W), _data(0, 1, H, W)
    args = lambda d: (d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"])
    with torch.no_grad():
        runs = {}
        for name, B, d, kw in (("alone", 1, d1, dict()), ("alone_overlap", 1, d1, dict(overlap=True, alternate_encoders=True)),
                               ("batch8", 8, d8, dict(overlap=True, alternate_encoders=True)), ("split2", 8, d8, dict(overlap=True, split_encoder=2)),
                               ("single_stream8", 8, d8, dict())):
            pipe = PairPipeline(net, B, H, W, cap=8192, **kw)
            for _ in range(2):
                pipe.run(*args(d))
            got = pipe.fetch()[0]
            runs[name] = dict(prob=pipe.raw["prob"][[0, B]].clone(), desc=pipe.raw["desc_nhwc"][[0, B]].clone(), **got)
        o, t, _ = net(d1)
        o8, t8, _ = net(d8)
    ref = runs["alone"]
    for name, r in runs.items():
        for k in ("prob", "desc", "kp_optical", "kp_thermal", "desc_optical", "desc_thermal"):
            assert torch.equal(r[k], ref[k]), (name, k, float((r[k].float() - ref[k].float()).abs().max()) if r[k].shape == ref[k].shape else "shape")
        assert r["match_q"].tolist() == ref["match_q"].tolist() and r["match_t"].tolist() == ref["match_t"].tolist(), name
    assert torch.equal(o["prob"][0, 0], ref["prob"][0]) and torch.equal(t["prob"][0, 0], ref["prob"][1])
    assert torch.equal(o8["prob"][0], o["prob"][0]) and torch.equal(t8["desc"][0], t["desc"][0]) and torch.equal(o8["encoder_output"][0], o["encoder_output"][0])


def test_pipeline_edge_cases(gpu_lib):
    """Empty inputs (everything masked), capacity overflow and an under-iterated async NMS are detected, not silent."""
    from xpoint_amd.predict import PairPipeline
    H, W, B = 96, 128, 1
    net = _net(synth.xpoint_exp1_config(H, W))
    data = _data(2, B, H, W)
    zero = torch.zeros_like(data["optical"]["valid_mask"])
    with torch.no_grad():
        out = PairPipeline(net, B, H, W, cap=512).run(data["optical"]["image"], data["thermal"]["image"], zero, zero).fetch()[0]
        assert len(out["kp_optical"]) == 0 and len(out["kp_thermal"]) == 0 and len(out["match_q"]) == 0
        half = data["thermal"]["valid_mask"].clone(); half[..., : W // 2] = False
        out = PairPipeline(net, B, H, W, cap=512).run(data["optical"]["image"], data["thermal"]["image"], zero, half).fetch()[0]
        assert len(out["kp_optical"]) == 0 and len(out["kp_thermal"]) > 0 and len(out["match_q"]) == 0
        small = PairPipeline(net, B, H, W, cap=8).run(data["optical"]["image"], data["thermal"]["image"])
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="exceed capacity"):
            small.verify()
        # input contract: one uint8 image (0..255) beside one float image ([0, 1]) would silently run on two scales — refused
        u8 = (data["optical"]["image"] * 255.0).to(torch.uint8)
        with pytest.raises(ValueError, match="both images as uint8"):
            PairPipeline(net, B, H, W, cap=512).run(u8, data["thermal"]["image"])
        with pytest.raises(ValueError, match="exactly"):
            PairPipeline(net, B, H, W, cap=512).run(u8[..., :-1], u8[..., :-1])


def test_async_nms_reports_non_convergence(gpu_lib):
    """Forms of xp_box_nms.  An image whose bit masks fit one workgroup's LDS (every model size up to ~1000 x 700) is ONE band: local maxima, one
    suppression pass, then a per-image finisher that iterates to the fixed point INSIDE its launch — no sweep count, always converged, also on a
    640-long suppression chain.  Larger images are split into row bands whose finishers see their neighbours as the launch found them; a chain that
    crosses band borders needs one launch per border: the stream-ordered mode enqueues a fixed number and reports what is left, the synchronous mode
    runs until nothing is."""
    import ctypes
    from xpoint_amd import _lib as L
    lib = L.load()
    for (H, W, vertical, sweeps_needed) in ((64, 640, False, False), (2048, 640, True, True)):
        p = torch.zeros(1, H, W)
        if vertical:
            p[0, :, 300:304] = (0.5 + 0.4 * torch.arange(H) / H)[:, None]   # monotone ridge down the image: the chain of decisions crosses every band border
        else:
            p[0, 30:34, :] = 0.5 + 0.4 * torch.arange(W) / W                # monotone ridge along a row band: a 640-long chain inside one band
        pd = p.cuda(); out = torch.empty_like(pd)
        ws = torch.empty(lib.xp_box_nms_workspace_bytes(1, H, W, 1), dtype=torch.uint8, device="cuda")
        left = ctypes.c_int(-1)
        L.check(lib.xp_box_nms(L.ptr(pd), L.ptr(out), L.ptr(ws), ws.numel(), 1, H, W, 8.0, 0.015, 0.1, 0, 1, 1, None, L.current_stream()), "nms")
        L.check(lib.xp_box_nms_check(L.ptr(ws), 1, H, W, ctypes.byref(left), L.current_stream()), "check")
        ref = xo.box_nms(p.unsqueeze(1), 8, 0.015)[:, 0]
        if sweeps_needed:
            assert left.value > 0                                    # one launch cannot carry the chain across the band borders, and says so
            L.check(lib.xp_box_nms(L.ptr(pd), L.ptr(out), L.ptr(ws), ws.numel(), 1, H, W, 8.0, 0.015, 0.1, 0, 1, 0, ctypes.byref(left), L.current_stream()), "nms")
        else:
            assert left.value == 0
        assert torch.equal(out.cpu(), ref)


def test_multispectral_two_encoder_routing(gpu_lib, golden):
    """`multispectral: true` (XPoint.py:98-100, 284-305) — optical images through encoder_optical,
    thermal through encoder_thermal, shared heads — against the real reference on the reduced VMamba model: the pair
    forward, a mixed-flag batch, and the batched PairPipeline (single stream and overlapped)."""
    from xpoint_amd.predict import PairPipeline, predict_align_image_pair
    g = golden("g14_multispectral.npz")
    H, W, B = 64, 96, 2
    cfg = synth.xpoint_exp1_config(H, W, vssm={"EMBED_DIM": 32})
    cfg["multispectral"] = True
    net = _net(cfg)
    data = _data(0, B, H, W)
    with torch.no_grad():
        o, t, _ = net(data)
        for spec, r in (("optical", o), ("thermal", t)):
            for k in ("prob", "desc", "encoder_output"):
                err = float(np.abs(r[k].cpu().numpy() - g[f"pair/{spec}/{k}"]).max())
                assert err < TOL, (spec, k, err)
        mixed = {"image": torch.cat([data["optical"]["image"][:1], data["thermal"]["image"][:1], data["optical"]["image"][1:]], 0),
                 "is_optical": torch.tensor([[True], [False], [True]]).cuda()}
        r = net.forward_impl(mixed)
        assert float(np.abs(r["prob"].cpu().numpy() - g["mixed/prob"]).max()) < TOL
        assert float(np.abs(r["desc"].cpu().numpy() - g["mixed/desc"]).max()) < TOL
        # the encoders really differ, and the batched pipelines route the two halves like the per-pair flow does
        assert float((o["prob"] - t["prob"]).abs().max()) > 0.1
        _, _, ref = predict_align_image_pair(net, _data(0, B, H, W))
        for ov in (False, True):
            got = PairPipeline(net, B, H, W, cap=1024, overlap=ov, split_encoder=2 if ov else 0).run(
                data["optical"]["image"], data["thermal"]["image"], data["optical"]["valid_mask"], data["thermal"]["valid_mask"]).fetch()
            for a, b in zip(got, ref):
                assert torch.equal(a["kp_optical"], b["kp_optical"].cpu()) and torch.equal(a["kp_thermal"], b["kp_thermal"].cpu())
                assert a["match_q"].tolist() == [m.queryIdx for m in b["matches"]]
    with pytest.raises(RuntimeError):
        net.forward_raw(data["optical"]["image"])          # flags are mandatory for a two-encoder model


@pytest.mark.parametrize("mode,nprod,lo,hi", [("x2", 3, 1e-7, 1e-4), ("bf16", 1, 1e-3, 1e-1)])
def test_precision_classes_match_their_cpu_emulation(gpu_lib, mode, nprod, lo, hi):
    """gemm_mode "x2" / "bf16" (xp_set_dense_products 3 / 1; SURVEY.md 8(f) rank 3: the mixed-precision class).  A network whose dense
    operands are truncated is sensitive to perturbations far below the truncation (a 1e-7 relative change of the input image moves the
    CPU emulation's own output by as much as the HIP path differs from it), so element-wise agreement with the emulation is asserted
    at kernel level (test_gpu_kernels.py::test_dense_precision_classes_kernel_level); here: the HIP forward deviates from the fp32
    reference by the same amount as the oracle's restatement of the class (oracle.DENSE_PRODUCTS) does, is closer to that restatement
    than to the reference, and the default class is untouched."""
    from xpoint_amd import models
    H, W, B = 64, 96, 1
    cfg = synth.xpoint_exp1_config(H, W)
    sd_np = synth.make_state_dict(cfg)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
    net = models.XPoint(cfg); net.load_state_dict(sd, strict=True); net.to("cuda:0").eval()
    data = synth.to_torch(synth.make_pair_batch(0, B, H, W), "cuda:0")
    data_cpu = synth.to_torch(synth.make_pair_batch(0, B, H, W))
    with torch.no_grad():
        ref, _, _ = xo.xpoint_forward(data_cpu, sd)
        xo.DENSE_PRODUCTS = nprod
        try:
            emu, _, _ = xo.xpoint_forward(data_cpu, sd)
        finally:
            xo.DENSE_PRODUCTS = 6
        net.gemm_mode = mode
        got, _, _ = net(data)
        net.gemm_mode = "x3"
        base, _, _ = net(data)
    assert _lib().load().xp_get_dense_products() == 6
    rms = lambda a, b: float((a.double() - b.double()).pow(2).mean().sqrt())
    for key in ("prob", "encoder_output"):
        g, e, r = got[key].cpu(), emu[key], ref[key]
        assert 0.5 * rms(e, r) < rms(g, r) < 2.0 * rms(e, r), (key, rms(g, r), rms(e, r))      # the same error class ...
        assert rms(g, e) < rms(g, r), (key, rms(g, e), rms(g, r))                              # ... correlated with its restatement
    e_ref = float((got["prob"].cpu() - ref["prob"]).abs().max())
    e_base = float((base["prob"].cpu() - ref["prob"]).abs().max())
    assert lo < e_ref < hi, e_ref                       # a genuinely different precision class ...
    assert e_base < 1e-4 and e_base < e_ref              # ... and the default one still meets the bar


def test_cli_flows_end_to_end(gpu_lib, tmp_path):
    """python -m xpoint_amd.cli align / keypoints with the reference scripts' -y / -m / -v / -i options (predict_align_image_pair.py:24-37):
    YAML + <model-dir>/params.yaml + <version>.model + a folder dataset -> the same keypoints / matches as the library functions."""
    import os
    import yaml
    from PIL import Image
    from xpoint_amd import cli, datasets, models, predict
    H, W = 64, 96
    cfg = synth.xpoint_exp1_config(H, W)
    rng = np.random.default_rng(3)
    os.makedirs(tmp_path / "data" / "optical"); os.makedirs(tmp_path / "data" / "thermal"); os.makedirs(tmp_path / "model")
    for i in range(2):
        Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(tmp_path / "data" / "optical" / f"s{i}.png")
        Image.fromarray(rng.integers(0, 256, (H, W), dtype=np.uint8)).save(tmp_path / "data" / "thermal" / f"s{i}.png")
    yaml.safe_dump({"dataset": {"type": "ImagePairDataset", "foldername": str(tmp_path / "data"), "height": H, "width": W},
                    "prediction": {"detection_threshold": 0.015, "nms": 8, "cpu_nms": True, "topk": 0, "reprojection_threshold": 3,
                                   "allow_gpu": True, "batchsize": 1, "num_worker": 0}}, open(tmp_path / "cfg.yaml", "w"))
    yaml.safe_dump({"model": cfg}, open(tmp_path / "model" / "params.yaml", "w"))
    sd = {"net__" + k: torch.from_numpy(np.array(v)) for k, v in synth.make_state_dict(cfg).items()}     # prefixed keys: fix_model_weigth_keys (utils.py:240-246)
    torch.save(sd, tmp_path / "model" / "latest.model")
    out = cli.main(["align", "-y", str(tmp_path / "cfg.yaml"), "-m", str(tmp_path / "model"), "-v", "latest", "-i", "0", "-n", "2", "-e",
                    "-o", str(tmp_path / "out.npz")])
    assert os.path.exists(tmp_path / "out.npz") and "1/matches" in out and out["0/H_est"].shape == (3, 3)
    # the same through the library
    conf = cli.load_config(str(tmp_path / "cfg.yaml"), str(tmp_path / "model"))
    ds, net = cli.build(conf, str(tmp_path / "model"), "latest", "cuda:0")
    with torch.no_grad():
        _, _, res = predict.predict_align_image_pair(net, ds.load_batch([1], "cuda:0"), conf["prediction"] and {k: conf["prediction"][k] for k in ("detection_threshold", "nms", "cpu_nms", "topk")})
    assert np.array_equal(out["1/kp_optical"], res[0]["kp_optical"].cpu().numpy())
    assert out["1/matches"].tolist() == [[m.queryIdx, m.trainIdx] for m in res[0]["matches"]]
    outk = cli.main(["keypoints", "-y", str(tmp_path / "cfg.yaml"), "-m", str(tmp_path / "model"), "-i", "1"])
    assert outk["1/kp_thermal"].shape[1] == 2


def test_c4_1024_topk4096_end_to_end_vs_reference(gpu_lib, golden, capsys):
    """BASELINE config C4 on the GPU against the REAL reference (tests/golden/g16_c4_1024.npz): one 1024x1024 pair, box NMS with
    keep_top_k 4096, 4096 x 4096 x 256 mutual-NN match.  Forward within 1e-4 (strided samples); keypoints and index pairs identical to
    the reference's up to attributed near-ties (incl. the top-k cut)."""
    from tests import parity
    from xpoint_amd import utils
    from xpoint_amd.predict import PairPipeline
    g = golden("g16_c4_1024.npz")
    _, H, W, K = [int(v) for v in g["meta"]]
    net = _net(synth.xpoint_exp1_config(H, W))
    data = _data(0, 1, H, W)
    pipe = PairPipeline(net, 1, H, W, cap=16384, cfg_prediction=dict(topk=K), nms_sweeps=12, overlap=True, split_encoder=2)
    with torch.no_grad():
        for _ in range(2):
            pipe.run(data["optical"]["image"], data["thermal"]["image"], data["optical"]["valid_mask"], data["thermal"]["valid_mask"])
        out = pipe.fetch()[0]
    prob = pipe.raw["prob"].cpu().numpy()
    desc = pipe.raw["desc_nhwc"].permute(0, 3, 1, 2).cpu().numpy()
    lines, kp_m = [], {}
    for j, spec in enumerate(("optical", "thermal")):
        assert float(np.abs(prob[j][::32] - g[f"prob_rows_{spec}"]).max()) < TOL
        assert float(np.abs(desc[j][:, ::16, ::16] - g[f"desc_cols_{spec}"]).max()) < TOL
        kp_m[spec] = out[f"kp_{spec}"].numpy()
        ref = g[f"kp_{spec}"].astype(np.int64)
        assert len(kp_m[spec]) == K == len(ref)
        cut = float(np.sort(prob[j][kp_m[spec][:, 0], kp_m[spec][:, 1]])[0])          # score of the last survivor kept
        rep, bad = parity.explain_keypoint_diff(kp_m[spec], ref, prob[j], 0.015, 8, tol=TOL, topk_cut=cut)
        lines.append(parity.format_report(f"1024x1024 {spec} keypoints (top-k {K})", rep))
        assert len(rep) == 0, lines[-1]          # hard pin: all 4 096 keypoints are the reference's today
        assert not bad, parity.format_report(spec, bad)
    mine = np.stack([out["match_q"], out["match_t"]], 1).astype(np.int64)
    vol = {"optical": pipe.raw["desc_nhwc"][0], "thermal": pipe.raw["desc_nhwc"][1]}
    desc_of = lambda side, pts: utils.interpolate_descriptors_nhwc(torch.from_numpy(pts), vol[side], H, W).cpu().numpy()
    rep, bad = parity.explain_match_diff(kp_m["optical"], kp_m["thermal"], g["kp_optical"], g["kp_thermal"], mine, g["matches"].astype(np.int64), desc_of, tol=TOL)
    lines.append(parity.format_report(f"1024x1024 mutual-NN pairs ({len(mine)} mine, {len(g['matches'])} reference)", rep))
    with capsys.disabled():
        print("\n" + "\n".join(lines))
    assert not bad, parity.format_report("matches", bad)
    assert len(rep) <= max(4, len(mine) // 50)
    assert len(rep) == 0 and len(mine) == len(g["matches"])          # hard pin: the 1 108 pairs are the reference's today, element for element


@pytest.mark.parametrize("overlap", [False, True])
def test_download_async_equals_fetch(gpu_lib, overlap):
    """Streaming use of PairPipeline: download_async() enqueues the result copies behind the step and the next run() before the host waits;
    the pinned buffers of step i (consumed after step i+1 was enqueued) hold exactly what a synchronous fetch() of step i returns — also
    when the images come from pinned host memory."""
    from xpoint_amd.predict import PairPipeline
    H, W, B = 96, 128, 2
    net = _net(synth.xpoint_exp1_config(H, W))
    seq = [_data(s, B, H, W) for s in (3, 8, 1)]
    with torch.no_grad():
        ref_pipe = PairPipeline(net, B, H, W, cap=2048)
        refs = [ref_pipe.run(d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"]).fetch() for d in seq]
        pipe = PairPipeline(net, B, H, W, cap=2048, overlap=overlap, alternate_encoders=overlap)
        got, prev = [], None

        def consume(bufs, ev):
            ev.synchronize()
            got.append({k: v.clone() for k, v in bufs.items()})
        for d in seq:
            o, t = d["optical"]["image"].cpu().pin_memory(), d["thermal"]["image"].cpu().pin_memory()
            pipe.run(o, t, d["optical"]["valid_mask"], d["thermal"]["valid_mask"])
            cur = pipe.download_async()
            if prev is not None:
                consume(*prev)
            prev = cur
        consume(*prev)
        torch.cuda.synchronize()
        pipe.verify()
    for g, ref in zip(got, refs):
        for i in range(B):
            no, nt, nm = int(g["counts"][i]), int(g["counts"][B + i]), int(g["match_count"][i])
            assert torch.equal(g["kp"][i, :no].long(), ref[i]["kp_optical"]) and torch.equal(g["kp"][B + i, :nt].long(), ref[i]["kp_thermal"])
            assert g["match_q"][i, :nm].tolist() == ref[i]["match_q"].tolist() and g["match_t"][i, :nm].tolist() == ref[i]["match_t"].tolist()


def test_streaming_depth_u8_upload_and_kernel_downloads(gpu_lib):
    """Round 5 streaming path: (1) xp_u8_to_unit_f32 = the loader's `astype(float32) / 255` bit for bit; (2) xp_copy_to_mapped_host copies any byte count into
    pinned host memory; (3) a PairPipeline fed 8-bit pinned images, with `depth` steps kept in flight by the host and the result lists downloaded by the
    kernel copy (depth + 1 pinned buffer sets), returns exactly what synchronous runs on the float images return."""
    import collections
    import ctypes
    from xpoint_amd import _lib as L
    from xpoint_amd.predict import PairPipeline
    lib = L.load()
    u8 = torch.randint(0, 256, (3, 1, 37, 53), dtype=torch.uint8)
    dev8 = u8.cuda(); out = torch.empty(u8.shape, dtype=torch.float32, device="cuda")
    L.check(lib.xp_u8_to_unit_f32(ctypes.c_void_p(dev8.data_ptr()), L.ptr(out), u8.numel(), L.current_stream()), "xp_u8_to_unit_f32")
    assert torch.equal(out.cpu(), torch.from_numpy(u8.numpy().astype(np.float32) / 255.0))
    for nbytes in (16, 4096, 100003):
        src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda"); dst = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
        L.check(lib.xp_copy_to_mapped_host(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), nbytes, L.current_stream()), "xp_copy_to_mapped_host")
        torch.cuda.synchronize()
        assert torch.equal(dst, src.cpu())
    H, W, B = 96, 128, 2
    net = _net(synth.xpoint_exp1_config(H, W))
    seq = [_data(s, B, H, W) for s in (3, 8, 1, 5, 2, 9)]
    q8 = lambda img: (img * 255.0).round().clamp(0, 255).to(torch.uint8)
    with torch.no_grad():
        ref_pipe = PairPipeline(net, B, H, W, cap=2048)
        refs = [ref_pipe.run(q8(d["optical"]["image"]).float() / 255.0, q8(d["thermal"]["image"]).float() / 255.0, d["optical"]["valid_mask"],
                             d["thermal"]["valid_mask"]).fetch() for d in seq]
        pipe = PairPipeline(net, B, H, W, cap=2048, overlap=True, alternate_encoders=3)
        assert pipe.depth == 3
        got, pend = [], collections.deque()
        for d in seq:
            o, t = q8(d["optical"]["image"]).cpu().pin_memory(), q8(d["thermal"]["image"]).cpu().pin_memory()
            pipe.run(o, t, d["optical"]["valid_mask"], d["thermal"]["valid_mask"])
            pend.append(pipe.download_async())
            assert pipe.host_sets == pipe.depth + 1
            if len(pend) >= pipe.depth:                      # the host consumes step i - depth + 1 only now: depth steps were in flight
                bufs, ev = pend.popleft(); ev.synchronize(); got.append({k: v.clone() for k, v in bufs.items()})
        while pend:
            bufs, ev = pend.popleft(); ev.synchronize(); got.append({k: v.clone() for k, v in bufs.items()})
        torch.cuda.synchronize()
        pipe.verify()
    assert len(got) == len(refs)
    for g, ref in zip(got, refs):
        for i in range(B):
            no, nt, nm = int(g["counts"][i]), int(g["counts"][B + i]), int(g["match_count"][i])
            assert torch.equal(g["kp"][i, :no].long(), ref[i]["kp_optical"]) and torch.equal(g["kp"][B + i, :nt].long(), ref[i]["kp_thermal"])
            assert g["match_q"][i, :nm].tolist() == ref[i]["match_q"].tolist() and g["match_t"][i, :nm].tolist() == ref[i]["match_t"].tolist()


def test_fp16_range_trip_is_localised_to_the_offending_launches(gpu_lib):
    """VERDICT r5 item 7 (the range guard's cliff).  A LayerNorm gain of 1e6 in front of stage 2's first MLP puts fc1's operand (and then fc2's) beyond the
    fp16 range.  Round 6: the host finds the offending dense launches by bisection over the per-launch override mask (xp_set_dense_override) and sends
    only THOSE to the split-bf16 planes — the weight set stays on "h2" (one warning naming the launches), results within 1e-4 of the all-x3 run, later
    calls silent; eager API, single-stream / overlapped / captured pipelines; a second model sharing nothing starts from mask 0.  And at the bench
    size the re-routed forward keeps >= 0.90 of the pure-h2 rate (measured 0.970; the old behaviour, the whole weight set on x3, is 0.757)."""
    import time
    from xpoint_amd.predict import PairPipeline
    H, W, B = 64, 96, 1
    cfg = synth.xpoint_exp1_config(H, W)
    data = _data(5, B, H, W)
    args = (data["optical"]["image"], data["thermal"]["image"], data["optical"]["valid_mask"], data["thermal"]["valid_mask"])
    sd = {k: v.clone() for k, v in synth.make_torch_state_dict(cfg).items()}
    sd["encoder.layers.2.blocks.0.norm2.weight"] *= 1.0e6
    fc1, fc2 = 1 + 5 * (2 * 2 + 0) + 3, 1 + 5 * (2 * 2 + 0) + 4            # launch numbers of stage 2 / block 0 fc1 and fc2 (include/xpoint_hip.h)
    # fc1's operand is LN(x) * 1e6, fc2's the GELU of that; the block then adds ~1e5 to the residual stream, which goes un-normalised into the downsample
    # convolution after stage 2 (launch 41 + 2): three offenders, found one by one
    want = (1 << fc1) | (1 << fc2) | (1 << (41 + 2))
    with torch.no_grad():
        ref_net = _net(cfg, sd); ref_net.gemm_mode = "x3"
        ro, rt, _ = ref_net(data)
        net = _net(cfg, sd)
        with pytest.warns(RuntimeWarning, match="now run on the split-bf16 planes"):
            o, t, _ = net(data)
        assert net.effective_gemm_mode() == "h2" and net._h2_mask == want, hex(net._h2_mask)
        assert net.engine_key() == f"h2+{net._h2_mask:x}"
        for a, b in ((o["prob"], ro["prob"]), (t["prob"], rt["prob"]), (o["desc"], ro["desc"]), (t["desc"], rt["desc"])):
            assert bool(torch.isfinite(a).all()) and float((a - b).abs().max()) < 1e-4
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)           # later calls: the mask is in place, nothing trips, nothing warns
            o2, _, _ = net(data)
        assert torch.equal(o2["prob"], o["prob"])
        net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True)      # a new weight set starts clean
        assert net._h2_mask == 0 and net.engine_key() == "h2"
        # pipelines: found at the synchronisation point, the latest call re-run with the mask, graphs re-captured with it
        eager = None
        for kw in (dict(), dict(overlap=True, alternate_encoders=True), dict(graph=True), dict(overlap=True, alternate_encoders=True, graph=True)):
            graph = kw.pop("graph", False)
            net = _net(cfg, sd)
            pipe = PairPipeline(net, B, H, W, cap=4096, **kw)
            if graph:
                with pytest.warns(RuntimeWarning, match="now run on the split-bf16 planes"):
                    step = pipe.capture(*args)
                step(*args)
                got = pipe.fetch()
            else:
                pipe.run(*args)
                with pytest.warns(RuntimeWarning, match="now run on the split-bf16 planes"):
                    got = pipe.fetch()
            assert net.effective_gemm_mode() == "h2" and net._h2_mask == want, hex(net._h2_mask)
            if eager is None:
                eager = got
            assert torch.equal(got[0]["kp_optical"], eager[0]["kp_optical"]) and got[0]["match_q"].tolist() == eager[0]["match_q"].tolist()
            with warnings.catch_warnings():
                warnings.simplefilter("error", RuntimeWarning)
                pipe.run(*args)
                again = pipe.fetch()
            assert not pipe.repaired and torch.equal(again[0]["kp_optical"], eager[0]["kp_optical"])
        # the override is a per-call setting of the library: left at 0 afterwards, and a clean model is unaffected
        from xpoint_amd import _lib
        assert int(_lib.load().xp_get_dense_override()) == 0
        clean = _net(cfg)
        c1, _, _ = clean(data)
        assert clean._h2_mask == 0
        # rate at the bench size: pure h2 vs the same weights with the three launches re-routed (mask set by hand: the timing needs no trip)
        Hb, Wb, Bb = 480, 640, 4
        cfgb = synth.xpoint_exp1_config(Hb, Wb)
        netb = _net(cfgb)
        img = _data(0, Bb, Hb, Wb)
        x = torch.cat([img["optical"]["image"], img["thermal"]["image"]])

        def rate():
            for _ in range(3):
                netb.forward_raw(x, check=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                netb.forward_raw(x, check=False)
            torch.cuda.synchronize()
            return 10 / (time.perf_counter() - t0)
        r0 = rate(); netb._h2_mask = want; r1 = rate(); netb._h2_mask = 0; r0b = rate()
        netb._h2_off = True; rx = rate(); netb._h2_off = False
        print(f"forward rate, {2 * Bb} images 480x640: h2 {r0:.1f} / {r0b:.1f}, three launches on x3 {r1:.1f} ({r1 / max(r0, r0b):.3f}), whole set on x3 {rx:.1f} ({rx / max(r0, r0b):.3f})")
        assert r1 >= 0.90 * min(r0, r0b), (r0, r1, r0b)          # measured 0.970; the margin is for a noisy box


def test_fp16_range_overflow_falls_back_to_x3(gpu_lib, monkeypatch):
    """The default dense engine splits f32 operands into two fp16 values: activations beyond 65504 overflow, and the heads' ReLU would turn a NaN
    encoder map into finite, wrong scores.  Every forward reports it through a device status word (xp_xpoint_forward_ex); the host then re-runs
    on the split-bf16 engine (no range limit), keeps it for that weight set and WARNS — the default never raises and never returns the
    overflowed results (VERDICT r2 weak 1 / ADVICE r2).  Provoked with a LayerNorm gain of 1e6 in front of a stage-2 GEMM.  Checked for the
    eager API (every call, not only the first), the single-stream pipeline, the overlapped pipeline and a captured (hipGraph) pipeline."""
    from xpoint_amd import models
    from xpoint_amd.predict import PairPipeline
    # Round 6: a trip is first LOCALISED (test above); the whole-weight-set fallback tested here is what remains when that fails — more offending
    # launches than MAX_LOCALISED.  Forced with MAX_LOCALISED = 0: the choreography below is the round-3 / round-4 one, unchanged.
    monkeypatch.setattr(models.XPoint, "MAX_LOCALISED", 0)
    H, W, B = 64, 96, 1
    cfg = synth.xpoint_exp1_config(H, W)
    data = _data(5, B, H, W)
    args = (data["optical"]["image"], data["thermal"]["image"], data["optical"]["valid_mask"], data["thermal"]["valid_mask"])
    sd = {k: v.clone() for k, v in synth.make_torch_state_dict(cfg).items()}
    sd["encoder.layers.2.blocks.0.norm2.weight"] *= 1.0e6
    with torch.no_grad():
        ref_net = _net(cfg, sd); ref_net.gemm_mode = "x3"
        ro, rt, _ = ref_net(data)
        ref = PairPipeline(ref_net, B, H, W, cap=4096).run(*args).fetch()
        assert bool(torch.isfinite(ro["encoder_output"]).all())      # fine on x3: the out-of-range value is an INTERMEDIATE (norm2's output, fc1's operand)
        # eager API
        net = _net(cfg, sd)
        assert net.gemm_mode == "h2" and net.effective_gemm_mode() == "h2"
        with pytest.warns(RuntimeWarning, match="re-running on gemm_mode 'x3'"):
            o, t, _ = net(data)
        assert net.effective_gemm_mode() == "x3" and net.gemm_mode == "h2"
        assert torch.equal(o["prob"], ro["prob"]) and torch.equal(t["desc"], rt["desc"])
        o2, _, _ = net(data)                                   # stays on x3: no second warning, same results
        assert torch.equal(o2["prob"], ro["prob"])
        net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True)      # a new weight set gets the default engine back
        assert net.effective_gemm_mode() == "h2"
        net(data)
        assert net.effective_gemm_mode() == "h2"
        # pipelines: the trip is found at the synchronisation point and the latest call is re-run
        for kw in (dict(), dict(overlap=True, alternate_encoders=True), dict(graph=True), dict(overlap=True, alternate_encoders=True, graph=True)):
            graph = kw.pop("graph", False)
            net = _net(cfg, sd)
            pipe = PairPipeline(net, B, H, W, cap=4096, **kw)
            if graph:
                with pytest.warns(RuntimeWarning, match="re-running on gemm_mode 'x3'"):
                    step = pipe.capture(*args)                  # trips during the warm-up: the graphs are captured on x3
                step(*args)
                got = pipe.fetch()
            else:
                pipe.run(*args)
                with pytest.warns(RuntimeWarning, match="re-running on gemm_mode 'x3'"):
                    got = pipe.fetch()
            assert net.effective_gemm_mode() == "x3"
            assert torch.equal(got[0]["kp_optical"], ref[0]["kp_optical"]) and torch.equal(got[0]["kp_thermal"], ref[0]["kp_thermal"])
            assert got[0]["match_q"].tolist() == ref[0]["match_q"].tolist() and got[0]["match_t"].tolist() == ref[0]["match_t"].tolist()
            pipe.run(*args)
            got = pipe.fetch()                                  # later steps: x3 from the start, nothing to repair
            assert torch.equal(got[0]["kp_optical"], ref[0]["kp_optical"])
        # a graph captured on h2 that trips LATER (other inputs) is re-captured on x3
        net = _net(cfg, sd)
        pipe = PairPipeline(net, B, H, W, cap=4096)
        zero = torch.zeros_like(args[0])
        step = pipe.capture(zero, zero, args[2], args[3])          # black images: in range?  (either way the result below must be the x3 one)
        step(*args)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            got = pipe.fetch()
        assert net.effective_gemm_mode() == "x3"
        assert torch.equal(got[0]["kp_optical"], ref[0]["kp_optical"]) and got[0]["match_q"].tolist() == ref[0]["match_q"].tolist()
        # every pipeline owns its status word: a trip raised by a step in flight on pipeline A is not consumed by pipeline B's verify() on the
        # same model, nor by reading / clearing the model's shared word (what an eager forward's check does); A still finds and repairs it
        net = _net(cfg, sd)
        pipe_a = PairPipeline(net, B, H, W, cap=4096)
        pipe_b = PairPipeline(net, B, H, W, cap=4096)
        assert pipe_a.status_word().data_ptr() != pipe_b.status_word().data_ptr() != net.status_word("cuda").data_ptr()
        pipe_a.run(*args)                                       # trips (h2), nobody has looked yet
        torch.cuda.synchronize()
        assert int(pipe_a.status_word().item()) != 0 and int(pipe_b.status_word().item()) == 0 and int(net.status_word("cuda").item()) == 0
        assert pipe_b._settle_engine() is False                 # B's own check (the first thing its verify() does): nothing to settle, and A's bits stay where they are
        net.status_word("cuda").zero_()
        assert int(pipe_a.status_word().item()) != 0 and net.effective_gemm_mode() == "h2"
        with pytest.warns(RuntimeWarning, match="re-running on gemm_mode 'x3'"):
            got = pipe_a.fetch()
        assert net.effective_gemm_mode() == "x3" and int(pipe_a.status_word().item()) == 0
        assert torch.equal(got[0]["kp_optical"], ref[0]["kp_optical"]) and got[0]["match_q"].tolist() == ref[0]["match_q"].tolist()
        # a trip judged against the engine its forwards were ENQUEUED with: B's step ran on h2 and tripped; before B looks, A's trip has already
        # switched the shared model to x3 — B must still repair (re-run on x3, no second warning), not raise "non-finite on x3"; eager and captured
        net = _net(cfg, sd)
        pipe_a = PairPipeline(net, B, H, W, cap=4096)
        pipe_b = PairPipeline(net, B, H, W, cap=4096)
        pipe_b.run(*args)
        pipe_a.run(*args)
        torch.cuda.synchronize()
        assert int(pipe_a.status_word().item()) != 0 and int(pipe_b.status_word().item()) != 0
        with pytest.warns(RuntimeWarning, match="re-running on gemm_mode 'x3'"):
            pipe_a.fetch()
        assert net.effective_gemm_mode() == "x3" and int(pipe_b.status_word().item()) != 0
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)      # the model is on x3 already: B repairs silently
            got = pipe_b.fetch()
        assert pipe_b.repaired and int(pipe_b.status_word().item()) == 0
        assert torch.equal(got[0]["kp_optical"], ref[0]["kp_optical"]) and got[0]["match_q"].tolist() == ref[0]["match_q"].tolist()
        pipe_b.run(*args)
        got = pipe_b.fetch()                                    # later steps: x3 from the start, nothing to repair
        assert not pipe_b.repaired and torch.equal(got[0]["kp_optical"], ref[0]["kp_optical"])
        # (captured pipelines: every replay is recorded with the engine its graphs were captured on)
        step = pipe_b.capture(*args)
        assert pipe_b._graph_engine == "x3"
        step(*args)
        assert pipe_b._engine_enqueued == "x3" and not pipe_b._h2_pending
        # genuinely non-finite weights: no engine can help -> raise, on every engine
        sd_nan = {k: v.clone() for k, v in synth.make_torch_state_dict(cfg).items()}
        sd_nan["encoder.layers.1.blocks.0.mlp.fc1.bias"][3] = float("nan")
        for mode in ("h2", "x3", "f32"):
            net = _net(cfg, sd_nan); net.gemm_mode = mode
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                with pytest.raises(RuntimeError, match="non-finite"):
                    net(data)


@pytest.mark.parametrize("gemm_mode", ["h2", "x3", "f32"])
def test_trained_like_weights_vs_reference(gpu_lib, golden, gemm_mode, capsys):
    """G19: weights with trained-like statistics — LayerNorm gains log-uniform 0.0625..16, 1 % outlier channels x100 in the
    patch-embed / downsample convolutions and the residual writers, dt bias at both ends of its range; the residual stream reaches ~1.5e3 and goes
    un-normalised into the downsample convolutions and the heads (VMamba.py:1405-1440,1500-1505) — on a plain pair and on a pair at the contrast
    extremes, 224x320, against the REAL reference: prob / desc within 1e-4 on all three f32-grade back ends (the default one without tripping its
    range guard), keypoints and mutual-NN pairs identical up to attributed near-ties."""
    from tests import parity
    from xpoint_amd import utils
    from xpoint_amd.predict import predict_align_image_pair
    g = golden("g19_trained_like.npz")
    _, H, W = [int(v) for v in g["meta"]]
    cfg = synth.xpoint_exp1_config(H, W)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_trained_like_state_dict(cfg).items()}
    net = _net(cfg, sd)
    net.gemm_mode = gemm_mode
    lines = []
    for c, dnp in enumerate((synth.make_pair_batch(0, 1, H, W), synth.make_contrast_pair(1, H, W))):
        data = synth.to_torch(dnp, "cuda")
        with torch.no_grad():
            raw_o, raw_t, _ = net(data)
            raw = {"optical": {k: (v.clone() if torch.is_tensor(v) else v) for k, v in raw_o.items()},
                   "thermal": {k: (v.clone() if torch.is_tensor(v) else v) for k, v in raw_t.items()}}
            _, _, res = predict_align_image_pair(net, data)
        assert net.effective_gemm_mode() == gemm_mode           # in range: the guard did not trip
        for spec in ("optical", "thermal"):
            r = raw[spec]
            assert float(r["encoder_output"].abs().max()) > 500.0          # the fixture's point: a large raw residual stream
            e_p = float(np.abs(r["prob"].cpu().numpy() - g[f"c{c}/{spec}/prob"]).max())
            d = r["desc"].cpu().numpy()
            e_d = float(np.abs((d if spec == "optical" else d[:, :, ::2, ::2]) - g[f"c{c}/{spec}/desc"]).max())
            lines.append(f"g19 case {c} {spec} [{gemm_mode}]: prob err {e_p:.2e}, desc err {e_d:.2e}, enc absmax {float(r['encoder_output'].abs().max()):.0f}")
            assert e_p < TOL and e_d < TOL, lines[-1]
        kp_m = {"optical": res[0]["kp_optical"].cpu().numpy(), "thermal": res[0]["kp_thermal"].cpu().numpy()}
        for spec in ("optical", "thermal"):
            rep, bad = parity.explain_keypoint_diff(kp_m[spec], g[f"c{c}/kp_{spec}"].astype(np.int64), raw[spec]["prob"][0, 0].cpu().numpy(), 0.015, 8, tol=TOL)
            if rep:
                lines.append(parity.format_report(f"g19 case {c} {spec} keypoints", rep))
            assert not bad, parity.format_report(f"g19 case {c} {spec}: UNEXPLAINED", bad)
        ms = res[0]["matches"]
        mine = np.array([[m.queryIdx, m.trainIdx] for m in ms], dtype=np.int64).reshape(-1, 2)
        vol = {"optical": raw["optical"]["desc_nhwc"][0], "thermal": raw["thermal"]["desc_nhwc"][0]}
        desc_of = lambda side, pts: utils.interpolate_descriptors_nhwc(torch.from_numpy(pts).cuda(), vol[side], H, W).cpu().numpy()
        rep, bad = parity.explain_match_diff(kp_m["optical"], kp_m["thermal"], g[f"c{c}/kp_optical"], g[f"c{c}/kp_thermal"], mine,
                                             g[f"c{c}/matches"].astype(np.int64), desc_of, tol=TOL)
        if rep:
            lines.append(parity.format_report(f"g19 case {c} mutual-NN pairs", rep))
        assert not bad, parity.format_report(f"g19 case {c}: UNEXPLAINED match differences", bad)
        lines.append(f"g19 case {c} [{gemm_mode}]: {len(kp_m['optical'])}/{len(kp_m['thermal'])} keypoints, {len(mine)} pairs vs reference {len(g[f'c{c}/matches'])}")
    with capsys.disabled():
        print("\n" + "\n".join(lines))


def test_pipeline_heals_nms_non_convergence(gpu_lib):
    """Images too large for a one-band finisher (1024 x 1024: BASELINE config C4) repeat the banded finisher a fixed number of times in the stream-
    ordered mode; an image with a suppression chain across more band borders than that must not fail the step: verify() raises the count (kept),
    recomputes the latest call's post-processing and the results equal those of a pipeline that had enough launches from the start — eager and captured.
    (Whether one launch suffices depends on the heat map: the test requires equal results either way and, if the repair ran, that the count was raised.)"""
    import warnings
    from xpoint_amd.predict import PairPipeline
    H, W, B = 1024, 1024, 1
    net = _net(synth.xpoint_exp1_config(H, W))
    d = _data(2, B, H, W)
    args = (d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"])
    with torch.no_grad():
        ref = PairPipeline(net, B, H, W, cap=32768, nms_sweeps=16).run(*args).fetch()
        for graph in (False, True):
            pipe = PairPipeline(net, B, H, W, cap=32768, nms_sweeps=1)
            step = pipe.capture(*args) if graph else pipe.run
            step(*args)
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                got = pipe.fetch()
            repaired = any("NMS needed more than 1 sweeps" in str(w.message) for w in rec)
            assert repaired == (pipe.sweeps > 1)
            for i in range(B):
                assert torch.equal(got[i]["kp_optical"], ref[i]["kp_optical"]) and torch.equal(got[i]["kp_thermal"], ref[i]["kp_thermal"])
                assert got[i]["match_q"].tolist() == ref[i]["match_q"].tolist()
            step(*args)
            got = pipe.fetch()                 # a raised count is kept: converges without a second repair
            assert torch.equal(got[0]["kp_optical"], ref[0]["kp_optical"])


def _allclose_report(got, ref, rtol, atol):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    return float(d.max()), float(np.sqrt((d * d).mean())), float((d <= atol + rtol * np.abs(ref)).mean())


@pytest.mark.parametrize("cls", ["amp16", "amp16f"])
@pytest.mark.parametrize("tag,H,W", [("64x96", 64, 96), ("224x320", 224, 320), ("480x640", 480, 640)])
def test_mixed_precision_class_vs_reference_g20(gpu_lib, golden, tag, H, W, cls, capsys):
    """SURVEY.md 8(f) rank 3 — the reference's `mixed_precision: true` deployment class (XPoint.py:182: autocast around the forward), pinned:
    g20 = the REAL reference under float16 CPU autocast (the harness points torch.cuda.amp.autocast at torch.autocast("cpu", float16); half is
    autocast's default dtype).  gemm_mode "amp16" (xp_set_amp_mode) rounds every inter-op activation to fp16 where autocast ends in a half tensor and
    feeds fp16-rounded conv / linear weights to single exact fp16 x fp16 products, scan / out_norm / softmax / normalize in f32.
    cls "amp16f" = the same recipe with HALF STORAGE (xp_xpoint_forward_f16: fp16 tensors in HBM, one-product fp16 MFMA GEMMs by LDS-DMA) — the fast
    deployment class, held to the same bars.
    The fp16 recipe is chaotic at the output level — the CPU restatement of the very same recipe (oracle AMP16, bit-equal to the reference op by
    op) ends 4e-3 .. 6e-3 from g20 in `prob`, exactly as far as the f32 forward is — so the per-op pin is test_mixed_precision_ops_vs_reference_taps
    and this test bounds the end-to-end NOISE: (1) >= 99.9 % of the elements of every output within the reference's own fp16 tolerances rtol 3e-3 /
    atol 5e-3 (test_selective_scan.py:401-403) — relaxed to "no worse than the f32 class's own fraction minus 0.2 %" where the f32 class itself is below
    99.9 %, floor 99.5 % — and none beyond 5 atol; (2) rms distance to g20 no larger than 1.6 x the rms distance between g20 and the
    f32 class (the recipe's own noise level); reported: keypoint and match agreement with the reference's mixed-precision lists.
    """
    from xpoint_amd.predict import predict_align_image_pair
    g = golden("g20_mixed_precision_fp16.npz")
    RTOL, ATOL = 3e-3, 5e-3
    cfg = synth.xpoint_exp1_config(H, W)
    net = _net(cfg)
    data = _data(0, 1, H, W)
    outs = {}
    for mode in (cls, "h2"):
        net.gemm_mode = mode
        with torch.no_grad():
            o, t, _ = net(data)
        outs[mode] = {"optical": {k: o[k].cpu().numpy() for k in ("prob", "desc", "encoder_output")},
                      "thermal": {k: t[k].cpu().numpy() for k in ("prob", "desc", "encoder_output")}}
        assert net.effective_gemm_mode() == mode
    lines = []
    for spec in ("optical", "thermal"):
        cmp = []
        if tag == "64x96":
            cmp = [(k, (lambda a: a), g[f"{tag}/{spec}/{k}"]) for k in ("prob", "desc", "encoder_output")]
        elif tag == "224x320":
            cmp = [("prob", (lambda a: a), g[f"{tag}/{spec}/prob"]), ("desc", (lambda a: a[:, :, ::2, ::2]), g[f"{tag}/{spec}/desc"])]
        else:
            cmp = [("prob", (lambda a: a[0, 0, ::16]), g[f"{tag}/{spec}/prob_rows"]), ("desc", (lambda a: a[0][:, ::6, ::8]), g[f"{tag}/{spec}/desc_cols"])]
        for k, view, ref in cmp:
            e_amp, rms_amp, frac = _allclose_report(view(outs[cls][spec][k]), ref, RTOL, ATOL)
            e_f32, rms_f32, frac_f32 = _allclose_report(view(outs["h2"][spec][k]), ref, RTOL, ATOL)
            lines.append(f"g20 {tag} {spec} {k}: {cls} vs reference-amp max {e_amp:.2e} rms {rms_amp:.2e}, {100 * frac:.3f} % within rtol 3e-3 / atol 5e-3 "
                         f"(f32 class vs reference-amp: max {e_f32:.2e} rms {rms_f32:.2e})")
            # (1): 99.9 % — or, where the recipe's own noise already puts the f32 class below that (the smallest size has a few thousand elements, and a
            # one-ulp change in any kernel moves a handful of them across the bound), within 0.2 % of the f32 class's own fraction; never below 99.5 %
            assert frac >= max(0.995, min(0.999, frac_f32 - 0.002)) and e_amp <= 5 * ATOL * max(1.0, float(np.abs(ref).max())), lines[-1] + f" (f32 class: {100 * frac_f32:.3f} %)"
            assert rms_amp <= 1.6 * rms_f32, lines[-1]
        # the encoder output is a half tensor in the reference: every value of the class's is fp16-representable too
        enc = torch.from_numpy(outs[cls][spec]["encoder_output"])
        assert torch.equal(enc, enc.to(torch.float16).to(torch.float32))
    net.gemm_mode = cls
    with torch.no_grad():
        _, _, res = predict_align_image_pair(net, data)
    kp = {"optical": {tuple(p) for p in res[0]["kp_optical"].cpu().numpy().tolist()}, "thermal": {tuple(p) for p in res[0]["kp_thermal"].cpu().numpy().tolist()}}
    rk = {s: {tuple(int(v) for v in p) for p in g[f"{tag}/kp_{s}"]} for s in ("optical", "thermal")}
    agree = {s: len(kp[s] & rk[s]) / max(1, len(kp[s] | rk[s])) for s in kp}
    ko, kt = res[0]["kp_optical"].cpu().numpy(), res[0]["kp_thermal"].cpu().numpy()
    mine = {(tuple(ko[m.queryIdx]), tuple(kt[m.trainIdx])) for m in res[0]["matches"]}
    gko, gkt = g[f"{tag}/kp_optical"], g[f"{tag}/kp_thermal"]
    ref_m = {(tuple(int(v) for v in gko[q]), tuple(int(v) for v in gkt[t])) for q, t in g[f"{tag}/matches"]}
    m_agree = len(mine & ref_m) / max(1, len(mine | ref_m))
    lines.append(f"g20 {tag}: keypoint agreement (IoU of the lists) optical {agree['optical']:.4f} thermal {agree['thermal']:.4f}; mutual-NN pair agreement {m_agree:.4f} "
                 f"({len(mine)} pairs vs {len(ref_m)})")
    with capsys.disabled():
        print("\n" + "\n".join(lines))
    assert min(agree.values()) > 0.9 and m_agree > 0.7


def test_mixed_precision_ops_vs_reference_taps(gpu_lib, golden, capsys):
    """The fp16 class is chaotic at the output level (two faithful implementations of the recipe end ~4e-3 apart — as far as f32 is from either), so
    the recipe is pinned OP BY OP: g20 holds the half tensors the REAL reference produced inside its first VSS block (and stem / downsample / head)
    under float16 autocast; every device kernel of the class (xp_set_amp_mode(1)) is fed the reference's INPUT tap and must reproduce the reference's
    OUTPUT tap: >= 99 % of the elements bit-equal, the rest within 2 fp16 ulps (summation order before the rounding)."""
    import ctypes
    from xpoint_amd import _lib as L
    g = golden("g20_mixed_precision_fp16.npz")
    tp = lambda k: torch.from_numpy(g[f"64x96/tap/{k}"].astype(np.float32)).cuda()
    H, W = 64, 96
    cfg = synth.xpoint_exp1_config(H, W)
    sd = synth.make_torch_state_dict(cfg)
    r16 = lambda t: t.to(torch.float16).to(torch.float32)
    st = L.current_stream()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    lines = []
    keep = []                       # device temporaries must outlive the (asynchronous) launches that read them

    def dev(t):
        t = t.contiguous().cuda()
        keep.append(t)
        return t

    def h2w(w2d):
        w2d = dev(r16(w2d))
        buf = torch.empty(L.load().xp_split_weights_h2_bytes(*w2d.shape), dtype=torch.uint8, device="cuda"); keep.append(buf)
        L.call("xp_split_weights_h2", L.ptr(w2d), vp(buf), w2d.shape[0], w2d.shape[1], st)
        return buf

    def check(name, mine, ref, frac=0.99, ulps=2.0):
        ref = ref.float()
        eq = float((mine == ref).float().mean())
        ulp = torch.clamp(torch.abs(ref), min=float(ref.pow(2).mean().sqrt())) * 2.0 ** -10          # fp16 spacing at the value, floored at the tensor's rms
        worst = float(((mine - ref).abs() / ulp).max())
        lines.append(f"{name:34s} bit-equal {eq:.5f}, worst {worst:.2f} fp16 ulp")
        assert eq >= frac and worst <= ulps, lines[-1]

    def gemm(A2d, wbuf, N, K, bias=None, res=None, act=0):
        M = A2d.shape[0]
        C = torch.empty((M, N), device="cuda")
        b = dev(r16(bias)) if bias is not None else None
        L.call("xp_gemm_nt_h2", L.ptr(A2d), vp(wbuf), L.ptr(C), L.ptr(b), None, None, L.ptr(res), M, N, K, K, N, N if res is not None else 0, act, st)
        return C

    def ln(x2d, w, b):
        x2d = dev(x2d); y = torch.empty_like(x2d)
        L.call("xp_layernorm", L.ptr(x2d), L.ptr(y), L.ptr(dev(w)), L.ptr(dev(b)), x2d.shape[0], x2d.shape[1], 1e-5, 0, st)
        return y
    p = "encoder.layers.0.blocks.0."
    L.call("xp_set_amp_mode", 1)
    try:
        X0 = tp("b0/in").contiguous()                                   # (1, 16, 24, 96) NHWC half values
        Hs, Ws, C = X0.shape[1:]
        M = Hs * Ws
        check("norm (LayerNorm)", ln(tp("b0.norm/in").view(M, C), sd[p + "norm.weight"], sd[p + "norm.bias"]), tp("b0.norm/out").view(M, C))
        check("in_proj", gemm(tp("b0.in_proj/in").view(M, C).contiguous(), h2w(sd[p + "op.in_proj.weight"]), C, C), tp("b0.in_proj/out").view(M, C))
        xin = tp("b0.conv2d/in").permute(0, 2, 3, 1).contiguous()       # the reference runs the depthwise conv in NCHW
        y = torch.empty_like(xin)
        L.call("xp_dwconv3x3_silu", L.ptr(xin), L.ptr(dev(r16(sd[p + "op.conv2d.weight"]).reshape(C, 9).t())), L.ptr(y), 1, Hs, Ws, C, st)
        check("conv2d + SiLU", y, tp("b0.act/out").permute(0, 2, 3, 1).contiguous())
        # SS2D core: x_proj (half conv1d) -> dt projection (half, rounded before the f32 bias) -> f32 scan, merge, out_norm
        order = [0, 2, 1, 3]
        R = sd[p + "op.dt_projs_weight"].shape[2]
        XW = 4 * (R + 2)
        xd = gemm(y.view(M, C), h2w(sd[p + "op.x_proj_weight"][order].reshape(XW, C)), XW, C)
        out = torch.empty((M, C), device="cuda")
        ws = torch.empty(L.load().xp_ss2d_core_workspace_bytes(1, Hs, Ws, C) // 4 + 16, device="cuda")
        A = dev((-torch.exp(sd[p + "op.A_logs"].float())).view(4, C)[order])
        L.call("xp_ss2d_core_fwd", L.ptr(y), L.ptr(xd), L.ptr(dev(r16(sd[p + "op.dt_projs_weight"])[order].permute(0, 2, 1))),
               L.ptr(dev(sd[p + "op.dt_projs_bias"][order])), L.ptr(A), L.ptr(dev(sd[p + "op.Ds"].view(4, C)[order])),
               L.ptr(dev(sd[p + "op.out_norm.weight"])), L.ptr(dev(sd[p + "op.out_norm.bias"])), L.ptr(out), L.ptr(ws), ws.numel() * 4, 1, Hs, Ws, C, R, 1,
               1e-5, st)
        ref_on = tp("b0.out_norm/out").view(M, C)                       # f32 in the reference (out_norm runs on the scan's f32 output)
        e = float((out - ref_on).abs().max())
        lines.append(f"{'SS2D core -> out_norm (f32)':34s} max |err| {e:.2e} (values up to {float(ref_on.abs().max()):.1f})")
        assert e < 2e-4 * max(1.0, float(ref_on.abs().max())), lines[-1]
        L.call("xp_round_f16", L.ptr(out), L.ptr(out), M * C, st)
        check("SS2D core output .to(half)", out, r16(ref_on), frac=0.98)
        # out_proj + first residual (norm2's input is the block's x after it)
        x1 = gemm(dev(r16(ref_on)), h2w(sd[p + "op.out_proj.weight"]), C, C, res=X0.view(M, C).contiguous())
        check("out_proj + residual", x1, tp("b0.norm2/in").view(M, C), frac=0.98)
        check("norm2", ln(tp("b0.norm2/in").view(M, C).contiguous(), sd[p + "norm2.weight"], sd[p + "norm2.bias"]), tp("b0.norm2/out").view(M, C))
        h = gemm(tp("b0.fc1/in").view(M, C).contiguous(), h2w(sd[p + "mlp.fc1.weight"]), 4 * C, C, bias=sd[p + "mlp.fc1.bias"], act=1)
        check("fc1 + GELU", h, tp("b0.mlp_act/out").view(M, 4 * C))
        x2 = gemm(tp("b0.fc2/in").view(M, 4 * C).contiguous(), h2w(sd[p + "mlp.fc2.weight"]), C, 4 * C, bias=sd[p + "mlp.fc2.bias"], res=tp("b0.norm2/in").view(M, C).contiguous())
        check("fc2 + residual (block output)", x2, tp("b0/out").view(M, C), frac=0.98)
        # stem: image -> conv + LN + GELU (first three stages of patch_embed) is inside xp_stem_conv_ln_gelu; whole patch_embed = stem + conv + LN
        img = synth.to_torch(synth.make_pair_batch(0, 1, H, W), "cuda")["optical"]["image"]
        q = "encoder.patch_embed."
        E = C
        w0 = dev(r16(sd[q + "0.weight"]).double().sum(dim=1).permute(1, 2, 0).reshape(9, -1).float())
        s1 = torch.empty((1, H // 2, W // 2, E // 2), device="cuda")
        L.call("xp_stem_conv_ln_gelu", L.ptr(img), L.ptr(w0), L.ptr(dev(r16(sd[q + "0.bias"]))), L.ptr(dev(sd[q + "2.weight"])), L.ptr(dev(sd[q + "2.bias"])),
               L.ptr(s1), 1, H, W, E // 2, 1e-5, st)
        w5 = r16(sd[q + "5.weight"]).permute(0, 2, 3, 1).reshape(E, -1).contiguous()
        c2 = torch.empty((1, H // 4, W // 4, E), device="cuda")
        L.call("xp_conv3x3_nhwc_h2", L.ptr(s1), vp(h2w(w5)), L.ptr(c2), L.ptr(dev(r16(sd[q + "5.bias"]))), None, None, 1, H // 2, W // 2, E // 2, E, 2, 0, 0, st)
        check("patch_embed (stem, conv, LN)", ln(c2.view(-1, E), sd[q + "7.weight"], sd[q + "7.bias"]), tp("patch_embed/out").view(-1, E), frac=0.97, ulps=3.0)
        # head: reflection pad + conv + ReLU + BatchNorm (one launch: epilogue act 2 + affine)
        enc = tp("head_det.1/in")                                       # (1, 48, 10, 14): already reflection-padded, f32 container of half values
        d = "detector_head_convolutions."
        sc = sd[d + "3.weight"].double() / torch.sqrt(sd[d + "3.running_var"].double() + 1e-5)
        sh = sd[d + "3.bias"].double() - sd[d + "3.running_mean"].double() * sc
        inner = enc[:, :, 1:-1, 1:-1].permute(0, 2, 3, 1).contiguous()
        hb = torch.empty((1, inner.shape[1], inner.shape[2], 256), device="cuda")
        L.call("xp_conv3x3_nhwc_h2", L.ptr(inner), vp(h2w(r16(sd[d + "1.weight"]).permute(0, 2, 3, 1).reshape(256, -1).contiguous())), L.ptr(hb),
               L.ptr(dev(r16(sd[d + "1.bias"]))), L.ptr(dev(sc.float())), L.ptr(dev(sh.float())), 1, inner.shape[1], inner.shape[2], 48, 256, 1, 1, 2, st)
        check("head conv + ReLU + BatchNorm", hb, tp("head_det.3/out").permute(0, 2, 3, 1).contiguous(), frac=0.97, ulps=3.0)
    finally:
        L.call("xp_set_amp_mode", 0)
    with capsys.disabled():
        print("\n" + "\n".join(lines))


def test_mixed_precision_ops_vs_reference_taps_f16(gpu_lib, golden, capsys):
    """The op-by-op pin of test_mixed_precision_ops_vs_reference_taps for the HALF-STORAGE kernels of the fast class (gemm_mode "amp16f": csrc/gemm_f16.hip,
    elementwise_f16.hip, the fp16 instances of ss2d.hip): every kernel is fed the reference's input tap AS fp16 and must reproduce the reference's output tap
    (g20: the real reference under float16 autocast) — >= 99 % of the elements bit-equal, the rest within 2 fp16 ulps."""
    import ctypes
    from xpoint_amd import _lib as L
    g = golden("g20_mixed_precision_fp16.npz")
    tp = lambda k: torch.from_numpy(g[f"64x96/tap/{k}"].astype(np.float32)).cuda()
    H, W = 64, 96
    cfg = synth.xpoint_exp1_config(H, W)
    sd = synth.make_torch_state_dict(cfg)
    st = L.current_stream()
    lines, keep = [], []

    def dev(t, half=False):
        t = t.contiguous().cuda()
        if half:
            t = t.half()
        keep.append(t)
        return t

    def check(name, mine, ref, frac=0.99, ulps=2.0):
        mine, ref = mine.float(), ref.float()
        eq = float((mine == ref).float().mean())
        ulp = torch.clamp(torch.abs(ref), min=float(ref.pow(2).mean().sqrt())) * 2.0 ** -10
        worst = float(((mine - ref).abs() / ulp).max())
        lines.append(f"[f16 storage] {name:34s} bit-equal {eq:.5f}, worst {worst:.2f} fp16 ulp")
        assert eq >= frac and worst <= ulps, lines[-1]

    def gemm(A2d, w2d, bias=None, res=None, act=0, c_f32=0):
        A, Wt = dev(A2d, True), dev(w2d, True)
        M, K = A.shape; N = Wt.shape[0]
        C = torch.empty((M, N), device="cuda", dtype=torch.float32 if c_f32 else torch.float16)
        b = dev(bias.half().float()) if bias is not None else None
        R = dev(res, True) if res is not None else None
        L.call("xp_gemm_nt_f16", L.ptr(A), L.ptr(Wt), L.ptr(C), c_f32, L.ptr(b), None, None, L.ptr(R), M, N, K, K, N, N if res is not None else 0, act, st)
        return C

    def ln(x2d, w, b):
        x = dev(x2d, True); y = torch.empty_like(x)
        L.call("xp_layernorm_f16", L.ptr(x), L.ptr(y), L.ptr(dev(w)), L.ptr(dev(b)), x.shape[0], x.shape[1], 1e-5, st)
        return y
    r16 = lambda t: t.to(torch.float16).to(torch.float32)
    p = "encoder.layers.0.blocks.0."
    X0 = tp("b0/in").contiguous()
    Hs, Ws, C = X0.shape[1:]
    M = Hs * Ws
    check("norm (LayerNorm)", ln(tp("b0.norm/in").view(M, C), sd[p + "norm.weight"], sd[p + "norm.bias"]), tp("b0.norm/out").view(M, C))
    check("in_proj", gemm(tp("b0.in_proj/in").view(M, C), sd[p + "op.in_proj.weight"]), tp("b0.in_proj/out").view(M, C))
    xin = dev(tp("b0.conv2d/in").permute(0, 2, 3, 1), True)
    y = torch.empty_like(xin)
    L.call("xp_dwconv3x3_silu_f16", L.ptr(xin), L.ptr(dev(r16(sd[p + "op.conv2d.weight"]).reshape(C, 9).t())), L.ptr(y), None, 1, Hs, Ws, C, st)
    check("conv2d + SiLU", y, tp("b0.act/out").permute(0, 2, 3, 1).contiguous())
    order = [0, 2, 1, 3]
    R = sd[p + "op.dt_projs_weight"].shape[2]
    XW = 4 * (R + 2)
    xd = gemm(y.view(M, C).float(), sd[p + "op.x_proj_weight"][order].reshape(XW, C))
    out = torch.empty((M, C), device="cuda", dtype=torch.float16)
    ws = torch.empty(L.load().xp_ss2d_core_workspace_bytes(1, Hs, Ws, C) // 4 + 16, device="cuda")
    A = dev((-torch.exp(sd[p + "op.A_logs"].float())).view(4, C)[order])
    L.call("xp_ss2d_core_fwd_f16", L.ptr(y), L.ptr(xd), None, None, L.ptr(dev(r16(sd[p + "op.dt_projs_weight"])[order].permute(0, 2, 1))),
           L.ptr(dev(sd[p + "op.dt_projs_bias"][order])), L.ptr(A), L.ptr(dev(sd[p + "op.Ds"].view(4, C)[order])),
           L.ptr(dev(sd[p + "op.out_norm.weight"])), L.ptr(dev(sd[p + "op.out_norm.bias"])), L.ptr(out), L.ptr(ws), ws.numel() * 4, 1, Hs, Ws, C, R, 1, 1e-5, st)
    ref_on = tp("b0.out_norm/out").view(M, C)
    check("SS2D core output .to(half)", out, r16(ref_on), frac=0.98)
    check("out_proj + residual", gemm(r16(ref_on), sd[p + "op.out_proj.weight"], res=X0.view(M, C)), tp("b0.norm2/in").view(M, C), frac=0.98)
    check("norm2", ln(tp("b0.norm2/in").view(M, C), sd[p + "norm2.weight"], sd[p + "norm2.bias"]), tp("b0.norm2/out").view(M, C))
    check("fc1 + GELU", gemm(tp("b0.fc1/in").view(M, C), sd[p + "mlp.fc1.weight"], bias=sd[p + "mlp.fc1.bias"], act=1), tp("b0.mlp_act/out").view(M, 4 * C))
    check("fc2 + residual (block output)", gemm(tp("b0.fc2/in").view(M, 4 * C), sd[p + "mlp.fc2.weight"], bias=sd[p + "mlp.fc2.bias"], res=tp("b0.norm2/in").view(M, C)),
          tp("b0/out").view(M, C), frac=0.98)
    img = synth.to_torch(synth.make_pair_batch(0, 1, H, W), "cuda")["optical"]["image"]
    q = "encoder.patch_embed."
    E = C
    w0 = dev(r16(sd[q + "0.weight"]).double().sum(dim=1).permute(1, 2, 0).reshape(9, -1).float())
    s1 = torch.empty((1, H // 2, W // 2, E // 2), device="cuda", dtype=torch.float16)
    L.call("xp_stem_conv_ln_gelu_f16", L.ptr(img), L.ptr(w0), L.ptr(dev(r16(sd[q + "0.bias"]))), L.ptr(dev(sd[q + "2.weight"])), L.ptr(dev(sd[q + "2.bias"])),
           L.ptr(s1), 1, H, W, E // 2, 1e-5, st)
    w5 = dev(sd[q + "5.weight"].permute(0, 2, 3, 1).reshape(E, -1), True)
    c2 = torch.empty((1, H // 4, W // 4, E), device="cuda", dtype=torch.float16)
    L.call("xp_conv3x3_nhwc_f16", L.ptr(s1), L.ptr(w5), L.ptr(c2), 0, L.ptr(dev(r16(sd[q + "5.bias"]))), None, None, 1, H // 2, W // 2, E // 2, E, 2, 0, 0, st)
    check("patch_embed (stem, conv, LN)", ln(c2.view(-1, E).float(), sd[q + "7.weight"], sd[q + "7.bias"]), tp("patch_embed/out").view(-1, E), frac=0.97, ulps=3.0)
    enc = tp("head_det.1/in")
    d = "detector_head_convolutions."
    sc = sd[d + "3.weight"].double() / torch.sqrt(sd[d + "3.running_var"].double() + 1e-5)
    sh = sd[d + "3.bias"].double() - sd[d + "3.running_mean"].double() * sc
    inner = dev(enc[:, :, 1:-1, 1:-1].permute(0, 2, 3, 1), True)
    hb = torch.empty((1, inner.shape[1], inner.shape[2], 256), device="cuda", dtype=torch.float16)
    L.call("xp_conv3x3_nhwc_f16", L.ptr(inner), L.ptr(dev(sd[d + "1.weight"].permute(0, 2, 3, 1).reshape(256, -1), True)), L.ptr(hb), 0,
           L.ptr(dev(r16(sd[d + "1.bias"]))), L.ptr(dev(sc.float())), L.ptr(dev(sh.float())), 1, inner.shape[1], inner.shape[2], 48, 256, 1, 1, 2, st)
    check("head conv + ReLU + BatchNorm", hb, tp("head_det.3/out").permute(0, 2, 3, 1).contiguous(), frac=0.97, ulps=3.0)
    torch.cuda.synchronize()
    with capsys.disabled():
        print("\n" + "\n".join(lines))


def test_fast_mixed_precision_class_tracks_parity_class(gpu_lib, capsys):
    """gemm_mode "amp16f" (half storage, one-product GEMMs) computes the recipe of "amp16" (f32 containers, split kernels with zero low planes): identical
    rounding points, only the order of f32 accumulations differs — so the two classes must sit far closer to each other than either sits to the reference
    (the class is chaotic: one-ulp differences are amplified through eight blocks).  480 x 640, deep stages on the sequential scan form in the fast class."""
    H, W = 480, 640
    net = _net(synth.xpoint_exp1_config(H, W))
    data = _data(0, 1, H, W)
    outs = {}
    for mode in ("amp16", "amp16f", "h2"):
        net.gemm_mode = mode
        with torch.no_grad():
            o, t, _ = net(data)
        outs[mode] = {k: torch.cat([o[k], t[k]]).cpu() for k in ("prob", "desc", "encoder_output")}
    lines = []
    for k in ("prob", "desc", "encoder_output"):
        a, b, f = outs["amp16"][k], outs["amp16f"][k], outs["h2"][k]
        rms_ab = float((a - b).pow(2).mean().sqrt()); rms_af = float((a - f).pow(2).mean().sqrt())
        lines.append(f"{k}: rms(amp16f - amp16) {rms_ab:.2e}, rms(amp16 - f32 class) {rms_af:.2e}, max |amp16f - amp16| {float((a - b).abs().max()):.2e}")
        assert rms_ab <= 1.5 * rms_af, lines[-1]
    enc = outs["amp16f"]["encoder_output"]
    assert torch.equal(enc, enc.half().float())
    with capsys.disabled():
        print("\n" + "\n".join(lines))
