"""CPU: the near-tie attribution helpers (tests/parity.py) and the oracle against the reference's C2 end-to-end
fixture (tests/golden/g15_c2_batch8.npz: the real reference's keypoints and NNMatcher index pairs at 480x640)."""
import numpy as np
import torch

from oracle import xpoint_oracle as xo
from xpoint_amd import synth
from tests import parity


def _kp(prob, thr=0.3, size=8):
    p = torch.from_numpy(prob)[None, None]
    out = xo.box_nms(p, size, thr)
    return torch.nonzero((out[0, 0] > thr).float()).numpy()


def test_keypoint_attribution_explains_budget_noise_and_flags_real_errors():
    rng = np.random.default_rng(3)
    prob = rng.random((96, 128)).astype(np.float32)
    prob = (np.floor(prob * 4096) / 4096).astype(np.float32)          # many exact and near ties
    noisy = (prob + rng.uniform(-4e-5, 4e-5, prob.shape)).astype(np.float32)
    a, b = _kp(prob), _kp(noisy)
    rep, bad = parity.explain_keypoint_diff(b, a, noisy, 0.3, 8, tol=1e-4)
    assert len(rep) > 0, "the perturbation was meant to flip some decisions"
    assert not bad, parity.format_report("unexplained", bad)
    # a keypoint that is simply wrong (far from every tie) must NOT be explained
    smooth = np.zeros((64, 64), np.float32); smooth[10, 10] = 0.9; smooth[40, 40] = 0.8
    rep, bad = parity.explain_keypoint_diff(np.array([[10, 10]]), np.array([[10, 10], [40, 40]]), smooth, 0.3, 8)
    assert len(bad) == 1 and bad[0]["kp"] == (40, 40) and bad[0]["side"] == "reference only"
    assert parity.explain_keypoint_diff(a, a, prob, 0.3) == ([], [])


def test_match_attribution():
    rng = np.random.default_rng(5)
    n1, n2, D = 60, 70, 32
    kpo = np.stack([np.arange(n1), np.arange(n1)], 1); kpt = np.stack([np.arange(n2), 2 * np.arange(n2)], 1)
    d1 = rng.standard_normal((n1, D)); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    d2 = rng.standard_normal((n2, D)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    d2[7] = d1[3]; d2[9] = d1[3] + 1e-7 * rng.standard_normal(D)         # planted near-tie: query 3 -> 7 or 9
    table = {"optical": {tuple(p): d for p, d in zip(kpo.tolist(), d1)}, "thermal": {tuple(p): d for p, d in zip(kpt.tolist(), d2)}}
    desc_of = lambda side, pts: np.array([table[side][tuple(p)] for p in pts.tolist()])
    ms = np.array([[m.queryIdx, m.trainIdx] for m in xo.get_matches(d1.astype(np.float32), d2.astype(np.float32))])
    assert any(q == 3 for q, _ in ms)
    flipped = ms.copy(); i = int(np.nonzero(ms[:, 0] == 3)[0][0]); flipped[i, 1] = 16 - ms[i, 1]     # 7 <-> 9
    rep, bad = parity.explain_match_diff(kpo, kpt, kpo, kpt, flipped, ms, desc_of)
    assert len(rep) == 2 and not bad, parity.format_report("m", rep)
    wrong = ms.copy(); j = int(np.nonzero(ms[:, 0] != 3)[0][0]); wrong[j, 1] = (ms[j, 1] + 1) % n2            # a plain error
    rep, bad = parity.explain_match_diff(kpo, kpt, kpo, kpt, wrong, ms, desc_of)
    assert bad, "a wrong match with a clear distance gap must be flagged"
    # a keypoint present on one side only explains the matches it takes part in
    rep, bad = parity.explain_match_diff(kpo, kpt, kpo, kpt[:-1], ms, ms[ms[:, 1] != n2 - 1], desc_of)
    assert all("one side only" in r["why"] or "differs between" in r["why"] for r in rep) and not bad


def test_oracle_c2_indices_vs_reference_fixture(golden):
    """The CPU oracle reproduces the REAL reference's 480x640 keypoints and NNMatcher index pairs (pair 0 of the C2 batch);
    differences, if any, must be attributed (the oracle's forward equals the reference's bit for bit at one thread, but
    this test runs at the host's thread count)."""
    g = golden("g15_c2_batch8.npz")
    H, W = 480, 640
    cfg = synth.xpoint_exp1_config(H, W)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}
    data = synth.to_torch(synth.make_pair_batch(0, 1, H, W))
    with torch.no_grad():
        res, (o, t, _), _ = xo.predict_align_image_pair(data, sd)
    r = res[0]
    for spec, out in (("optical", o), ("thermal", t)):
        rep, bad = parity.explain_keypoint_diff(r[f"kp_{spec}"].numpy(), g[f"p0/kp_{spec}"], out["prob"][0, 0].numpy(), 0.015, 8, tol=1e-5)
        assert not bad, parity.format_report(spec, rep)
    ms = xo.nnmatcher(r["desc_optical"].numpy(), r["desc_thermal"].numpy(), threshold=10.0)
    mine = np.array([[m.queryIdx, m.trainIdx] for m in ms])
    if np.array_equal(r["kp_optical"].numpy(), g["p0/kp_optical"]) and np.array_equal(r["kp_thermal"].numpy(), g["p0/kp_thermal"]):
        same = np.array_equal(mine, g["p0/matches"])
    else:
        same = False
    if not same:
        do, dt = o["desc"][0], t["desc"][0]
        desc_of = lambda side, pts: xo.interpolate_descriptors(torch.from_numpy(pts), do if side == "optical" else dt, H, W).numpy()
        rep, bad = parity.explain_match_diff(r["kp_optical"].numpy(), r["kp_thermal"].numpy(), g["p0/kp_optical"], g["p0/kp_thermal"],
                                             mine, g["p0/matches"], desc_of, tol=1e-5)
        assert not bad, parity.format_report("matches", rep)


def test_homography_oracle_selfcheck():
    """oracle/homography_oracle.py (the CPU restatement of the shipped robust estimator; its GPU twin is compared with it in
    tests/test_gpu_evaluation.py): recovers a known model under 40 % outliers, is deterministic, distinct samples, no model below 4."""
    from oracle import homography_oracle as ho
    rng = np.random.default_rng(11)
    Ht = np.array([[1.05, 0.03, 12.0], [-0.02, 0.97, -7.0], [4e-5, -3e-5, 1.0]])
    src = np.stack([rng.uniform(0, 640, 300), rng.uniform(0, 480, 300)], 1)
    q = np.concatenate([src, np.ones((300, 1))], 1) @ Ht.T
    dst = q[:, :2] / q[:, 2:] + rng.normal(0, 0.3, (300, 2))
    out = rng.random(300) < 0.4
    dst[out] = np.stack([rng.uniform(0, 640, out.sum()), rng.uniform(0, 480, out.sum())], 1)
    H, mask, n, best = ho.find_homography(src.astype(np.float32), dst.astype(np.float32), 3.0, 1500, seed=3)
    H2, mask2, n2, best2 = ho.find_homography(src.astype(np.float32), dst.astype(np.float32), 3.0, 1500, seed=3)
    assert best == best2 and np.array_equal(H, H2) and n == n2
    corners = np.array([[0, 0, 1], [640, 0, 1], [0, 480, 1], [640, 480, 1]], float)
    pa = corners @ H.T; pb = corners @ Ht.T
    assert np.abs(pa[:, :2] / pa[:, 2:] - pb[:, :2] / pb[:, 2:]).max() < 1.0 and abs(H[2, 2] - 1) < 1e-12
    assert (mask.astype(bool) & ~out).sum() > 0.95 * (~out).sum() and n == int(mask.sum())
    idx = ho.sample4(3, 0, np.arange(200), 7)
    assert all(len(set(r)) == 4 for r in idx.tolist()) and idx.min() >= 0 and idx.max() < 7
    assert ho.find_homography(src[:3].astype(np.float32), dst[:3].astype(np.float32))[2] == 0


def test_gray_conversion_fixture(golden):
    """tests/golden/g17_gray.npz (8-bit R, G, B -> gray, OpenCV's RGB2Gray<uchar> fixed point restated in exact integers by the oracle):
    the oracle reproduces it, and so does the package's host path (xpoint_amd.datasets), which is a separate numpy implementation."""
    from xpoint_amd.datasets import rgb_to_gray_u8, gray_lut
    g = golden("g17_gray.npz")
    assert np.array_equal(xo.bgr2gray_u8(g["rgb"]), g["gray"])
    assert np.array_equal(rgb_to_gray_u8(g["rgb"][None])[0], g["gray"])
    assert np.array_equal(gray_lut()[g["gray"]], g["value"]) and np.array_equal(xo.gray_to_float(g["gray"]), g["value"])
    assert np.array_equal(g["gray"][:256], np.arange(256))                      # r = g = b -> the grey level itself


def test_oracle_vs_round3_reference_fixtures(golden):
    """The CPU oracle against the round-3 fixtures of the REAL reference: one pair of g18 (C3: pairs 8..63; pair 37 = rank 4's shard) end to end,
    the trained-like weight set g19 (both cases: prob / desc to 1e-5, the oracle equals the reference bit for bit at one thread) and the
    RegNet head on the C5 crops (g21, two pairs)."""
    import zlib
    g18, g19, g21 = golden("g18_c3_pairs8to63.npz"), golden("g19_trained_like.npz"), golden("g21_c5_hm.npz")
    # --- g18: header table consistent, one pair reproduced
    hdr = {int(r[0]): r for r in g18["header"]}
    assert sorted(hdr) == list(range(8, 64))
    for i in (8, 37, 63):
        assert int(hdr[i][6]) == zlib.crc32(np.ascontiguousarray(g18[f"p{i}/matches"]).astype("<i2").tobytes())
    H, W = 480, 640
    cfg = synth.xpoint_exp1_config(H, W)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}
    with torch.no_grad():
        res, (o, t, _), _ = xo.predict_align_image_pair(synth.to_torch(synth.make_pair_batch(37, 1, H, W)), sd)
    for spec, out in (("optical", o), ("thermal", t)):
        rep, bad = parity.explain_keypoint_diff(res[0][f"kp_{spec}"].numpy(), g18[f"p37/kp_{spec}"], out["prob"][0, 0].numpy(), 0.015, 8, tol=1e-5)
        assert not bad, parity.format_report(spec, rep)
        assert len(rep) <= 4
    # --- g19: trained-like statistics
    _, H, W = [int(v) for v in g19["meta"]]
    cfg = synth.xpoint_exp1_config(H, W)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_trained_like_state_dict(cfg).items()}
    ln = sd["encoder.layers.0.blocks.0.norm.weight"]
    assert float(ln.min()) < 0.1 and float(ln.max()) > 10.0                        # heavy-tailed gains
    for c, dnp in enumerate((synth.make_pair_batch(0, 1, H, W), synth.make_contrast_pair(1, H, W))):
        with torch.no_grad():
            res, (o, t, _), _ = xo.predict_align_image_pair(synth.to_torch(dnp), sd)
        for spec, out in (("optical", o), ("thermal", t)):
            assert float(out["encoder_output"].abs().max()) > 500.0
            assert abs(float(out["encoder_output"].abs().max()) - float(g19[f"c{c}/{spec}/enc_absmax"][0])) < 1e-2
            assert float(np.abs(out["prob"].numpy() - g19[f"c{c}/{spec}/prob"]).max()) < 2e-5
            d = out["desc"].numpy()
            assert float(np.abs((d if spec == "optical" else d[:, :, ::2, ::2]) - g19[f"c{c}/{spec}/desc"]).max()) < 2e-5
            rep, bad = parity.explain_keypoint_diff(res[0][f"kp_{spec}"].numpy(), g19[f"c{c}/kp_{spec}"], out["prob"][0, 0].numpy(), 0.015, 8, tol=2e-5)
            assert not bad, parity.format_report(f"g19 c{c} {spec}", rep)
    # --- g21: RegNet head on the 256x256 crops of the C5 pairs
    cfg = synth.xpoint_exp1_config(256, 256, hm_head=True)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}
    for i in (0, 5):
        data = synth.to_torch(synth.make_pair_batch(i, 1, 480, 640))
        for spec in ("optical", "thermal"):
            data[spec]["image"] = data[spec]["image"][:, :, :256, :256].contiguous()
        with torch.no_grad():
            _, _, hm = xo.xpoint_forward(data, sd, hm_head=True)
        assert float(np.abs(hm.numpy().reshape(-1) - g21["hm"][i]).max()) < 1e-5


def test_oracle_amp16_recipe_vs_reference_taps(golden):
    """The oracle's AMP16 switch (the mixed-precision recipe restated on f32 tensors: every autocast op = f32 arithmetic on half-rounded inputs, half-
    rounded output) against g20 = the REAL reference under float16 CPU autocast.  Op by op — each op fed the reference's input tap — the restatement is
    bit-equal to the reference in >= 99.9 % of the elements and within one fp16 ulp elsewhere; end to end the two sit as far apart as g20 sits from the
    f32 forward (the fp16 recipe is chaotic at the output level), which is what the GPU test's noise bounds are sized by."""
    import torch.nn.functional as F
    g = golden("g20_mixed_precision_fp16.npz")
    tp = lambda k: torch.from_numpy(g[f"64x96/tap/{k}"].astype(np.float32))
    H, W = 64, 96
    cfg = synth.xpoint_exp1_config(H, W)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}
    p = "encoder.layers.0.blocks.0."

    def check(name, mine, ref, frac=0.999, ulps=1.01):
        eq = float((mine == ref).float().mean())
        ulp = torch.clamp(ref.abs(), min=float(ref.pow(2).mean().sqrt())) * 2.0 ** -10      # fp16 spacing at the value, floored at the tensor's rms
        worst = float(((mine - ref).abs() / ulp).max())
        assert eq >= frac and worst <= ulps, (name, eq, worst)
    xo.AMP16 = True
    try:
        with torch.no_grad():
            check("patch_embed", xo.patch_embed(tp("patch_embed/in")[:, :1], sd, "encoder.patch_embed."), tp("patch_embed/out"), frac=0.995, ulps=2.01)
            check("norm", xo._h(F.layer_norm(tp("b0.norm/in"), (96,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)), tp("b0.norm/out"))
            check("ss2d (in_proj .. out_proj)", xo.ss2d(tp("b0.op/in"), sd, p + "op."), tp("b0.op/out"), frac=0.98, ulps=2.01)
            check("block 0", xo.vss_block(tp("b0/in"), sd, p), tp("b0/out"), frac=0.97, ulps=3.01)
            check("block 1", xo.vss_block(tp("b1/in"), sd, "encoder.layers.0.blocks.1."), tp("b1/out"), frac=0.97, ulps=3.01)
            check("downsample 0", xo.downsample(tp("ds0/in"), sd, "encoder.layers.0.downsample."), tp("ds0/out"), frac=0.995, ulps=2.01)
            d = "detector_head_convolutions."
            x = xo._conv(tp("head_det.1/in"), sd[d + "1.weight"], sd[d + "1.bias"])
            check("head conv", x, tp("head_det.1/out"), frac=0.995, ulps=1.01)
            check("head relu + bn", xo._bn(F.relu(tp("head_det.3/in")), sd, d + "3."), tp("head_det.3/out"))
            data = synth.to_torch(synth.make_pair_batch(0, 1, H, W))
            o, t, _ = xo.xpoint_forward(data, sd)
    finally:
        xo.AMP16 = False
    with torch.no_grad():
        o32, _, _ = xo.xpoint_forward(data, sd)
    for k in ("prob", "desc", "encoder_output"):
        ref = g[f"64x96/optical/{k}"]
        e_amp = float(np.abs(o[k].numpy() - ref).max()); e_f32 = float(np.abs(o32[k].numpy() - ref).max())
        assert e_amp < 3.0 * e_f32 + 1e-3, (k, e_amp, e_f32)          # same noise level as the recipe's own distance from f32
    assert torch.equal(o["encoder_output"], o["encoder_output"].half().float())


def test_bench_pmc_lookup_covers_the_committed_counter_files():
    """bench.py maps the library's HIP-event tags to the kernel names of the committed rocprofv3 PMC summaries (profiles/pmc_traffic.json,
    pmc_mfma.json) for roofline.traffic / frac_mfma_busy_pmc: every tag that can become the dominant kernel of a bench line must resolve to exactly
    one kernel of the committed files (a renamed template or a new schedule would otherwise surface as traffic_error in the driver's line)."""
    import importlib.util, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)          # bench.py imports no torch at module level (the launcher parent must stay GPU-free)
    for fn in ("pmc_traffic.json", "pmc_mfma.json"):
        names = list(json.load(open(os.path.join(root, "profiles", fn)))["kernels"])
        for tag in ("gemm_ring_h2s_256x256", "gemm_ring_h2s_256x128", "gemm_ring_h2s_128x128", "gemm_h2_mfma_128x128", "gemm_h2_mfma_64x128", "proj_mlp_fused_h2_c192", "proj_mlp_fused_h2_c96",
                    "conv3x3_h2r_mfma_128x96", "conv3x3_h2r_mfma_128x128"):
            assert bench.pmc_kernel_for_tag(tag, names) in names, (fn, tag)
    # the fast mixed-precision class has its own counter files (bench.py --precision-class amp16f reads pmc_*_amp16f.json)
    for fn in ("pmc_mfma_amp16f.json", "pmc_traffic_amp16f.json"):
        path = os.path.join(root, "profiles", fn)
        if not os.path.exists(path):
            continue
        names = list(json.load(open(path))["kernels"])
        for tag in ("gemm_f16_mfma_128x128", "gemm_f16_mfma_128x96_k32", "gemm_f16_mfma_128x192", "gemm_f16_mfma_256x128", "conv3x3_f16_mfma_128x128",
                    "conv3x3_f16_mfma_128x96", "ln_mlp_fused_f16_c96", "ln_mlp_fused_f16_c192", "ln_proj_f16_c96", "ln_proj_f16_c192"):
            assert bench.pmc_kernel_for_tag(tag, names) in names, (fn, tag)
    # tags that cover one template instance per stage (the chunked SS2D passes) resolve to the whole family, in either spelling of the kernel name
    for fn in ("pmc_traffic.json", "pmc_traffic_amp16f.json"):
        path = os.path.join(root, "profiles", fn)
        if os.path.exists(path):
            names = list(json.load(open(path))["kernels"])
            for tag in ("ss2d_pass1", "ss2d_pass3_row", "ss2d_pass3_col_ln"):
                fam = bench.pmc_family_for_tag(tag, names)
                assert len(fam) >= 2 and all(n in names for n in fam), (fn, tag, fam)      # stages 0 and 1 (stage 2 took the sequential form in round 4)
            assert len(bench.pmc_family_for_tag("ss2d_pass2", names)) == 1
            assert set(bench.pmc_family_for_tag("ss2d_pass3_row", names)).isdisjoint(bench.pmc_family_for_tag("ss2d_pass3_col_ln", names))
    names = list(json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))["kernels"])
    for tag in ("ss2d_pass2", "dwconv3x3_silu", "layernorm"):          # HBM-bound tags: traffic file only
        try:
            bench.pmc_kernel_for_tag(tag, names)
        except KeyError as e:               # several template instances share a tag (ss2d passes, layernorm widths): reported, not swallowed
            assert "matches" in str(e)
