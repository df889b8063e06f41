"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/xpoint_hip.h declares, and the
host-side logic that needs no GPU (context / parameter layout / weight packing / config and error behaviour)
is correct.  No compute entry point is called here."""
import ctypes
import os

import numpy as np
import pytest
import torch

from xpoint_amd import _lib, synth


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _lib.exported_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    bound = set(_lib._SIGNATURES) | set(_lib._SIZE_QUERIES) | {"xp_version", "xp_last_error"}
    assert set(declared) == bound
    assert _lib.load().xp_version() == 100


def test_argument_errors_surface_with_message():
    lib = _lib.load()
    rc = lib.xp_selective_scan_fwd(None, None, None, None, None, None, None, None, None, 1, 4, 4, 8, 1, 1, 1, None)
    assert rc < 0 and b"null" in lib.xp_last_error()
    with pytest.raises(_lib.XPointHipError):
        _lib.call("xp_gemm_nt", None, None, None, None, None, None, None, 1, 1, 4, 4, 1, 1, 0, None)


def test_context_layout_and_weight_packing():
    from xpoint_amd import models
    cfg = synth.xpoint_exp1_config(480, 640)
    net = models.XPoint(cfg)
    sd = synth.make_torch_state_dict(cfg)
    r = net.load_state_dict(sd, strict=True)
    assert r.missing_keys == [] and r.unexpected_keys == []
    blob = net.pack_weights()
    # device-format blob: every reference tensor except BN statistics (folded into scale/shift) and the 3->1
    # channel fold of the stem; 81.6 MB of fp32 as SURVEY.md 5.8 estimates
    assert blob.dtype == torch.float32 and 20.3e6 < blob.numel() < 20.5e6
    lay = net._layout
    assert list(lay)[0] == "stem.w" and "s3.b1.fc2_w" in lay and "desc2.shift" in lay
    offs = sorted(lay.values())
    for (o1, n1), (o2, _) in zip(offs, offs[1:]):
        assert o1 + n1 <= o2 and o2 % 4 == 0                       # no overlap, 16-byte aligned
    o, n = lay["s0.b0.A"]
    A = -torch.exp(sd["encoder.layers.0.blocks.0.op.A_logs"].float()).view(4, 96)[[0, 2, 1, 3]].reshape(-1)
    assert torch.equal(blob[o:o + n], A)
    o, n = lay["s1.b0.xproj_w"]
    assert torch.equal(blob[o:o + n], sd["encoder.layers.1.blocks.0.op.x_proj_weight"][[0, 2, 1, 3]].reshape(-1))
    lib = _lib.load()
    assert lib.xp_forward_workspace_bytes(net._ctx, 16, 480, 640) > 0
    Hc, Wc, Ce = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.xp_forward_shapes(net._ctx, 1, 480, 640, ctypes.byref(Hc), ctypes.byref(Wc), ctypes.byref(Ce)) == 0
    assert (Hc.value, Wc.value, Ce.value) == (60, 80, 48)


def test_model_config_and_state_dict_errors():
    from xpoint_amd import models
    cfg = synth.xpoint_exp1_config(64, 96)
    net = models.XPoint(cfg)
    assert net.takes_pair() and net.get_encoder_downsample_ratio() == 8
    with pytest.raises(ValueError):
        net.set_force_return_logits("yes")
    sd = synth.make_torch_state_dict(cfg)
    bad = dict(sd); bad["encoder.patch_embed.0.weight"] = torch.zeros(48, 3, 5, 5)
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)
    extra = dict(sd); extra["bogus.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        net.load_state_dict(extra, strict=True)
    assert models.XPoint(cfg).load_state_dict(extra, strict=False).unexpected_keys == ["bogus.weight"]
    with pytest.raises(RuntimeError):                                 # forward before weights / on CPU
        models.XPoint(cfg).eval()({"optical": {"image": torch.zeros(1, 1, 64, 96)}, "thermal": {"image": torch.zeros(1, 1, 64, 96)}})
    # multispectral: two encoders' keys (thermal first, XPoint.py:98-100); conv-encoder multispectral stays unimplemented
    c2 = synth.xpoint_exp1_config(64, 96); c2["multispectral"] = True
    keys = list(models.XPoint(c2).expected_keys())
    assert keys[0].startswith("encoder_thermal.") and any(k.startswith("encoder_optical.") for k in keys) and "encoder.patch_embed.0.weight" not in keys
    with pytest.raises(NotImplementedError):
        c3 = synth.multipoint_config(); c3["multispectral"] = True
        models.XPoint(c3)
    from xpoint_amd.utils import fix_model_weigth_keys, dict_update
    assert list(fix_model_weigth_keys({"module__encoder.x": 1, "y": 2})) == ["encoder.x", "y"]
    assert dict_update({"a": {"b": 1, "c": 2}}, {"a": {"b": 3}}) == {"a": {"b": 3, "c": 2}}


def test_config_precision_opt_in(monkeypatch):
    """The f32 class is the default whatever `mixed_precision` says (the parity target is the reference's CPU path, which never autocasts, XPoint.py:182);
    use_config_precision() / XP_HONOR_MIXED_PRECISION=1 select what the reference runs on a GPU."""
    from xpoint_amd import models
    cfg = synth.xpoint_exp1_config(64, 96)
    assert cfg["mixed_precision"] is True
    net = models.XPoint(cfg)
    assert net.gemm_mode == "h2" and net.use_config_precision().gemm_mode == "amp16f"
    cfg2 = synth.xpoint_exp1_config(64, 96); cfg2["mixed_precision"] = False
    assert models.XPoint(cfg2).use_config_precision().gemm_mode == "h2"
    monkeypatch.setenv("XP_HONOR_MIXED_PRECISION", "1")
    assert models.XPoint(cfg).gemm_mode == "amp16f" and models.XPoint(cfg2).gemm_mode == "h2"
    monkeypatch.setenv("XP_GEMM_MODE", "x3")
    assert models.XPoint(cfg).gemm_mode == "x3"


def test_get_matches_host_errors():
    from xpoint_amd.utils import get_matches
    a = np.zeros((0, 256), np.float32)
    assert get_matches(a, a) == []
    with pytest.raises(ValueError):
        get_matches(np.zeros((2, 4), np.float32), np.zeros((2, 4), np.float32), "nope")
    with pytest.raises(NotImplementedError):
        get_matches(np.zeros((2, 4), np.float32), np.zeros((2, 4), np.float32), "flann")


def test_evaluation_host_logic_vs_reference_golden(golden):
    """§8(f): the host-side halves of xpoint_amd.evaluation (precision / recall / NN-mAP / M-score assembly, keypoint
    warping and filtering) against numbers produced by the reference's own benchmark_evaluation.py (g13)."""
    import numpy as np
    from xpoint_amd import evaluation as ev
    g = golden("g13_eval_metrics.npz")
    for seed in (0, 1):
        dd = {}
        for th in (2, 4):
            dd[th] = {k: g[f"s{seed}/desc/{th}/{k}"].tolist() for k in ("tp_optical", "tp_thermal", "distance_optical", "distance_thermal",
                                                                        "m_score_optical", "m_score_thermal")}
            dd[th]["n_gt_optical"] = int(g[f"s{seed}/desc/{th}/n_gt_optical"]); dd[th]["n_gt_thermal"] = int(g[f"s{seed}/desc/{th}/n_gt_thermal"])
        res = ev.compute_desc_dict(dd)
        for th in (2, 4):
            for k in ("nn_map_optical", "nn_map_thermal", "nn_map", "m_score"):
                assert abs(float(res[th][k]) - float(g[f"s{seed}/res/{th}/{k}"])) < 1e-12, (seed, th, k)
            assert np.allclose(res[th]["precision_thermal"], g[f"s{seed}/res/{th}/precision_thermal"], atol=1e-12)
    # warp_keypoints: (y, x) points through a homography acting on (x, y, 1); int return truncates toward zero
    h = np.array([[1.0, 0.0, 2.5], [0.0, 1.0, -3.5], [0.0, 0.0, 1.0]])
    kp = np.array([[10, 20], [0, 0], [5, 1]])
    assert ev.warp_keypoints(kp, h).tolist() == [[6, 22], [-3, 2], [1, 3]]
    assert np.allclose(ev.warp_keypoints(kp, h, float), [[6.5, 22.5], [-3.5, 2.5], [1.5, 3.5]])
    assert ev.filter_points(np.array([[-1, 2], [3, 4], [5, 200], [9, 9]]), (10, 100)).tolist() == [[3, 4], [9, 9]]
    assert ev.warp_keypoints(np.zeros((0, 2)), h).shape == (0, 2)


# ------------------------------------------------------------------------------------------------ data ingest (folder mode)
def _write_pair_folder(root, n=3, H0=70, W0=100, gray_thermal=True):
    from PIL import Image
    rng = np.random.default_rng(5)
    os.makedirs(os.path.join(root, "optical")); os.makedirs(os.path.join(root, "thermal"))
    raw = []
    for i in range(n):
        o = rng.integers(0, 256, (H0, W0, 3), dtype=np.uint8)
        t = rng.integers(0, 256, (H0, W0) if gray_thermal else (H0, W0, 3), dtype=np.uint8)
        Image.fromarray(o).save(os.path.join(root, "optical", f"im{i}.png"))
        Image.fromarray(t).save(os.path.join(root, "thermal", f"im{i}.png"))
        raw.append((o, t))
    open(os.path.join(root, "optical", "notes.txt"), "w").write("ignored")
    return raw


def test_image_pair_dataset_folder_mode(tmp_path):
    """ImagePairDataset folder mode (reference datasets/ImagePairDataset.py:48-74,122-128,199-208,254-274,331-420): member list,
    gray conversion (OpenCV 8-bit fixed point, restated), / 255, crop to multiples of 32 with the reference's RNG call order,
    output structure, error behaviour."""
    import random
    from xpoint_amd.datasets import ImagePairDataset, rgb_to_gray_u8, gray_lut
    raw = _write_pair_folder(str(tmp_path))
    ds = ImagePairDataset({"foldername": str(tmp_path), "height": 70, "width": 100})
    assert len(ds) == 3 and ds.memberslist == ["im0.png", "im1.png", "im2.png"]
    random.seed(3)
    s = ds[1]
    random.seed(3)
    i_h = random.randint(0, 70 - 64); i_w = random.randint(0, 100 - 96)          # 70 // 32 * 32 = 64, 100 // 32 * 32 = 96
    o, t = raw[1]
    g = ((o[..., 2].astype(np.int64) * 1868 + o[..., 1].astype(np.int64) * 9617 + o[..., 0].astype(np.int64) * 4899 + 8192) >> 14)
    assert np.array_equal(rgb_to_gray_u8(o), g.astype(np.uint8))
    exp_o = (g / 255.0)[i_h:i_h + 64, i_w:i_w + 96].astype(np.float32)
    exp_t = (t / 255.0)[i_h:i_h + 64, i_w:i_w + 96].astype(np.float32)
    assert s["optical"]["image"].shape == (1, 64, 96) and s["optical"]["image"].dtype == torch.float32
    assert np.array_equal(s["optical"]["image"][0].numpy(), exp_o) and np.array_equal(s["thermal"]["image"][0].numpy(), exp_t)
    assert s["optical"]["valid_mask"].dtype == torch.bool and bool(s["optical"]["valid_mask"].all())
    assert s["optical"]["is_optical"].tolist() == [True] and s["thermal"]["is_optical"].tolist() == [False] and s["name"] == "im1.png"
    assert np.array_equal(gray_lut()[g], (g / 255.0).astype(np.float32))           # the device path's table == numpy's division
    # no size requested -> whole image; the known gray values: pure R / G / B / white
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]]], dtype=np.uint8)
    assert rgb_to_gray_u8(px).tolist() == [[76, 150, 29, 255]]
    # errors, as the reference raises them
    with pytest.raises(ValueError):
        ImagePairDataset({"foldername": str(tmp_path / "missing")})
    with pytest.raises(ValueError):
        ImagePairDataset({})
    (tmp_path / "empty").mkdir()
    with pytest.raises(ValueError):
        ImagePairDataset({"foldername": str(tmp_path / "empty")})
    with pytest.raises(ValueError):
        ImagePairDataset({"foldername": str(tmp_path), "height": 128, "width": 96})[0]      # larger than the images
    with pytest.raises(NotImplementedError):
        ImagePairDataset({"foldername": str(tmp_path), "augmentation": {"photometric": {"enable": True}}})


def test_product_never_imports_the_oracle_and_has_no_cpu_fallback():
    """oracle/ is test infrastructure: no module of the product package may import it, and the product path raises (instead of
    silently computing on the CPU) when handed CPU tensors."""
    import glob
    import re
    pkg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "xpoint_amd")
    for f in glob.glob(os.path.join(pkg, "*.py")):
        src = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
    from xpoint_amd import models
    H, W = 64, 96
    cfg = synth.xpoint_exp1_config(H, W)
    net = models.XPoint(cfg)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.make_state_dict(cfg).items()}, strict=True)
    data = synth.to_torch(synth.make_pair_batch(0, 1, H, W))            # CPU tensors
    with pytest.raises(Exception):
        net.eval()(data)


def test_vmamba_needs_four_stages():
    """a 3-stage DEPTHS list would size the head for the wrong channel count / resolution; both the Python
    constructor and the C ABI refuse it (the reference fails with a channel mismatch in the head convolution)."""
    import ctypes
    import pytest
    from xpoint_amd import _lib, models, synth
    cfg = synth.xpoint_exp1_config(64, 96, vssm={"EMBED_DIM": 32, "DEPTHS": [1, 1, 1]})
    with pytest.raises(ValueError, match="4 stages"):
        models.XPoint(cfg)
    c = models._ModelCfg()
    c.embed_dim = 32; c.n_stages = 3; c.d_state = 1; c.mlp_ratio = 4.0; c.head_channels = 256; c.desc_size = 256; c.det_channels = 65
    ctx = ctypes.c_void_p()
    assert _lib.load().xp_ctx_create(ctypes.byref(c), ctypes.byref(ctx)) != 0
    assert b"4 stages" in _lib.load().xp_last_error()


def test_knob_registry_is_complete():
    """Every XP_* environment variable the sources read is in the ONE registry (csrc/xp_knobs.h, enumerated by xp_knob_count / xp_knob_info), and every
    registered knob is read somewhere: no undocumented switch, no stale entry."""
    import ctypes
    import glob
    import re
    lib = _lib.load()
    reg = {}
    for i in range(lib.xp_knob_count()):
        n, w, d = ctypes.c_char_p(), ctypes.c_char_p(), ctypes.c_char_p()
        assert lib.xp_knob_info(i, ctypes.byref(n), ctypes.byref(w), ctypes.byref(d)) == 0
        reg[n.value.decode()] = (w.value.decode(), d.value.decode())
    assert len(reg) == lib.xp_knob_count() and all(len(d) > 10 for _, d in reg.values())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    used = set()
    for path in glob.glob(os.path.join(root, "xpoint_amd", "csrc", "*")) + glob.glob(os.path.join(root, "xpoint_amd", "*.py")) + [os.path.join(root, "bench.py")]:
        if os.path.isdir(path) or path.endswith("xp_knobs.h"):
            continue
        txt = open(path, errors="ignore").read()
        used |= set(re.findall(r'getenv\("(XP_[A-Z0-9_]+)"', txt))
        used |= set(re.findall(r'environ(?:\.get)?[\(\[]\s*"(XP_[A-Z0-9_]+)"', txt))
    assert used - set(reg) == set(), f"read but not registered: {sorted(used - set(reg))}"
    assert set(reg) - used == set(), f"registered but read nowhere: {sorted(set(reg) - used)}"


def test_bench_pmc_numbers_are_tied_to_the_kernel_sources(tmp_path):
    """every PMC summary under profiles/ carries the hash of the kernel sources it was collected on; bench.py quotes its numbers with
    `traffic_stale` / `frac_mfma_busy_pmc_stale` = (hash differs from this tree), refuses a summary of another workload, and reports a missing file —
    shown by flipping one source byte in a copy of csrc/."""
    import importlib, json, shutil, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    from xpoint_amd import build
    h0 = build.source_hash()
    assert len(h0) == 16 and h0 == build.source_hash()
    csrc = tmp_path / "csrc"; shutil.copytree(os.path.join(root, "xpoint_amd", "csrc"), csrc, ignore=shutil.ignore_patterns("_obj"))
    assert build.source_hash(str(csrc)) == h0
    f = csrc / "mlp_fused.hip"; b = bytearray(f.read_bytes()); b[100] ^= 1; f.write_bytes(bytes(b))
    h1 = build.source_hash(str(csrc))
    assert h1 != h0
    tf, mf = tmp_path / "pmc_traffic.json", tmp_path / "pmc_mfma.json"
    kern = "mlp_fused_kernel<192, 8, 1, 3, true>"
    json.dump({"source_hash": h0, "workload": "c2", "kernels": {kern: {"launches": 4, "hbm_bytes_per_launch": 2.5e8}}}, open(tf, "w"))
    json.dump({"source_hash": h0, "workload": "c2", "kernels": {kern: {"launches": 4, "mfma_busy_frac": 0.34}}}, open(mf, "w"))
    fresh = bench.pmc_fields("proj_mlp_fused_h2_c192", "mfma", "c2", "f32", current_hash=h0, files=(str(tf), str(mf)))
    assert fresh["traffic"] == 250000000 and fresh["traffic_stale"] is False and fresh["frac_mfma_busy_pmc_stale"] is False and fresh["kernel_source_hash"] == h0
    stale = bench.pmc_fields("proj_mlp_fused_h2_c192", "mfma", "c2", "f32", current_hash=h1, files=(str(tf), str(mf)))
    assert stale["traffic"] == 250000000 and stale["traffic_stale"] is True and stale["frac_mfma_busy_pmc_stale"] is True and stale["traffic_source_hash"] == h0
    other = bench.pmc_fields("proj_mlp_fused_h2_c192", "mfma", "c4", "f32", current_hash=h0, files=(str(tf), str(mf)))     # C2's bytes are not C4's
    assert other["traffic"] is None and "workload" in other["traffic_error"] and "frac_mfma_busy_pmc_error" in other
    missing = bench.pmc_fields("proj_mlp_fused_h2_c192", "mfma", "c2", "f32", current_hash=h0, files=(str(tmp_path / "no.json"), str(mf)))
    assert missing["traffic"] is None and "traffic_error" in missing
    # per-configuration file names
    assert [os.path.basename(x) for x in bench.pmc_files("c2", "f32")] == ["pmc_traffic.json", "pmc_mfma.json"]
    assert [os.path.basename(x) for x in bench.pmc_files("c4", "f32")] == ["pmc_traffic_c4.json", "pmc_mfma_c4.json"]
    assert [os.path.basename(x) for x in bench.pmc_files("c2", "amp16f")] == ["pmc_traffic_amp16f.json", "pmc_mfma_amp16f.json"]
    # static reference data is labelled as such
    assert bench.dense_engine_ceilings().get("static_reference_data") is True
