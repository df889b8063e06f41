"""Deterministic synthetic weights and images.

The reference ships no XPoint weights and there is no network, so every parity
test, the oracle pinning run against the real reference, `smoke()` and `bench.py` use weights
generated here.  The generator is bit-reproducible across machines / numpy versions: a 64-bit
integer hash (FNV-1a of the tensor name + splitmix64 of the element index) gives a uniform
`k / 2**24`, and only exact IEEE-754 float32 multiply/add is applied afterwards (no libm).

State-dict names and shapes follow the reference model built from
`model_weights/XPoint-EXP1/params.yaml` (reference
`xpoint/models/XPoint.py:65-142`, `xpoint/models/vmamba_src/VMamba.py:381-491,1405-1440`).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

_MASK = (1 << 64) - 1


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def hash_uniform(name: str, n: int) -> np.ndarray:
    """n float32 values in [0,1), each exactly k/2**24, from hash(name, index)."""
    seed = np.uint64(_fnv1a(name))
    with np.errstate(over="ignore"):
        z = seed + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    k = (z >> np.uint64(40)).astype(np.float32)  # top 24 bits, exact in float32
    return k * np.float32(1.0 / 16777216.0)


def uniform(name: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(name, n)
    out = u * np.float32(hi - lo) + np.float32(lo)
    return out.astype(np.float32).reshape(shape)


# ------------------------------------------------------------------------------------------
# model dimension helpers
# ------------------------------------------------------------------------------------------

VSSM_TINY_SEG = dict(EMBED_DIM=96, DEPTHS=[2, 2, 2, 2], SSM_D_STATE=1, SSM_RATIO=1.0, SSM_DT_RANK="auto",
                     SSM_CONV=3, SSM_CONV_BIAS=False, SSM_FORWARDTYPE="v05_noz", MLP_RATIO=4.0,
                     DOWNSAMPLE="v3", PATCHEMBED="v2")


def xpoint_exp1_config(height=480, width=640, hm_head=False, vssm=None, descriptor_size=256) -> dict:
    """The `model:` block of reference model_weights/XPoint-EXP1/params.yaml:89-135 as a dict,
    with the H/W patch of reference benchmark.py:70-76 applied and the hm head switchable
    (reference configs/cipdp.yaml:48 `disable_hmhead: true`)."""
    v = dict(VSSM_TINY_SEG)
    if vssm:
        v.update(vssm)
    return {
        "type": "XPoint",
        "bn_first": False, "descriptor_head": True, "descriptor_size": descriptor_size,
        "final_batchnorm": True,
        "homography_regression_head": {"check": bool(hm_head), "type": "RegNet"},
        "intepolation_mode": "bilinear", "mixed_precision": True, "multispectral": False,
        "normalize_descriptors": True, "reflection_pad": True, "takes_pair": True,
        "use_attention": {
            "check": True, "type": "VMamba", "height": height, "width": width,
            "model_parameters": {"DATA": {"IMG_SIZE": 512},
                                 "MODEL": {"DROP_PATH_RATE": 0.2, "NAME": "vssm_tiny_segmentation",
                                           "TYPE": "vssm", "VSSM": v}},
            "pretrained": {"check": True, "type_dir": "", "yaml_file": ""},
        },
    }


def vssm_dims(vssm: dict):
    E = int(vssm["EMBED_DIM"])
    depths = list(vssm["DEPTHS"])
    dims = [E * (2 ** i) for i in range(len(depths))]
    N = int(vssm["SSM_D_STATE"])
    ratio = float(vssm.get("SSM_RATIO", 1.0))
    assert ratio == 1.0, "only SSM_RATIO 1.0 (the XPoint config) is supported"
    mlp = float(vssm.get("MLP_RATIO", 4.0))
    dtr = vssm.get("SSM_DT_RANK", "auto")
    ranks = [int(math.ceil(d / 16)) if dtr == "auto" else int(dtr) for d in dims]
    return dims, depths, N, ranks, mlp


def xpoint_state_spec(cfg: dict) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """Ordered {reference state_dict key: (shape, kind)} for the VMamba XPoint model
    (order = encoder -> hm_regressor -> detector head -> descriptor head)."""
    vssm = cfg["use_attention"]["model_parameters"]["MODEL"]["VSSM"]
    dims, depths, N, ranks, mlp = vssm_dims(vssm)
    E = dims[0]
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()

    def ln(prefix, c):
        spec[prefix + ".weight"] = ((c,), "ln_w")
        spec[prefix + ".bias"] = ((c,), "ln_b")

    def bn(prefix, c):
        spec[prefix + ".weight"] = ((c,), "bn_w")
        spec[prefix + ".bias"] = ((c,), "bn_b")
        spec[prefix + ".running_mean"] = ((c,), "bn_mean")
        spec[prefix + ".running_var"] = ((c,), "bn_var")
        spec[prefix + ".num_batches_tracked"] = ((), "bn_count")

    # multispectral (XPoint.py:98-100): two encoders, created thermal first; else one shared encoder
    prefixes = ["encoder_thermal.", "encoder_optical."] if cfg.get("multispectral", False) else ["encoder."]
    for p in prefixes:
        spec[p + "patch_embed.0.weight"] = ((E // 2, 3, 3, 3), "conv_w")
        spec[p + "patch_embed.0.bias"] = ((E // 2,), "conv_b:27")
        ln(p + "patch_embed.2", E // 2)
        spec[p + "patch_embed.5.weight"] = ((E, E // 2, 3, 3), "conv_w")
        spec[p + "patch_embed.5.bias"] = ((E,), "conv_b:%d" % (9 * (E // 2)))
        ln(p + "patch_embed.7", E)
        for s, (C, depth) in enumerate(zip(dims, depths)):
            R = ranks[s]
            H4 = int(C * mlp)
            for j in range(depth):
                b = f"{p}layers.{s}.blocks.{j}."
                ln(b + "norm", C)
                spec[b + "op.x_proj_weight"] = ((4, R + 2 * N, C), "xproj")
                spec[b + "op.A_logs"] = ((4 * C, N), "A_logs")
                spec[b + "op.Ds"] = ((4 * C,), "Ds")
                spec[b + "op.dt_projs_weight"] = ((4, C, R), "dt_w")
                spec[b + "op.dt_projs_bias"] = ((4, C), "dt_b")
                ln(b + "op.out_norm", C)
                spec[b + "op.in_proj.weight"] = ((C, C), "lin_w")
                spec[b + "op.conv2d.weight"] = ((C, 1, 3, 3), "conv_w")
                spec[b + "op.out_proj.weight"] = ((C, C), "lin_w")
                ln(b + "norm2", C)
                spec[b + "mlp.fc1.weight"] = ((H4, C), "lin_w")
                spec[b + "mlp.fc1.bias"] = ((H4,), "lin_b")
                spec[b + "mlp.fc2.weight"] = ((C, H4), "lin_w")
                spec[b + "mlp.fc2.bias"] = ((C,), "lin_b")
            if s < len(dims) - 1:
                d = f"{p}layers.{s}.downsample."
                spec[d + "1.weight"] = ((2 * C, C, 3, 3), "conv_w")
                spec[d + "1.bias"] = ((2 * C,), "conv_b:%d" % (9 * C))
                ln(d + "3", 2 * C)
    enc_c = E // 2
    if cfg.get("homography_regression_head", {}).get("check", False):
        spec["hm_regressor.layer1.0.weight"] = ((96, enc_c, 3, 3), "conv_w")
        bn("hm_regressor.layer1.1", 96)
        spec["hm_regressor.layer1.3.weight"] = ((192, 96, 3, 3), "conv_w")
        bn("hm_regressor.layer1.4", 192)
        spec["hm_regressor.fc.1.weight"] = ((64, 256), "lin_w")
        spec["hm_regressor.fc.1.bias"] = ((64,), "lin_b")
        spec["hm_regressor.fc.4.weight"] = ((8, 64), "lin_w")
        spec["hm_regressor.fc.4.bias"] = ((8,), "lin_b")
    head_c = 256
    dsz = int(cfg.get("descriptor_size", 256))
    for hname, outc in (("detector_head_convolutions", 65), ("descriptor_head_convolutions", dsz)):
        spec[f"{hname}.1.weight"] = ((head_c, enc_c, 3, 3), "conv_w")
        spec[f"{hname}.1.bias"] = ((head_c,), "conv_b:%d" % (9 * enc_c))
        bn(f"{hname}.3", head_c)
        spec[f"{hname}.4.weight"] = ((outc, head_c, 1, 1), "conv_w")
        spec[f"{hname}.4.bias"] = ((outc,), "conv_b:%d" % head_c)
        bn(f"{hname}.5", outc)
    return spec


# Scale applied to the detector's last 1x1 conv so the synthetic heat-map is peaky (a trained
# detector is; a random one is nearly uniform).  Calibrated once against the reference oracle
# (oracle/refharness/make_golden.py) and frozen: see tests/golden/README.md.
DETECTOR_GAIN = 8.0


def _gen(name: str, shape, kind: str, tag: str) -> np.ndarray:
    key = tag + "/" + name
    if kind == "ln_w":
        return uniform(key, shape, 0.8, 1.2)
    if kind == "ln_b":
        return uniform(key, shape, -0.1, 0.1)
    if kind == "bn_w":
        return uniform(key, shape, 0.5, 1.5)
    if kind in ("bn_b", "bn_mean"):
        return uniform(key, shape, -0.2, 0.2)
    if kind == "bn_var":
        return uniform(key, shape, 0.5, 1.5)
    if kind == "bn_count":
        return np.zeros((), dtype=np.int64)
    if kind in ("lin_w", "xproj"):
        bound = 1.0 / math.sqrt(shape[-1])
        return uniform(key, shape, -bound, bound)
    if kind == "lin_b":
        return uniform(key, shape, -0.05, 0.05)
    if kind == "conv_w":
        fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / math.sqrt(fan_in)
        return uniform(key, shape, -bound, bound)
    if kind.startswith("conv_b:"):
        bound = 1.0 / math.sqrt(int(kind.split(":")[1]))
        return uniform(key, shape, -bound, bound)
    if kind == "A_logs":
        n = shape[1]
        base = np.log(np.arange(1, n + 1, dtype=np.float64)).astype(np.float32)[None, :]
        return (uniform(key, shape, -0.5, 0.5) + base).astype(np.float32)
    if kind == "Ds":
        return uniform(key, shape, 0.5, 1.5)
    if kind == "dt_w":
        bound = shape[-1] ** -0.5
        return uniform(key, shape, -bound, bound)
    if kind == "dt_b":
        # ~ softplus^-1 of dt in [1e-3, 0.1]  (reference VMamba.py:176-183), without libm
        return uniform(key, shape, -6.9, -2.25)
    raise ValueError(kind)


def make_state_dict(cfg: dict, tag: str = "xpoint-synth-v1", detector_gain: float = DETECTOR_GAIN) -> "OrderedDict[str, np.ndarray]":
    """Numpy state dict keyed exactly like the reference `XPoint(cfg).state_dict()`."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, (shape, kind) in xpoint_state_spec(cfg).items():
        sd[name] = _gen(name, shape, kind, tag)
    k = "detector_head_convolutions.4.weight"
    sd[k] = (sd[k] * np.float32(detector_gain)).astype(np.float32)
    return sd


def make_torch_state_dict(cfg: dict, **kw):
    import torch
    return OrderedDict((k, torch.from_numpy(np.array(v, copy=True))) for k, v in make_state_dict(cfg, **kw).items())


# ------------------------------------------------------------------------------------------
# "trained-like" weight statistics: the parity fixtures above use default-init-like weights whose
# activations stay below ~14.  Trained VMamba checkpoints have heavy-tailed LayerNorm gains and a few outlier channels in the
# residual stream, which reach the un-normalised downsample convolutions (reference VMamba.py:1405-1440).  This profile keeps
# the generator's exactness rules (hash uniforms, exact float32 multiply / add, powers of two by ldexp — no libm).
# ------------------------------------------------------------------------------------------

def _hash_int(name: str, n: int, lo: int, hi: int) -> np.ndarray:
    """n integers in [lo, hi] from the name's hash stream (exact: floor of a 24-bit uniform times the range)."""
    u = hash_uniform(name, n).astype(np.float64)
    return (np.floor(u * (hi - lo + 1)).astype(np.int64) + lo).clip(lo, hi)


def make_trained_like_state_dict(cfg: dict, tag: str = "xpoint-trainedlike-v1", detector_gain: float = DETECTOR_GAIN,
                                 gain_exp=(-4, 3), outlier_frac=0.01, outlier_scale=100.0, head_in_exp=-7) -> "OrderedDict[str, np.ndarray]":
    """make_state_dict with trained-like statistics:
      * every LayerNorm gain log-uniform over 2^gain_exp[0] .. 2^(gain_exp[1] + 1) (default 0.0625 .. 16): (1 + u) * 2^k;
      * outlier_frac of the OUTPUT channels (at least one) of the patch-embed and downsample convolutions scaled by outlier_scale —
        those feed LayerNorms whose statistics they then dominate — and the same fraction of the block out_proj / fc2 rows, which write
        straight into the residual stream the downsample convolutions read un-normalised;
      * dt_projs_bias at both ends of its range (softplus^-1 of 1e-3 and of 0.1) instead of uniform between them;
      * the heads' first convolutions scaled by 2^head_in_exp: the encoder output is the RAW residual stream (no final norm, VMamba.py:1500-1505),
        here ~1e3, and trained heads are adapted to their input's scale (their BatchNorm statistics are fixed numbers, not a normalisation)."""
    sd = make_state_dict(cfg, tag=tag, detector_gain=detector_gain)
    spec = xpoint_state_spec(cfg)
    for name, (shape, kind) in spec.items():
        key = tag + "/" + name
        if kind == "ln_w":
            k = _hash_int(key + "/exp", shape[0], gain_exp[0], gain_exp[1])
            sd[name] = np.ldexp(np.float32(1.0) + hash_uniform(key + "/mant", shape[0]), k).astype(np.float32)
        elif kind == "dt_b":
            bit = _hash_int(key + "/end", int(np.prod(shape)), 0, 1).reshape(shape)
            sd[name] = np.where(bit == 1, np.float32(-2.25), np.float32(-6.9)).astype(np.float32)
        elif (kind == "conv_w" and (".patch_embed." in name or ".downsample." in name)) or \
                (kind == "lin_w" and (name.endswith("op.out_proj.weight") or name.endswith("mlp.fc2.weight"))):
            co = shape[0]
            n_out = max(1, int(round(outlier_frac * co)))
            rows = _hash_int(key + "/outliers", n_out, 0, co - 1)
            w = sd[name].copy()
            w[rows] = (w[rows] * np.float32(outlier_scale)).astype(np.float32)
            sd[name] = w
    for h in ("detector_head_convolutions", "descriptor_head_convolutions"):
        sd[h + ".1.weight"] = np.ldexp(sd[h + ".1.weight"], head_in_exp).astype(np.float32)
    return sd


def make_contrast_pair(pair_index: int, H: int, W: int):
    """A pair at the contrast extremes: the optical image nearly flat (0.5 +- 2^-7, a hazy frame), the thermal one saturated (values pushed
    to exactly 0 / 1 outside the middle band, a hot-spot frame).  Same dict layout as make_pair_batch(pair_index, 1, H, W)."""
    d = make_pair_batch(pair_index, 1, H, W)
    o = d["optical"]["image"]
    d["optical"]["image"] = (np.float32(0.5) + (o - np.float32(0.5)) * np.float32(1.0 / 64.0)).astype(np.float32)
    t = d["thermal"]["image"]
    d["thermal"]["image"] = np.where(t < np.float32(0.25), np.float32(0.0), np.where(t > np.float32(0.75), np.float32(1.0), t)).astype(np.float32)
    return d


# ------------------------------------------------------------------------------------------
# conv-encoder XPoint (BASELINE config 1, reference XPoint.py:451-466, model_weights/multipoint/params.yaml)
# ------------------------------------------------------------------------------------------

def multipoint_config(descriptor_size=64) -> dict:
    """`model:` block of reference model_weights/multipoint/params.yaml (conv encoder, takes_pair False)."""
    return {"type": "XPoint", "bn_first": False, "descriptor_head": True, "descriptor_size": descriptor_size,
            "final_batchnorm": True, "intepolation_mode": "bilinear", "multispectral": False,
            "normalize_descriptors": True, "reflection_pad": True}


CONV_XPOINT_CONVS = [1, 5, 10, 14, 19, 23, 28, 32]      # nn.Sequential indices
CONV_XPOINT_CHANNELS = [1, 64, 64, 64, 64, 128, 128, 128, 128]   # channel_version 0: [1,64,64,128,128], double convolution


def conv_xpoint_state_spec(cfg: dict):
    spec = OrderedDict()
    ch = CONV_XPOINT_CHANNELS
    for li, idx in enumerate(CONV_XPOINT_CONVS):
        ci, co = ch[li], ch[li + 1]
        spec[f"encoder.{idx}.weight"] = ((co, ci, 3, 3), "conv_w")
        spec[f"encoder.{idx}.bias"] = ((co,), "conv_b:%d" % (9 * ci))
        b = f"encoder.{idx + 2}"
        spec[b + ".weight"] = ((co,), "bn_w"); spec[b + ".bias"] = ((co,), "bn_b")
        spec[b + ".running_mean"] = ((co,), "bn_mean"); spec[b + ".running_var"] = ((co,), "bn_var")
        spec[b + ".num_batches_tracked"] = ((), "bn_count")
    enc_c, head_c, dsz = 128, 256, int(cfg.get("descriptor_size", 256))
    for hname, outc in (("detector_head_convolutions", 65), ("descriptor_head_convolutions", dsz)):
        spec[f"{hname}.1.weight"] = ((head_c, enc_c, 3, 3), "conv_w")
        spec[f"{hname}.1.bias"] = ((head_c,), "conv_b:%d" % (9 * enc_c))
        for bi, c in ((3, head_c), (5, outc)):
            b = f"{hname}.{bi}"
            spec[b + ".weight"] = ((c,), "bn_w"); spec[b + ".bias"] = ((c,), "bn_b")
            spec[b + ".running_mean"] = ((c,), "bn_mean"); spec[b + ".running_var"] = ((c,), "bn_var")
            spec[b + ".num_batches_tracked"] = ((), "bn_count")
            if bi == 3:
                spec[f"{hname}.4.weight"] = ((outc, head_c, 1, 1), "conv_w")
                spec[f"{hname}.4.bias"] = ((outc,), "conv_b:%d" % head_c)
    # reorder to the reference's registration order (1, 3, 4, 5 per head)
    ordered = OrderedDict()
    for k in spec:
        if k.startswith("encoder."):
            ordered[k] = spec[k]
    for hname in ("detector_head_convolutions", "descriptor_head_convolutions"):
        for bi in (1, 3, 4, 5):
            for k in spec:
                if k.startswith(f"{hname}.{bi}."):
                    ordered[k] = spec[k]
    return ordered


def make_conv_xpoint_state_dict(cfg: dict, tag: str = "convxpoint-synth-v1", detector_gain: float = 24.0):
    sd = OrderedDict()
    for name, (shape, kind) in conv_xpoint_state_spec(cfg).items():
        sd[name] = _gen(name, shape, kind, tag)
    k = "detector_head_convolutions.4.weight"
    sd[k] = (sd[k] * np.float32(detector_gain)).astype(np.float32)
    return sd


# ------------------------------------------------------------------------------------------
# SuperPointMagicLeap (BASELINE config 1, reference SuperPointMagicLeap.py:16-29)
# ------------------------------------------------------------------------------------------

def superpoint_state_spec():
    c1, c2, c3, c4, c5, d1 = 64, 64, 128, 128, 256, 256
    layers = [("conv1a", 1, c1, 3), ("conv1b", c1, c1, 3), ("conv2a", c1, c2, 3), ("conv2b", c2, c2, 3),
              ("conv3a", c2, c3, 3), ("conv3b", c3, c3, 3), ("conv4a", c3, c4, 3), ("conv4b", c4, c4, 3),
              ("convPa", c4, c5, 3), ("convPb", c5, 65, 1), ("convDa", c4, c5, 3), ("convDb", c5, d1, 1)]
    spec = OrderedDict()
    for n, ci, co, k in layers:
        spec[n + ".weight"] = ((co, ci, k, k), "conv_w")
        spec[n + ".bias"] = ((co,), "conv_b:%d" % (ci * k * k))
    return spec


def make_superpoint_state_dict(tag: str = "superpoint-synth-v1", detector_gain: float = 4.0):
    sd = OrderedDict()
    for name, (shape, kind) in superpoint_state_spec().items():
        sd[name] = _gen(name, shape, kind, tag)
    sd["convPb.weight"] = (sd["convPb.weight"] * np.float32(detector_gain)).astype(np.float32)
    return sd


# ------------------------------------------------------------------------------------------
# images
# ------------------------------------------------------------------------------------------

def make_image(pair_index: int, spectrum: str, H: int, W: int) -> np.ndarray:
    """(1,H,W) float32 uniform [0,1) image; seed = 1000 + pair_index."""
    return hash_uniform(f"image/{1000 + int(pair_index)}/{spectrum}/{H}x{W}", H * W).reshape(1, H, W)


def make_pair_batch(first_pair: int, batch: int, H: int, W: int):
    """Returns the reference `data` dict layout (numpy): optical/thermal image (B,1,H,W) f32,
    valid_mask (B,1,H,W) bool all ones, is_optical (B,1) bool
    (reference xpoint/datasets/ImagePairDataset.py:403-423)."""
    out = {}
    for spectrum, flag in (("optical", True), ("thermal", False)):
        img = np.stack([make_image(first_pair + i, spectrum, H, W) for i in range(batch)], 0)
        out[spectrum] = {
            "image": img.astype(np.float32),
            "valid_mask": np.ones((batch, 1, H, W), dtype=bool),
            "is_optical": np.full((batch, 1), flag, dtype=bool),
        }
    return out


def to_torch(data, device="cpu"):
    import torch
    if isinstance(data, dict):
        return {k: to_torch(v, device) for k, v in data.items()}
    return torch.from_numpy(np.ascontiguousarray(data)).to(device)


def make_eval_case(seed: int, B: int = 2, H: int = 96, W: int = 128, n_base: int = 260, n_extra: int = 90, D: int = 256):
    """Synthetic input of the evaluation harness (reference benchmark_evaluation.py): heat maps with peaks at the images of
    common scene points under two homographies (plus unrelated peaks), coarse descriptor maps sampled from one smooth
    field in the scene frame (plus noise), valid masks with a border.  Returns numpy arrays:
    prob_optical/prob_thermal (B,1,H,W), desc_optical/desc_thermal (B,D,H/8,W/8), mask_* (B,1,H,W), H_optical/H_thermal (B,3,3)."""
    out = {k: [] for k in ("prob_optical", "prob_thermal", "desc_optical", "desc_thermal", "mask_optical", "mask_thermal",
                           "H_optical", "H_thermal")}
    Hc, Wc = H // 8, W // 8
    freq = uniform(f"eval{seed}/freq", (D, 2), -0.09, 0.09).astype(np.float64)
    phase = uniform(f"eval{seed}/phase", (D,), 0.0, 6.283).astype(np.float64)
    for b in range(B):
        hs = []
        for spec in ("optical", "thermal"):
            r = uniform(f"eval{seed}/{b}/{spec}/h", (8,), -1.0, 1.0).astype(np.float64)
            # (x, y) convention of cv2.perspectiveTransform; mild rotation / scale / shift / perspective
            hm = np.array([[1.0 + 0.06 * r[0], 0.05 * r[1], 4.0 * r[2]],
                           [0.05 * r[3], 1.0 + 0.06 * r[4], 4.0 * r[5]],
                           [2e-4 * r[6], 2e-4 * r[7], 1.0]])
            hs.append(hm)
        base = np.stack([uniform(f"eval{seed}/{b}/bx", (n_base,), 6.0, W - 7.0), uniform(f"eval{seed}/{b}/by", (n_base,), 6.0, H - 7.0)], 1).astype(np.float64)
        for si, spec in enumerate(("optical", "thermal")):
            hm = hs[si]
            pts = np.concatenate([base, np.ones((n_base, 1))], 1) @ hm.T
            pts = pts[:, :2] / pts[:, 2:3]
            prob = np.zeros((H, W), np.float32)
            score = uniform(f"eval{seed}/{b}/{spec}/score", (n_base,), 0.05, 0.95)
            for (x, y), sc in zip(np.rint(pts).astype(int), score):
                if 0 <= y < H and 0 <= x < W:
                    prob[y, x] = max(prob[y, x], sc)
            ex = np.stack([uniform(f"eval{seed}/{b}/{spec}/ex", (n_extra,), 0, W - 1), uniform(f"eval{seed}/{b}/{spec}/ey", (n_extra,), 0, H - 1)], 1)
            for (x, y), sc in zip(np.rint(ex).astype(int), uniform(f"eval{seed}/{b}/{spec}/es", (n_extra,), 0.02, 0.6)):
                prob[y, x] = max(prob[y, x], sc)
            mask = np.zeros((H, W), np.float32); mask[3:H - 3, 2:W - 4] = 1.0
            # coarse descriptor map: cell centre -> scene frame -> smooth field (+ noise), L2-normalised over channels
            cy, cx = np.meshgrid((np.arange(Hc) + 0.5) * 8 - 0.5, (np.arange(Wc) + 0.5) * 8 - 0.5, indexing="ij")
            cells = np.stack([cx.ravel(), cy.ravel(), np.ones(Hc * Wc)], 1) @ np.linalg.inv(hm).T
            cells = cells[:, :2] / cells[:, 2:3]
            f = np.sin(cells @ freq.T + phase[None, :]) + 0.15 * uniform(f"eval{seed}/{b}/{spec}/dn", (Hc * Wc, D), -1.0, 1.0)
            f = f / np.linalg.norm(f, axis=1, keepdims=True)
            out[f"prob_{spec}"].append(prob[None]); out[f"mask_{spec}"].append(mask[None])
            out[f"desc_{spec}"].append(f.T.reshape(D, Hc, Wc).astype(np.float32))
            out[f"H_{spec}"].append(hm.astype(np.float32))
    return {k: np.stack(v) for k, v in out.items()}
