"""Conv backbones of BASELINE config 1 and the RegNet homography head (config 5), orchestrated op by op from
Python over the C ABI (implicit-GEMM 3x3 convolutions on the fp32 MFMA engine, max-pool, softmax/shuffle, L2
normalisation).  They are small networks run at small sizes; the fused single-call path is the VMamba one.

  ConvEncoderXPoint     reference xpoint/models/XPoint.py:331-336,451-466 + heads :112-138
                        (model_weights/multipoint/params.yaml)
  SuperPointMagicLeap   reference xpoint/models/SuperPointMagicLeap.py:9-88 (model_weights/superpoint/params.yaml)
  regnet_forward        reference xpoint/models/RegNet.py:7-52 (valid at 256x256 input only)
"""
from __future__ import annotations

import collections

import numpy as np
import torch

from . import nnops as ops

_LoadResult = collections.namedtuple("_IncompatibleKeys", ["missing_keys", "unexpected_keys"])


def _load_into(store, spec, state_dict, strict):
    sd = dict(state_dict)
    missing = [k for k in spec if k not in sd]
    unexpected = [k for k in sd if k not in spec]
    for k, (shape, _) in spec.items():
        if k in sd and tuple(sd[k].shape) != tuple(shape):
            raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(shape)}")
    if strict and (missing or unexpected):
        raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...")
    for k in spec:
        if k in sd:
            t = sd[k]
            t = torch.from_numpy(np.array(t, copy=True)) if isinstance(t, np.ndarray) else t
            store[k] = t.detach().to("cpu").clone()
    return _LoadResult(missing, unexpected)


class ConvEncoderXPointImpl:
    """Device side of xpoint_amd.models.XPoint when the config selects the conv encoder."""
    CONVS = [1, 5, 10, 14, 19, 23, 28, 32]

    def __init__(self, ref_state, device):
        s = ref_state
        self.layers = []
        for li, idx in enumerate(self.CONVS):
            sc, sh = ops.bn_affine(s, f"encoder.{idx + 2}.")
            self.layers.append(dict(w=ops.conv_w(s[f"encoder.{idx}.weight"]).to(device), b=s[f"encoder.{idx}.bias"].float().to(device),
                                    sc=sc.to(device), sh=sh.to(device), pool=li in (1, 3, 5)))
        det, dsc = "detector_head_convolutions.", "descriptor_head_convolutions."
        self.head_w = ops.conv_w(torch.cat([s[det + "1.weight"], s[dsc + "1.weight"]], 0)).to(device)
        self.head_b = torch.cat([s[det + "1.bias"], s[dsc + "1.bias"]]).float().to(device)
        a, b = ops.bn_affine(s, det + "3."), ops.bn_affine(s, dsc + "3.")
        self.head_sc = torch.cat([a[0], b[0]]).to(device); self.head_sh = torch.cat([a[1], b[1]]).to(device)
        self.hc = s[det + "1.weight"].shape[0]
        self.det_w = s[det + "4.weight"].reshape(s[det + "4.weight"].shape[0], -1).float().contiguous().to(device)
        self.det_b = s[det + "4.bias"].float().to(device)
        self.det_sc, self.det_sh = [t.to(device) for t in ops.bn_affine(s, det + "5.")]
        self.desc_w = s[dsc + "4.weight"].reshape(s[dsc + "4.weight"].shape[0], -1).float().contiguous().to(device)
        self.desc_b = s[dsc + "4.bias"].float().to(device)
        self.desc_sc, self.desc_sh = [t.to(device) for t in ops.bn_affine(s, dsc + "5.")]

    def forward_raw(self, images, want_logits=False):
        x = ops.gray_to_nhwc4(images.contiguous().float())
        for L in self.layers:       # [pad(reflect), conv3x3, ReLU, BN] (+ MaxPool2d after every second conv of blocks 1-3)
            x = ops.conv3x3(x, L["w"], L["b"], L["sc"], L["sh"], 1, True, "relu")
            if L["pool"]:
                x = ops.maxpool2(x)
        enc = x
        B, Hc, Wc, _ = enc.shape
        t = ops.conv3x3(enc, self.head_w, self.head_b, self.head_sc, self.head_sh, 1, True, "relu")     # both head trunks
        t2 = t.view(B * Hc * Wc, 2 * self.hc)
        logits = ops.linear(t2, self.det_w, self.det_b, self.det_sc, self.det_sh, lda=2 * self.hc).view(B, Hc, Wc, -1)
        desc_raw = ops.linear(t2[:, self.hc:], self.desc_w, self.desc_b, self.desc_sc, self.desc_sh, lda=2 * self.hc)
        desc = ops.l2norm_rows(desc_raw, 1e-12).view(B, Hc, Wc, -1)
        prob = None if want_logits else ops.softmax_shuffle(logits, 8, 0)
        return {"prob": prob, "desc_nhwc": desc, "enc_nhwc": enc, "logits_nhwc": logits}


class SuperPointMagicLeap(torch.nn.Module):
    """Drop-in for xpoint.models.SuperPointMagicLeap: forward(data) -> {'logits','desc','prob'}; takes_pair() False."""
    _ENC = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"]

    def __init__(self, config=None):
        super().__init__()
        self.config = config or {}
        self._ref_state = collections.OrderedDict()
        self._dev = None
        self._w = None

    def takes_pair(self):
        return False

    def expected_keys(self):
        from .synth import superpoint_state_spec
        return superpoint_state_spec()

    def state_dict(self, *a, **k):
        return collections.OrderedDict(self._ref_state)

    def load_state_dict(self, state_dict, strict=True):
        self._w = None
        return _load_into(self._ref_state, self.expected_keys(), state_dict, strict)

    def to(self, device=None, *a, **k):
        return self

    def _weights(self, device):
        if self._w is None or self._dev != device:
            s = self._ref_state
            if len(s) != len(self.expected_keys()):
                raise RuntimeError("SuperPointMagicLeap: weights not loaded")
            w = {}
            for n in self._ENC + ["convPa", "convDa"]:
                w[n] = (ops.conv_w(s[n + ".weight"]).to(device), s[n + ".bias"].float().to(device))
            for n in ("convPb", "convDb"):
                w[n] = (s[n + ".weight"].reshape(s[n + ".weight"].shape[0], -1).float().contiguous().to(device), s[n + ".bias"].float().to(device))
            self._w, self._dev = w, device
        return self._w

    def forward(self, data):
        img = data["image"]
        if not img.is_cuda:
            raise RuntimeError("xpoint_amd.SuperPointMagicLeap runs on the GPU only (no CPU fallback)")
        w = self._weights(img.device)
        x = ops.gray_to_nhwc4(img.contiguous().float())
        for i, n in enumerate(self._ENC):                      # zero-padded 3x3 + ReLU, pools after 1b/2b/3b (:38-48)
            x = ops.conv3x3(x, w[n][0], w[n][1], None, None, 1, False, "relu")
            if i in (1, 3, 5):
                x = ops.maxpool2(x)
        B, Hc, Wc, C = x.shape
        cPa = ops.conv3x3(x, w["convPa"][0], w["convPa"][1], None, None, 1, False, "relu")
        semi = ops.linear(cPa.view(B * Hc * Wc, -1), w["convPb"][0], w["convPb"][1]).view(B, Hc, Wc, -1)
        cDa = ops.conv3x3(x, w["convDa"][0], w["convDa"][1], None, None, 1, False, "relu")
        desc = ops.linear(cDa.view(B * Hc * Wc, -1), w["convDb"][0], w["convDb"][1])
        desc = ops.l2norm_rows(desc, -1.0).view(B, Hc, Wc, -1)             # desc / ||desc||, no eps (:59-60)
        prob = ops.softmax_shuffle(semi, 8, 1)                              # exp(x) / (sum + 1e-5) (:73-74)
        return {"logits": ops.nchw(semi), "desc": ops.nchw(desc), "prob": prob.unsqueeze(1), "desc_nhwc": desc}


def regnet_weights(ref_state, device):
    s = ref_state
    p = "hm_regressor."
    w = {"c1": ops.conv_w(s[p + "layer1.0.weight"]).to(device), "c2": ops.conv_w(s[p + "layer1.3.weight"]).to(device)}
    w["bn1"] = [t.to(device) for t in ops.bn_affine(s, p + "layer1.1.")]
    w["bn2"] = [t.to(device) for t in ops.bn_affine(s, p + "layer1.4.")]
    # the two convolutions run on the split-fp16 engine (f32-grade, 3x the exact-f32 MFMA kernel's rate): offline two-plane split of (Co, 9 Ci)
    w["c1_h2"] = ops.split_h2(w["c1"].reshape(w["c1"].shape[0], -1)); w["c2_h2"] = ops.split_h2(w["c2"].reshape(w["c2"].shape[0], -1))
    w["fc1"] = (s[p + "fc.1.weight"].float().contiguous().to(device), s[p + "fc.1.bias"].float().to(device))
    w["fc2"] = (s[p + "fc.4.weight"].float().contiguous().to(device), s[p + "fc.4.bias"].float().to(device))
    return w


def _adaptive_pool_matrix(Hh, Wh, out_h, out_w, device):
    """(out_h * out_w, Hh * Wh) matrix of F.adaptive_avg_pool2d((out_h, out_w)) on a row-major (Hh, Wh) map: bin i covers
    [floor(i * H / out), ceil((i + 1) * H / out)) — ATen's start / end index rule; the identity when the sizes agree."""
    import math
    P = torch.zeros((out_h * out_w, Hh * Wh), dtype=torch.float32)
    for i in range(out_h):
        y0, y1 = (i * Hh) // out_h, math.ceil((i + 1) * Hh / out_h)
        for j in range(out_w):
            x0, x1 = (j * Wh) // out_w, math.ceil((j + 1) * Wh / out_w)
            wgt = 1.0 / ((y1 - y0) * (x1 - x0))
            for y in range(y0, y1):
                P[i * out_w + j, y * Wh + x0:y * Wh + x1] = wgt
    return P.to(device)


def regnet_forward(w, enc1_nhwc, enc2_nhwc, adaptive_pool=False, enc_both=None):
    """RegNet.forward (RegNet.py:32-52), eval mode.  enc (B, H', W', 48) NHWC -> (B, 8).  The cost volume's channel count
    must equal fc.1's input (256), i.e. (H'/2)*(W'/2) == 256 <=> a 256x256 image: raises otherwise, like the
    reference's Linear does.
    adaptive_pool=True (NOT reference semantics — the reference has none beyond 256x256): the pooled cost-volume vector, which is a
    (H'/2, W'/2) map over the FIRST image's positions, is adaptive-average-pooled to the 16 x 16 grid the FC layer was sized for, so the head
    accepts any image size; at 256x256 the pooling matrix is the identity and the result is the reference's (tests: g9)."""
    def layer1(x):      # conv (no bias) -> BN -> ReLU, twice, then MaxPool2d(2); BOTH images in one batch (2B crops per launch)
        # c1 on the split-fp16 engine: its input is the encoder output, which the forward's range guard (XP_STATUS_ENC: |x| < 65504) has already vetted.
        # c2 on the exact-f32 kernel: its input ReLU(BN(c1(x))) is unbounded, the split engine turns |x| >= 65504 into NaN rows and the following
        # ReLU / max-pool would map those to 0 — a silently wrong hm; the exact kernel has no range limit and the layer is 16 K rows x 64.
        x = ops.conv3x3_h2(x, w["c1_h2"], w["c1"].shape[0], None, w["bn1"][0], w["bn1"][1], 1, False, "relu_after_affine")
        x = ops.conv3x3(x, w["c2"], None, w["bn2"][0], w["bn2"][1], 1, False, "relu_after_affine")
        return ops.maxpool2(x)
    B0 = enc1_nhwc.shape[0]
    if enc_both is not None:      # the caller's single (2B, H', W', 48) encoder output (models.predict_homography): both images in one batch, no copy
        ab = layer1(enc_both.contiguous())
    else:
        ab = None
    a, b = (ab[:B0], ab[B0:]) if ab is not None else (layer1(enc1_nhwc.contiguous()), layer1(enc2_nhwc.contiguous()))
    B, Hh, Wh, C = a.shape
    hw = Hh * Wh
    n_in = w["fc1"][0].shape[1]
    if hw != n_in and not adaptive_pool:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B}x{hw} and {n_in}x{w['fc1'][0].shape[0]})")
    an = ops.l2norm_rows(a.view(B * hw, C), 1e-12).view(B, hw, C)      # F.normalize over channels
    bn = ops.l2norm_rows(b.view(B * hw, C), 1e-12).view(B, hw, C)
    # bmm(x1^T, x2) (hw, hw) then adaptive_avg_pool2d over the second image's positions (RegNet.py:50-52) = each row of x1 against the MEAN row of x2:
    # one launch for the batch, the volume is never formed (round 4; before: two GEMM launches per sample)
    v = ops.costvolume_mean(an, bn)
    if hw != n_in:
        g = int(round(n_in ** 0.5))
        if g * g != n_in:
            raise RuntimeError(f"adaptive_pool: fc.1 input {n_in} is not a square grid")
        key = ("pool", Hh, Wh, str(a.device))
        if key not in w:
            w[key] = _adaptive_pool_matrix(Hh, Wh, g, g, a.device)
        v = ops.linear(v, w[key])                                      # (B, hw) x (n_in, hw)^T
    h = ops.linear(v, w["fc1"][0], w["fc1"][1], act="relu")            # Dropout = identity in eval
    return ops.linear(h, w["fc2"][0], w["fc2"][1])
