"""GPU parity of the BASELINE config-1 backbones (conv-encoder XPoint, SuperPointMagicLeap) and the RegNet
homography head (config 5, valid at 256x256 only) against goldens produced by the REAL reference."""
import numpy as np
import pytest
import torch

from oracle import xpoint_oracle as xo
from xpoint_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def test_conv_xpoint_vs_reference(gpu_lib, golden):
    from xpoint_amd import models
    g = golden("g12_conv_xpoint.npz")
    cfg = synth.multipoint_config()
    net = models.XPoint(cfg)
    net.load_state_dict(_t(synth.make_conv_xpoint_state_dict(cfg)), strict=True)
    net.to("cuda").eval()
    assert net.takes_pair() is False
    img = torch.from_numpy(synth.make_image(0, "optical", 64, 96)[None]).cuda()
    with torch.no_grad():
        r = net({"image": img, "is_optical": torch.ones(1, 1, dtype=torch.bool)})
    for k in ("prob", "desc", "encoder_output"):
        assert r[k].shape == g[f"64x96/{k}"].shape
        assert float(np.abs(r[k].cpu().numpy() - g[f"64x96/{k}"]).max()) < TOL, k
    img = torch.from_numpy(synth.make_image(0, "optical", 240, 320)[None]).cuda()      # BASELINE config 1 size
    with torch.no_grad():
        r = net({"image": img})
    assert float(np.abs(r["prob"].cpu().numpy() - g["240x320/prob"]).max()) < TOL
    assert abs(float(r["desc"].double().abs().sum()) - g["240x320/desc_sum"][1]) < 1e-5 * g["240x320/desc_sum"][1]


def test_config1_pair_flow_conv_backbone(gpu_lib):
    """BASELINE configs[0]: conv backbone, one synthetic 240x320 optical/thermal pair through the reference's
    non-pair call sequence (net(data['optical']), net(data['thermal'])) + NMS + sampling + matching."""
    from xpoint_amd import models
    from xpoint_amd.predict import predict_align_image_pair
    cfg = synth.multipoint_config()
    sd = _t(synth.make_conv_xpoint_state_dict(cfg))
    net = models.XPoint(cfg); net.load_state_dict(sd); net.to("cuda").eval()
    H, W = 240, 320
    data = synth.to_torch(synth.make_pair_batch(0, 1, H, W), "cuda")
    with torch.no_grad():
        o, t, res = predict_align_image_pair(net, data)
        # oracle on the same inputs: forward, NMS, keypoints, descriptors, exact matches
        dc = synth.to_torch(synth.make_pair_batch(0, 1, H, W))
        oo = xo.forward_impl(dc["optical"]["image"], sd); ot = xo.forward_impl(dc["thermal"]["image"], sd)
    po = xo.box_nms(oo["prob"] * dc["optical"]["valid_mask"], 8, 0.015)
    ko = xo.extract_keypoints(po[0, 0], 0.015)
    r = res[0]
    mine, ref = {tuple(x) for x in r["kp_optical"].cpu().tolist()}, {tuple(x) for x in ko.tolist()}
    assert len(mine ^ ref) <= max(2, len(ref) // 100), (len(mine), len(ref))
    oms = xo.get_matches(r["desc_optical"].cpu().numpy(), r["desc_thermal"].cpu().numpy())
    assert [(m.queryIdx, m.trainIdx) for m in r["matches"]] == [(m.queryIdx, m.trainIdx) for m in oms]
    assert r["desc_optical"].shape[1] == 64


def test_superpoint_vs_reference(gpu_lib, golden):
    from xpoint_amd import models
    g = golden("g11_superpoint.npz")
    net = models.SuperPointMagicLeap()
    net.load_state_dict(_t(synth.make_superpoint_state_dict()), strict=True)
    net.to("cuda").eval()
    assert net.takes_pair() is False
    img = torch.from_numpy(synth.make_image(0, "optical", 64, 96)[None]).cuda()
    with torch.no_grad():
        r = net({"image": img})
    for k in ("logits", "desc", "prob"):
        assert r[k].shape == g[f"64x96/{k}"].shape, k
        err = float(np.abs(r[k].cpu().numpy() - g[f"64x96/{k}"]).max())
        assert err < (5e-4 if k == "logits" else TOL), (k, err)     # logits carry the x4 detector gain
    img = torch.from_numpy(synth.make_image(0, "optical", 240, 320)[None]).cuda()
    with torch.no_grad():
        r = net({"image": img})
    assert float(np.abs(r["prob"].cpu().numpy() - g["240x320/prob"]).max()) < TOL
    with pytest.raises(RuntimeError):
        net({"image": torch.zeros(1, 1, 64, 96)})


def test_regnet_head_256(gpu_lib, golden):
    """RegNet head: stage-wise (reference encoder outputs in) and end to end through XPoint.forward at 256x256."""
    from xpoint_amd import models
    from xpoint_amd.convmodels import regnet_forward, regnet_weights
    g = golden("g9_regnet.npz")
    cfg = synth.xpoint_exp1_config(256, 256, hm_head=True)
    sd = synth.make_torch_state_dict(cfg)
    w = regnet_weights(sd, torch.device("cuda"))
    e1 = torch.from_numpy(g["enc_optical"]).permute(0, 2, 3, 1).contiguous().cuda()
    e2 = torch.from_numpy(g["enc_thermal"]).permute(0, 2, 3, 1).contiguous().cuda()
    hm = regnet_forward(w, e1, e2)
    assert hm.shape == (1, 8) and float(np.abs(hm.cpu().numpy() - g["hm"]).max()) < TOL
    net = models.XPoint(cfg); net.load_state_dict(sd, strict=True); net.to("cuda").eval()
    with torch.no_grad():
        o, t, hm2 = net(synth.to_torch(synth.make_pair_batch(7, 1, 256, 256), "cuda"))
    assert float(np.abs(hm2.cpu().numpy() - g["hm"]).max()) < 5e-4           # includes the encoder's own 1e-5-level error
    with torch.no_grad():                                                      # the head alone (encoder + RegNet, no detector / descriptor heads)
        d = synth.to_torch(synth.make_pair_batch(7, 1, 256, 256), "cuda")
        hm3 = net.predict_homography(d["optical"]["image"], d["thermal"]["image"])
    assert torch.equal(hm3, hm2)
    with pytest.raises(RuntimeError):                                          # 480x640: the reference fails too
        regnet_forward(w, torch.zeros(1, 60, 80, 48, device="cuda"), torch.zeros(1, 60, 80, 48, device="cuda"))


def test_regnet_head_large_activations_stay_finite(gpu_lib, golden):
    """the head's second convolution sees ReLU(BN(c1(x))), which is unbounded.  With BatchNorm scales that push it past fp16's range (65504) the
    split-fp16 engine would turn the rows into NaN and the following ReLU / max-pool would hide them as zeros; the layer runs on the exact-f32 kernel and the
    result equals a float64 torch evaluation of RegNet.forward (RegNet.py:32-52) on the same weights."""
    import torch.nn.functional as F
    from xpoint_amd.convmodels import regnet_forward, regnet_weights
    g = golden("g9_regnet.npz")
    cfg = synth.xpoint_exp1_config(256, 256, hm_head=True)
    sd = {k: v.clone() for k, v in synth.make_torch_state_dict(cfg).items()}
    p = [k for k in sd if k.endswith("layer1.1.weight")][0][:-len("layer1.1.weight")]
    sd[p + "layer1.1.weight"] = sd[p + "layer1.1.weight"] * 3.0e5            # BN1 gamma: c1's normalised output x 3e5 -> far beyond 65504
    w = regnet_weights(sd, torch.device("cuda"))
    e1 = torch.from_numpy(g["enc_optical"]).permute(0, 2, 3, 1).contiguous().cuda()
    e2 = torch.from_numpy(g["enc_thermal"]).permute(0, 2, 3, 1).contiguous().cuda()
    hm = regnet_forward(w, e1, e2).cpu().double()

    def bn(x, i):
        wt, b, m, v = (sd[p + f"layer1.{i}.{n}"].double() for n in ("weight", "bias", "running_mean", "running_var"))
        return (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + 1e-5) * wt[None, :, None, None] + b[None, :, None, None]

    def layer1(x):
        x = F.relu(bn(F.conv2d(x, sd[p + "layer1.0.weight"].double(), padding=1), 1))
        assert float(x.abs().max()) > 65504.0                                  # the case is what it claims to be
        x = F.relu(bn(F.conv2d(x, sd[p + "layer1.3.weight"].double(), padding=1), 4))
        return F.max_pool2d(x, 2)
    a, b = layer1(torch.from_numpy(g["enc_optical"]).double()), layer1(torch.from_numpy(g["enc_thermal"]).double())
    B, C, Hh, Wh = a.shape
    an, bn_ = F.normalize(a, dim=1).view(B, C, -1), F.normalize(b, dim=1).view(B, C, -1)
    v = torch.bmm(an.transpose(1, 2), bn_).mean(dim=2)                         # cost volume + global average pool over the second image's positions
    h = F.relu(F.linear(v, sd[p + "fc.1.weight"].double(), sd[p + "fc.1.bias"].double()))
    ref = F.linear(h, sd[p + "fc.4.weight"].double(), sd[p + "fc.4.bias"].double())
    assert bool(torch.isfinite(hm).all())
    assert float((hm - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), float((hm - ref).abs().max())


def test_regnet_adaptive_pool_generalisation(gpu_lib, golden):
    """Opt-in generalisation of the RegNet head beyond 256x256 (NOT reference semantics — the reference's head only accepts
    256x256, RegNet.py:38-52): the pooled cost-volume map is adaptive-average-pooled to the 16x16 grid of the FC layer.  (1) At 256x256 it
    IS the reference (g9, bit-identical to the default path); (2) at 480x640 and 224x320 it equals a torch restatement — RegNet.forward with
    F.adaptive_avg_pool2d on the (H'/2, W'/2) map — of the same weights; (3) models.XPoint.forward carries it (regnet_adaptive_pool)."""
    import torch.nn.functional as F
    from xpoint_amd import models
    from xpoint_amd.convmodels import regnet_forward, regnet_weights
    g = golden("g9_regnet.npz")
    cfg = synth.xpoint_exp1_config(256, 256, hm_head=True)
    sd = synth.make_torch_state_dict(cfg)
    w = regnet_weights(sd, torch.device("cuda"))
    e1 = torch.from_numpy(g["enc_optical"]).permute(0, 2, 3, 1).contiguous().cuda()
    e2 = torch.from_numpy(g["enc_thermal"]).permute(0, 2, 3, 1).contiguous().cuda()
    assert torch.equal(regnet_forward(w, e1, e2, adaptive_pool=True), regnet_forward(w, e1, e2))
    assert float(np.abs(regnet_forward(w, e1, e2, adaptive_pool=True).cpu().numpy() - g["hm"]).max()) < TOL

    def torch_head(x1, x2):          # RegNet.forward in fp64 with the pooling step added (x: (B, 48, H', W'))
        p = "hm_regressor."
        def layer1(x):
            for c, b in (("layer1.0", "layer1.1"), ("layer1.3", "layer1.4")):
                x = F.conv2d(x, sd[p + c + ".weight"].double(), None, padding=1)
                x = F.batch_norm(x, sd[p + b + ".running_mean"].double(), sd[p + b + ".running_var"].double(), sd[p + b + ".weight"].double(),
                                 sd[p + b + ".bias"].double(), False, 0.0, 1e-5)
                x = F.relu(x)
            return F.max_pool2d(x, 2, 2)
        a, b = layer1(x1), layer1(x2)
        N, C, Hh, Wh = a.shape
        cv = torch.bmm(F.normalize(a).reshape(N, C, Hh * Wh).transpose(1, 2), F.normalize(b).reshape(N, C, Hh * Wh)).reshape(N, Hh * Wh, Hh, Wh)
        v = F.adaptive_avg_pool2d(cv, (1, 1)).view(N, 1, Hh, Wh)
        v = F.adaptive_avg_pool2d(v, (16, 16)).reshape(N, 256)
        h = F.relu(F.linear(v, sd[p + "fc.1.weight"].double(), sd[p + "fc.1.bias"].double()))
        return F.linear(h, sd[p + "fc.4.weight"].double(), sd[p + "fc.4.bias"].double())
    for (Hc, Wc) in ((60, 80), (28, 40)):
        x1 = torch.from_numpy(synth.uniform(f"rg/a{Hc}", (2, 48, Hc, Wc), -1.0, 1.0)); x2 = torch.from_numpy(synth.uniform(f"rg/b{Hc}", (2, 48, Hc, Wc), -1.0, 1.0))
        ref = torch_head(x1.double(), x2.double())
        got = regnet_forward(w, x1.permute(0, 2, 3, 1).contiguous().cuda(), x2.permute(0, 2, 3, 1).contiguous().cuda(), adaptive_pool=True)
        assert got.shape == (2, 8) and float((got.cpu().double() - ref).abs().max()) < 2e-5
    cfg2 = synth.xpoint_exp1_config(224, 320, hm_head=True)
    net = models.XPoint(cfg2); net.load_state_dict(synth.make_torch_state_dict(cfg2), strict=True); net.to("cuda").eval()
    d = synth.to_torch(synth.make_pair_batch(2, 1, 224, 320), "cuda")
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            net(d)                                           # default: the reference's behaviour (shape error)
        net.regnet_adaptive_pool = True
        _, _, hm = net(d)
    assert hm.shape == (1, 8) and bool(torch.isfinite(hm).all())


def test_costvolume_mean_equals_bmm_then_pool(gpu_lib):
    """xp_costvolume_mean: v[n][p] = a[n][p] . mean_q b[n][q]  ==  adaptive_avg_pool2d(bmm(x1^T, x2).view(N, hw, H', W'), 1) of the reference
    (RegNet.py:44-52), evaluated here in float64 from the same unit rows; ragged sizes (hw not a multiple of the wave count, C not a multiple of 64)."""
    from xpoint_amd import nnops as ops
    for (B, hw, C) in [(3, 256, 192), (2, 77, 100), (1, 5, 64)]:
        g = torch.Generator().manual_seed(B * 1000 + hw)
        a = torch.nn.functional.normalize(torch.randn(B, hw, C, generator=g), dim=2)
        b = torch.nn.functional.normalize(torch.randn(B, hw, C, generator=g), dim=2)
        ref = torch.bmm(a.double(), b.double().transpose(1, 2)).mean(dim=2)          # (B, hw): mean over the second image's positions
        got = ops.costvolume_mean(a.cuda().contiguous(), b.cuda().contiguous()).cpu().double()
        assert float((got - ref).abs().max()) < 2e-7, (B, hw, C, float((got - ref).abs().max()))
