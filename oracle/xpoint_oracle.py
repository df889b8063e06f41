"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (torch fp32 functional ops + the plain-C kernels in oracle/csrc) of the XPoint
inference hot path, written from the reference's behaviour; each function cites the reference
file:line it follows.  It is NOT the product: only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg import it, and only as the checker / the CPU number.

Parity status: PINNED for everything except the two third-party boundaries.
  * pinned against the real reference imported in the build container
    (oracle/refharness/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py):
    selective scan, cross scan/merge, SS2D, VSS block, encoder, heads, full forward,
    interpolate_descriptors, NNMatcher-mode matching, RegNet head, SuperPointMagicLeap.
  * "parity unpinned": torchvision.ops.nms and cv2.BFMatcher are absent from /root/reference and
    from this image; box_nms / crossCheck matching restate their documented
    algorithms (see oracle/csrc/oracle_kernels.c).
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        _lib.xo_box_nms.restype = ctypes.c_int64
        _lib.xo_threshold_pairs.restype = ctypes.c_int64
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if torch.is_tensor(t) else t.ctypes.data)


# ------------------------------------------------------------------------------------------
# selective scan                                   reference csms6s.py:25-68 (selective_scan_torch)
# ------------------------------------------------------------------------------------------

def selective_scan(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True, return_last_state=False):
    """u, delta (B, K*C, L); A (K*C, N); B, C (B, K, N, L); D, delta_bias (K*C).
    Returns out (B, K*C, L) float32 [, last_state (B, K*C, N)]."""
    Batch, K, N, L = B.shape
    KC = u.shape[1]
    assert KC % K == 0 and delta.shape == u.shape and A.shape == (KC, N) and C.shape == B.shape
    if delta_bias is not None:
        delta = delta + delta_bias[..., None]                     # csms6s.py:47-48
    if delta_softplus:
        delta = F.softplus(delta)                                 # csms6s.py:49-50
    u, delta, A, B, C = u.float(), delta.float(), A.float(), B.float(), C.float()
    Bx = B.view(Batch, K, 1, N, L).expand(Batch, K, KC // K, N, L).reshape(Batch, KC, N, L)
    deltaA = torch.exp(torch.einsum('bdl,dn->bdln', delta, A)).contiguous()          # :55
    deltaB_u = torch.einsum('bdl,bdnl,bdl->bdln', delta, Bx, u).contiguous()          # :56
    y = torch.empty((Batch, KC, L), dtype=torch.float32)
    last = torch.empty((Batch, KC, N), dtype=torch.float32)
    Cc = C.contiguous()
    lib().xo_scan_recurrence(_p(deltaA), _p(deltaB_u), _p(Cc), _p(y), _p(last),
                             ctypes.c_int64(Batch), ctypes.c_int64(KC), ctypes.c_int64(L),
                             ctypes.c_int64(N), ctypes.c_int64(K))               # :58-65
    out = y if D is None else y + u * D.float().unsqueeze(-1)                     # :67
    return (out, last) if return_last_state else out


# ------------------------------------------------------------------------------------------
# cross scan / merge                                            reference csm_triton.py:22-85
# ------------------------------------------------------------------------------------------

def cross_scan(x):
    B, C, H, W = x.shape
    y = x.new_empty((B, 4, C, H * W))
    y[:, 0] = x.flatten(2, 3)                                     # row-major
    y[:, 1] = x.transpose(2, 3).flatten(2, 3)                     # column-major
    y[:, 2:4] = torch.flip(y[:, 0:2], dims=[-1])                  # both reversed
    return y


def cross_merge(ys):
    B, K, D, H, W = ys.shape
    y = ys.reshape(B, K, D, -1)
    y = y[:, 0:2] + y[:, 2:4].flip(dims=[-1]).view(B, 2, D, -1)   # csm_triton.py:60
    y = y[:, 0] + y[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, D, -1)  # :61
    return y                                                      # (B, D, H*W)


def _route_pixels(k, scans, H, W):
    """pixel index (h W + w) visited at position l of route k — csm_triton.py:22-53: scans 0: k = 0 row-major, 1 column-major, 2 / 3 their
    reverses; scans 1: four copies of route 0; scans 2: 0, 1 forward, 2, 3 reversed."""
    import numpy as np
    L = H * W
    transposed = scans == 0 and (k & 1)
    flipped = scans != 1 and (k >> 1)
    l = np.arange(L)
    lp = L - 1 - l if flipped else l
    return ((lp % H) * W + lp // H) if transposed else lp


def cross_scan_op(x, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0):
    """Index-table restatement of reference cross_scan_fn (csm_triton.py:501-507 -> cross_scan_fwd :22-53 / cross_scan1b1_fwd :88-131) for every
    layout: x (B,C,H,W) | (B,H,W,C) | (B,4,C,H,W) | (B,H,W,4,C) -> (B,4,C,L) | (B,L,4,C).  A pure permutation: any dtype."""
    import numpy as np
    if one_by_one:
        xc = x if in_channel_first else x.permute(0, 3, 4, 1, 2)           # (B,4,C,H,W)
        B, _, C, H, W = xc.shape
        flat = xc.reshape(B, 4, C, H * W)
        y = torch.stack([flat[:, k][..., torch.from_numpy(_route_pixels(k, scans, H, W))] for k in range(4)], dim=1)
    else:
        xc = x if in_channel_first else x.permute(0, 3, 1, 2)              # (B,C,H,W)
        B, C, H, W = xc.shape
        flat = xc.reshape(B, C, H * W)
        y = torch.stack([flat[..., torch.from_numpy(_route_pixels(k, scans, H, W))] for k in range(4)], dim=1)
    return y.contiguous() if out_channel_first else y.permute(0, 3, 1, 2).contiguous()


def cross_merge_op(ys, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0):
    """Restatement of reference cross_merge_fn (csm_triton.py:511-517 -> cross_merge_fwd :56-85 / cross_merge1b1_fwd :134-180): ys (B,4,C,H,W) if
    out_channel_first else (B,H,W,4,C); result in the scan's IN layout.  Adds in the tensor's dtype, associated as the reference does:
    (y0 + y2) + (y1 + y3) for scans 0 / 2 (every add rounded to the dtype), ((y0 + y1) + y2) + y3 for scans 1 (`y.sum(1)`: float32 accumulator, one rounding)."""
    import numpy as np
    yc = ys if out_channel_first else ys.permute(0, 3, 4, 1, 2)            # (B,4,C,H,W)
    B, _, C, H, W = yc.shape
    flat = yc.reshape(B, 4, C, H * W)
    back = []
    for k in range(4):
        pix = _route_pixels(k, scans, H, W)
        inv = np.empty_like(pix); inv[pix] = np.arange(H * W)              # position of every pixel in route k
        back.append(flat[:, k][..., torch.from_numpy(inv)])
    if one_by_one:
        out = torch.stack(back, dim=1)                                     # (B,4,C,L)
        return out.contiguous() if in_channel_first else out.permute(0, 3, 1, 2).contiguous()
    if scans == 1:      # torch's sum accumulates 16-bit inputs in float32 and rounds once
        f = [b.float() for b in back]
        out = (((f[0] + f[1]) + f[2]) + f[3]).to(ys.dtype)
    else:
        out = (back[0] + back[2]) + (back[1] + back[3])
    return out.contiguous() if in_channel_first else out.permute(0, 2, 1).contiguous()


# ------------------------------------------------------------------------------------------
# SS2D (forward_type v05_noz)                          reference VMamba.py:493-664
# ------------------------------------------------------------------------------------------

# ---------------------------------------------------------------------------------------------------------------------
# Precision classes of the dense layers.  DENSE_PRODUCTS = 6 (default) is the fp32 reference arithmetic and the only
# class pinned against the reference goldens.  3 and 1 restate what the HIP path computes under
# xp_set_dense_products(3 / 1): every operand of a dense layer (Linear, non-depthwise conv, x_proj) replaced by the sum
# of its first two / its first bf16 plane(s) (x0 = bf16(x), x1 = bf16(x - x0)) and only the products a0 b0 [+ a0 b1 + a1 b0]
# formed — 1 = bf16 operands with wide accumulation, the arithmetic class of the reference's `mixed_precision` autocast
# matmuls (XPoint.py:182).  The stem conv, the depthwise conv and the dt projection stay fp32 there, as in the HIP path.
# ---------------------------------------------------------------------------------------------------------------------
DENSE_PRODUCTS = 6

# AMP16 = True: the reference's mixed-precision recipe (XPoint.py:182 autocast; pinned by tests/golden/g20 = the real reference under float16 CPU
# autocast) restated on f32 tensors: every convolution / linear layer takes its input, weight and bias rounded to fp16 and returns a value rounded
# to fp16; LayerNorm, GELU, SiLU, eval BatchNorm and the residual adds return fp16-rounded values (half in -> half out; statistics and arithmetic
# in f32, which is what PyTorch's CPU kernels do internally); the dt projection (a half conv1d, VMamba.py:608) is rounded BEFORE the f32 delta_bias
# is added inside the scan (csms6s.py:47-50); the scan, cross merge and out_norm run in f32 and forward_corev2 returns y.to(x.dtype) (VMamba.py:646);
# softmax / normalize run in f32 on the heads' `.to(torch.float)` (XPoint.py:349,363).
AMP16 = False


def _h(x):
    """fp16 rounding of a value kept in f32 (identity unless AMP16)."""
    return x.to(torch.float16).to(torch.float32) if (AMP16 and x is not None) else x


def _planes(x, n):
    out, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def _dense(op, x, w, b, **kw):
    if AMP16:
        return _h(op(_h(x), _h(w), _h(b), **kw))
    if DENSE_PRODUCTS == 6:
        return op(x, w, b, **kw)
    terms = [(0, 0)] if DENSE_PRODUCTS == 1 else [(1, 0), (0, 1), (0, 0)]
    xs, ws = _planes(x, 2), _planes(w, 2)
    acc = None
    for i, j in terms:
        y = op(xs[i].double(), ws[j].double(), None, **kw)
        acc = y if acc is None else acc + y
    acc = acc.float()
    if b is not None:
        acc = acc + b.view([1, -1] + [1] * (acc.dim() - 2)) if op is not F.linear else acc + b
    return acc


def _lin(x, w, b=None):
    return _dense(F.linear, x, w, b)


def _conv(x, w, b=None, **kw):
    return _dense(F.conv2d, x, w, b, **kw)


def ss2d_core(x, sd, pre, return_parts=False):
    """x (B, C, H, W) after dwconv+SiLU -> (B, H, W, C) after out_norm.  VMamba.py:601-646."""
    B, D, H, W = x.shape
    L = H * W
    xw = sd[pre + "x_proj_weight"]                                # (4, R+2N, C)
    K, RN, _ = xw.shape
    dtw = sd[pre + "dt_projs_weight"]                             # (4, C, R)
    R = dtw.shape[2]
    N = (RN - R) // 2
    xs = cross_scan(x)
    x_dbl = _dense(F.conv1d, xs.view(B, -1, L), xw.reshape(-1, D, 1), None, groups=K)           # :605
    dts, Bs, Cs = torch.split(x_dbl.view(B, K, -1, L), [R, N, N], dim=2)                      # :606
    dts = _h(F.conv1d(dts.contiguous().view(B, -1, L), _h(dtw).reshape(K * D, -1, 1), groups=K))      # :608
    xs = xs.view(B, -1, L)
    As = -sd[pre + "A_logs"].float().exp()                                                    # :619
    Ds = sd[pre + "Ds"].float()
    delta_bias = sd[pre + "dt_projs_bias"].view(-1).float()
    ys = selective_scan(xs, dts.contiguous().view(B, -1, L), As, Bs.contiguous().view(B, K, N, L),
                        Cs.contiguous().view(B, K, N, L), Ds, delta_bias, True).view(B, K, -1, H, W)  # :628
    y = cross_merge(ys).view(B, -1, H, W)                                                     # :632
    y = y.view(B, -1, H * W).transpose(1, 2).contiguous().view(B, H, W, -1)                   # :642
    yn = F.layer_norm(y, (D,), sd[pre + "out_norm.weight"], sd[pre + "out_norm.bias"], 1e-5)  # :644
    if return_parts:
        return yn, dict(xs=xs, dts=dts, Bs=Bs, Cs=Cs, ys=ys, y_merged=y)
    return _h(yn)                                                                             # :646 y.to(x.dtype)


def ss2d(x, sd, pre):
    """x (B,H,W,C) -> (B,H,W,C).  VMamba.py:648-664 with disable_z (noz)."""
    t = _lin(x, sd[pre + "in_proj.weight"])                       # :649 (no bias)
    t = t.permute(0, 3, 1, 2).contiguous()                        # :654-655
    t = _h(F.conv2d(t, _h(sd[pre + "conv2d.weight"]), None, padding=1, groups=t.shape[1]))   # :657
    t = _h(F.silu(t))                                             # :658
    y = ss2d_core(t, sd, pre)                                     # :659
    return _lin(y, sd[pre + "out_proj.weight"])                   # :663


def vss_block(x, sd, pre):
    """VMamba.py:1222-1234 (post_norm False, DropPath identity in eval) + Mlp :110-128."""
    C = x.shape[-1]
    x = _h(x + ss2d(_h(F.layer_norm(x, (C,), sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5)), sd, pre + "op."))
    h = _h(F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5))
    h = _lin(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    h = _h(F.gelu(h))                                             # exact erf GELU
    h = _lin(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return _h(x + h)


def patch_embed(img, sd, pre):
    """VMamba.py:1405-1420 (_make_patch_embed_v2), after the gray->3ch cat of :1509-1510."""
    x = torch.cat((img, img, img), dim=1) if img.shape[1] == 1 else img
    x = _h(F.conv2d(_h(x), _h(sd[pre + "0.weight"]), _h(sd[pre + "0.bias"]), stride=2, padding=1))
    x = x.permute(0, 2, 3, 1)
    x = _h(F.layer_norm(x, (x.shape[-1],), sd[pre + "2.weight"], sd[pre + "2.bias"], 1e-5))
    x = _h(F.gelu(x.permute(0, 3, 1, 2)))
    x = _conv(x, sd[pre + "5.weight"], sd[pre + "5.bias"], stride=2, padding=1)
    x = x.permute(0, 2, 3, 1)
    return _h(F.layer_norm(x, (x.shape[-1],), sd[pre + "7.weight"], sd[pre + "7.bias"], 1e-5))


def downsample(x, sd, pre):
    """VMamba.py:1432-1440 (_make_downsample_v3)."""
    x = _conv(x.permute(0, 3, 1, 2), sd[pre + "1.weight"], sd[pre + "1.bias"], stride=2, padding=1)
    x = x.permute(0, 2, 3, 1)
    return _h(F.layer_norm(x, (x.shape[-1],), sd[pre + "3.weight"], sd[pre + "3.bias"], 1e-5))


def depth_to_space(x, bs):
    """VMamba.py:1500-1505: out[n,c,bs*h+i,bs*w+j] = x[n,(bs*i+j)*C'+c,h,w]."""
    N, C, H, W = x.shape
    x = x.view(N, bs, bs, C // (bs * bs), H, W).permute(0, 3, 4, 1, 5, 2).contiguous()
    return x.view(N, C // (bs * bs), H * bs, W * bs)


def _depths(sd, pre="encoder."):
    depths = []
    s = 0
    while f"{pre}layers.{s}.blocks.0.norm.weight" in sd:
        j = 0
        while f"{pre}layers.{s}.blocks.{j}.norm.weight" in sd:
            j += 1
        depths.append(j)
        s += 1
    return depths


def vssm_forward(img, sd, pre="encoder.", taps: Optional[dict] = None):
    """VMamba.py:1507-1525.  img (B,1,H,W) -> (B, E/2, H/8, W/8)."""
    x = patch_embed(img, sd, pre + "patch_embed.")
    if taps is not None:
        taps["patch_embed"] = x
    depths = _depths(sd, pre)
    for s, depth in enumerate(depths):
        for j in range(depth):
            x = vss_block(x, sd, f"{pre}layers.{s}.blocks.{j}.")
            if taps is not None:
                taps[f"block{s}.{j}"] = x
        if s < len(depths) - 1:
            x = downsample(x, sd, f"{pre}layers.{s}.downsample.")
    x = x.permute(0, 3, 1, 2)
    return depth_to_space(x, 4)


# ------------------------------------------------------------------------------------------
# heads                                                   reference XPoint.py:112-138,348-371
# ------------------------------------------------------------------------------------------

def _bn(x, sd, pre):
    return _h(F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                           False, 0.0, 1e-5))


def _head_trunk(x, sd, pre):
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")                    # ReflectionPad2d(1)
    x = _conv(x, sd[pre + "1.weight"], sd[pre + "1.bias"])
    x = _bn(F.relu(x), sd, pre + "3.")                            # bn_first False: ReLU then BN
    x = _conv(x, sd[pre + "4.weight"], sd[pre + "4.bias"])
    return _bn(x, sd, pre + "5.")


def detector_head(x, sd, return_logits=False):
    logits = _head_trunk(x, sd, "detector_head_convolutions.")
    if return_logits:
        return None, logits
    prob = F.softmax(logits, dim=1)                               # Softmax2d
    r = int(round(math.sqrt(logits.shape[1] - 1)))
    return F.pixel_shuffle(prob[:, :-1], r), None                 # XPoint.py:357-358


def descriptor_head(x, sd):
    d = _head_trunk(x, sd, "descriptor_head_convolutions.")
    return F.normalize(d, p=2, dim=1)                             # XPoint.py:365-366


def conv_encoder_forward(image, sd, pre="encoder."):
    """XPoint.py:331-336,451-466: 4 x [pad, conv3x3, ReLU, BN] x 2 with 3 max-pools (channel_version 0)."""
    x = image
    for li, idx in enumerate([1, 5, 10, 14, 19, 23, 28, 32]):
        x = F.pad(x, (1, 1, 1, 1), mode="reflect")
        x = F.relu(F.conv2d(x, sd[f"{pre}{idx}.weight"], sd[f"{pre}{idx}.bias"]))
        x = _bn(x, sd, f"{pre}{idx + 2}.")
        if li in (1, 3, 5):
            x = F.max_pool2d(x, 2, 2)
    return x


def forward_impl(image, sd, force_return_logits=False):
    """XPoint.py:283-323 with multispectral False (VMamba encoder, or the conv encoder when the state dict has one)."""
    enc = conv_encoder_forward(image, sd) if "encoder.1.weight" in sd else vssm_forward(image, sd)
    prob, logits = detector_head(enc, sd, force_return_logits)
    return {"prob": prob, "logits": logits, "desc": descriptor_head(enc, sd), "encoder_output": enc}


def regnet_forward(x1, x2, sd, pre="hm_regressor."):
    """RegNet.py:20-52 (eval: Dropout identity)."""
    def layer1(x):
        x = F.relu(_bn(F.conv2d(x, sd[pre + "layer1.0.weight"], None, padding=1), sd, pre + "layer1.1."))
        x = F.relu(_bn(F.conv2d(x, sd[pre + "layer1.3.weight"], None, padding=1), sd, pre + "layer1.4."))
        return F.max_pool2d(x, 2, 2)
    a, b = layer1(x1), layer1(x2)
    N, C, H, W = a.shape
    a = F.normalize(a).reshape(N, C, H * W)
    b = F.normalize(b).reshape(N, C, H * W)
    cv = torch.bmm(a.transpose(1, 2), b).reshape(N, H * W, H, W)
    v = F.adaptive_avg_pool2d(cv, (1, 1)).view(N, H * W)
    v = F.relu(F.linear(v, sd[pre + "fc.1.weight"], sd[pre + "fc.1.bias"]))
    return F.linear(v, sd[pre + "fc.4.weight"], sd[pre + "fc.4.bias"])


def xpoint_forward(data, sd, hm_head=False):
    """XPoint.py:181-214 (takes_pair True, fp32: autocast is a no-op on CPU)."""
    o = forward_impl(data["optical"]["image"], sd)
    t = forward_impl(data["thermal"]["image"], sd)
    hm = regnet_forward(o["encoder_output"], t["encoder_output"], sd) if hm_head else None
    return o, t, hm


# ------------------------------------------------------------------------------------------
# SuperPointMagicLeap                                 reference SuperPointMagicLeap.py:31-86
# ------------------------------------------------------------------------------------------

def superpoint_forward(image, sd):
    x = image
    for n in ("1a", "1b"):
        x = F.relu(F.conv2d(x, sd[f"conv{n}.weight"], sd[f"conv{n}.bias"], padding=1))
    x = F.max_pool2d(x, 2, 2)
    for n in ("2a", "2b"):
        x = F.relu(F.conv2d(x, sd[f"conv{n}.weight"], sd[f"conv{n}.bias"], padding=1))
    x = F.max_pool2d(x, 2, 2)
    for n in ("3a", "3b"):
        x = F.relu(F.conv2d(x, sd[f"conv{n}.weight"], sd[f"conv{n}.bias"], padding=1))
    x = F.max_pool2d(x, 2, 2)
    for n in ("4a", "4b"):
        x = F.relu(F.conv2d(x, sd[f"conv{n}.weight"], sd[f"conv{n}.bias"], padding=1))
    cPa = F.relu(F.conv2d(x, sd["convPa.weight"], sd["convPa.bias"], padding=1))
    semi = F.conv2d(cPa, sd["convPb.weight"], sd["convPb.bias"])
    cDa = F.relu(F.conv2d(x, sd["convDa.weight"], sd["convDa.bias"], padding=1))
    desc = F.conv2d(cDa, sd["convDb.weight"], sd["convDb.bias"])
    desc = desc.div(torch.unsqueeze(torch.norm(desc, p=2, dim=1), 1))     # no eps (:59-60)
    dense = torch.exp(semi)                                                # no max-subtraction (:73)
    dense = dense / (dense.sum(dim=1, keepdim=True) + 0.00001)
    prob = F.pixel_shuffle(dense[:, :-1], 8)                               # same cell order (:78-84)
    return {"logits": semi, "desc": desc, "prob": prob}


# ------------------------------------------------------------------------------------------
# post-processing                                        reference utils/utils.py:148-238
# ------------------------------------------------------------------------------------------

def box_nms(prob, size, min_prob, iou=0.1, keep_top_k=0):
    """prob (H,W) or (B,1,H,W) -> same shape.  utils.py:148-192."""
    if prob.dim() not in (2, 4):
        raise ValueError('The probability must be either 2D (H,W), or 4D (B, 1, H, W)')
    p = prob.detach().float().contiguous()
    H, W = p.shape[-2:]
    flat = p.view(-1, H, W)
    out = torch.zeros_like(flat)
    for b in range(flat.shape[0]):
        lib().xo_box_nms(_p(flat[b]), _p(out[b]), None, ctypes.c_int64(H), ctypes.c_int64(W),
                         ctypes.c_float(size), ctypes.c_float(min_prob), ctypes.c_float(iou),
                         ctypes.c_int64(keep_top_k))
    return out.view_as(p)


def extract_keypoints(prob_hw, thr, mask_hw=None):
    """predict_align_image_pair.py:242-243 / predict_keypoints.py:213-215: (N,2) int64 (y,x) row-major."""
    m = (prob_hw > thr).float()
    if mask_hw is not None:
        m = m * mask_hw.float()
    return torch.nonzero(m)


def interpolate_descriptors(keypoints, desc_lowres, H, W):
    """utils.py:229-238.  keypoints (N,2) (y,x); desc_lowres (C,Hc,Wc) -> (N,C)."""
    kp = keypoints.float().clone()
    kp[:, 0] = (kp[:, 0] / (float(H) * 0.5)) - 1.0
    kp[:, 1] = (kp[:, 1] / (float(W) * 0.5)) - 1.0
    kp = torch.flip(kp.view(1, 1, -1, 2), [3])
    d = F.grid_sample(desc_lowres.unsqueeze(0), kp, align_corners=True)[0, :, 0, :].transpose(0, 1)
    return F.normalize(d, p=2, dim=1)


class Match:
    """Stand-in for cv2.DMatch (queryIdx, trainIdx, distance)."""
    __slots__ = ("queryIdx", "trainIdx", "distance")

    def __init__(self, q, t, d):
        self.queryIdx, self.trainIdx, self.distance = int(q), int(t), float(d)


def nn_both(d1, d2):
    d1 = np.ascontiguousarray(d1, dtype=np.float32)
    d2 = np.ascontiguousarray(d2, dtype=np.float32)
    n1, dim = d1.shape
    n2 = d2.shape[0]
    idx12 = np.empty(n1, np.int32); dist12 = np.empty(n1, np.float64); gap12 = np.empty(n1, np.float64)
    idx21 = np.empty(n2, np.int32); dist21 = np.empty(n2, np.float64)
    lib().xo_nn_both(_p(d1), ctypes.c_int64(n1), _p(d2), ctypes.c_int64(n2), ctypes.c_int64(dim),
                     _p(idx12), _p(dist12), _p(gap12), _p(idx21), _p(dist21))
    return idx12, dist12, gap12, idx21, dist21


def get_matches(d1, d2, mode="strict_mnn", return_arrays=False):
    """matching.py:4-36 with method 'bfmatcher', crossCheck=True (third-party arithmetic, unpinned —
    SURVEY.md a15).  strict_mnn: {(q, t=nn12[q]) : nn21[t] == q} (== NNMatcher matching.py:61-64 without
    its threshold).  legacy_crosscheck: for every q, the nearest t among {t : nn21[t] == q}."""
    if d1.shape[0] == 0 or d2.shape[0] == 0:
        return ([], None) if return_arrays else []
    idx12, dist12, gap12, idx21, dist21 = nn_both(d1, d2)
    if mode == "strict_mnn":
        q = np.nonzero(idx21[idx12] == np.arange(len(idx12)))[0]
        t = idx12[q]; d = dist12[q]
    elif mode == "legacy_crosscheck":
        best = {}
        for tt in range(len(idx21)):
            qq = int(idx21[tt])
            if qq not in best or dist21[tt] < best[qq][1]:
                best[qq] = (tt, dist21[tt])
        q = np.array(sorted(best), dtype=np.int64)
        t = np.array([best[i][0] for i in q], dtype=np.int64)
        d = np.array([best[i][1] for i in q])
    else:
        raise ValueError("unknown mode " + mode)
    ms = [Match(a, b, c) for a, b, c in zip(q, t, d)]
    if return_arrays:
        return ms, dict(idx12=idx12, dist12=dist12, gap12=gap12, idx21=idx21, dist21=dist21)
    return ms


def knn2(d1, d2):
    """The two nearest targets of every query in exact (fp64 direct form) arithmetic, ties -> lower index: idx (n1, 2) int32, dist (n1, 2) float64."""
    d1 = np.ascontiguousarray(d1, dtype=np.float32); d2 = np.ascontiguousarray(d2, dtype=np.float32)
    idx = np.empty((d1.shape[0], 2), np.int32); dist = np.empty((d1.shape[0], 2), np.float64)
    lib().xo_knn2(_p(d1), ctypes.c_int64(d1.shape[0]), _p(d2), ctypes.c_int64(d2.shape[0]), ctypes.c_int64(d1.shape[1]), _p(idx), _p(dist))
    return idx, dist


def knn_ratio_matches(d1, d2, ratio_thresh=0.9):
    """matching.py:20-27: `matcher.knnMatch(desc_1, desc_2, 2)` + Lowe's ratio test `m.distance < 0.9 * n.distance` (distances as the float32 values a
    DMatch carries, compared as Python floats).  Fewer than two targets: the reference's `for m, n in all_matches` fails to unpack -> ValueError."""
    if d1.shape[0] == 0:
        return []
    if d2.shape[0] < 2:
        raise ValueError(f"not enough values to unpack (expected 2, got {d2.shape[0]})")
    idx, dist = knn2(d1, d2)
    df = dist.astype(np.float32).astype(np.float64)
    keep = df[:, 0] < ratio_thresh * df[:, 1]
    return [Match(q, idx[q, 0], df[q, 0]) for q in np.nonzero(keep)[0]]


def thresholdmatcher(d1, d2, threshold=0.4):
    """matching.py:77-102 (ThresholdMatcher.match): every (q, t) with sqrt(2 - 2 clip(<a, b>, -1, 1)) < threshold in row-major order — in exact
    (fp64) arithmetic; the reference's own float32 BLAS product is not bit-pinnable."""
    d1 = np.ascontiguousarray(d1, dtype=np.float32); d2 = np.ascontiguousarray(d2, dtype=np.float32)
    if d1.shape[0] == 0 or d2.shape[0] == 0:
        return []
    cap = d1.shape[0] * d2.shape[0]
    pairs = np.empty((cap, 2), np.int32); dist = np.empty(cap, np.float64)
    n = lib().xo_threshold_pairs(_p(d1), ctypes.c_int64(d1.shape[0]), _p(d2), ctypes.c_int64(d2.shape[0]), ctypes.c_int64(d1.shape[1]),
                                 ctypes.c_double(threshold), _p(pairs), _p(dist), ctypes.c_int64(cap))
    return [Match(a, b, c) for (a, b), c in zip(pairs[:n], dist[:n])]


def nnmatcher(d1, d2, threshold=0.7):
    """matching.py:38-75 (NNMatcher.match), numpy float32 exactly as the reference."""
    a = np.asarray(d1).transpose(); b = np.asarray(d2).transpose()
    if a.shape[1] == 0 or b.shape[1] == 0:
        return []
    dmat = np.dot(a.T, b)
    dmat = np.sqrt(2 - 2 * np.clip(dmat, -1, 1))
    idx = np.argmin(dmat, axis=1)
    scores = dmat[np.arange(dmat.shape[0]), idx]
    keep = scores < threshold
    idx2 = np.argmin(dmat, axis=0)
    keep = np.logical_and(keep, np.arange(len(idx)) == idx2[idx])
    return [Match(i1, i2, d) for i1, i2, d in zip(np.arange(a.shape[1])[keep], idx[keep], scores[keep])]


# ------------------------------------------------------------------------------------------
# the two prediction flows                          SURVEY.md 3.1 / 3.2
# ------------------------------------------------------------------------------------------

DEFAULT_PREDICTION = dict(detection_threshold=0.015, nms=8, topk=0,
                          matching=dict(method="bfmatcher", knn_matches=False, method_kwargs=dict(crossCheck=True)))


def predict_align_image_pair(data, sd, pred=DEFAULT_PREDICTION, hm_head=False, match_mode="strict_mnn"):
    """predict_align_image_pair.py:176-264: forward -> prob*mask -> box_nms -> nonzero ->
    interpolate_descriptors -> get_matches, per pair of the batch."""
    o, t, hm = xpoint_forward(data, sd, hm_head)
    H, W = data["optical"]["image"].shape[2:]
    res = []
    po = o["prob"] * data["optical"]["valid_mask"]
    pt = t["prob"] * data["thermal"]["valid_mask"]
    if pred["nms"] > 0:
        po = box_nms(po, pred["nms"], pred["detection_threshold"], keep_top_k=pred["topk"])
        pt = box_nms(pt, pred["nms"], pred["detection_threshold"], keep_top_k=pred["topk"])
    for i in range(po.shape[0]):
        ko = extract_keypoints(po[i].squeeze(), pred["detection_threshold"])
        kt = extract_keypoints(pt[i].squeeze(), pred["detection_threshold"])
        do = interpolate_descriptors(ko, o["desc"][i], H, W)
        dt = interpolate_descriptors(kt, t["desc"][i], H, W)
        ms = get_matches(do.numpy(), dt.numpy(), match_mode)
        res.append(dict(kp_optical=ko, kp_thermal=kt, desc_optical=do, desc_thermal=dt, matches=ms))
    return res, (o, t, hm), (po, pt)


def predict_keypoints(data, sd, pred=DEFAULT_PREDICTION):
    """predict_keypoints.py:144-216: NMS on the UNMASKED prob, mask applied at extraction."""
    o, t, _ = xpoint_forward(data, sd, False)
    out = []
    for spec, r in (("optical", o), ("thermal", t)):
        p = r["prob"]
        if pred["nms"] > 0:
            p = box_nms(p, pred["nms"], pred["detection_threshold"], keep_top_k=pred["topk"])
        out.append([extract_keypoints(p[i].squeeze(), pred["detection_threshold"], data[spec]["valid_mask"][i].squeeze())
                    for i in range(p.shape[0])])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Data ingest: reference datasets/ImagePairDataset.py:199-208 decodes with cv2.imread and converts with
# cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) / 255.0.  OpenCV is absent from /root/reference and from the image ("parity unpinned" for
# the cv2 call itself); this restates its 8-bit path from the published implementation — modules/imgproc/src/color_rgb.simd.hpp,
# RGB2Gray<uchar>: fixed point with yuv_shift = 14 and the BT.601 coefficients R2Y = 4899, G2Y = 9617, B2Y = 1868 (= 0.299, 0.587,
# 0.114 * 2^14, summing to 2^14), rounding offset 2^13, pinned opencv-python==4.10.0.82 (reference requirements.txt:2) — in exact
# Python integers, independently of the package's host path (xpoint_amd/datasets.py) and of the HIP kernel (xp_ingest_u8).
# ---------------------------------------------------------------------------------------------------------------------
def bgr2gray_u8(rgb):
    """rgb: (..., 3) uint8 in R, G, B order (the decoded pixel; cv2.imread stores the same pixel as B, G, R) -> (...) uint8."""
    import numpy as np
    a = np.asarray(rgb)
    flat = a.reshape(-1, a.shape[-1])
    out = np.empty(len(flat), dtype=np.uint8)
    for i, px in enumerate(flat.tolist()):
        r, g, b = px[0], px[1], px[2]
        out[i] = (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14
    return out.reshape(a.shape[:-1])


def gray_to_float(gray_u8):
    """numpy's `gray / 255.0` of the reference (float64 division) as the float32 the network receives."""
    import numpy as np
    return (np.asarray(gray_u8).astype(np.float64) / 255.0).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------------------
# Perspective warp (predict_align_image_pair.py:308 cv2.warpPerspective(im, H_est, (W, H), borderMode=BORDER_CONSTANT)).
# PARITY UNPINNED: OpenCV is absent; oracle/csrc/oracle_kernels.c restates its documented INTER_LINEAR fixed-point scheme.
# ---------------------------------------------------------------------------------------------------------------------------------
def warp_perspective(img, M, dsize=None, inverse_map=False):
    """img: (H, W) or (H, W, C) numpy uint8 / float32; M (3, 3) forward map src -> dst (x, y); dsize = (width, height) as in cv2."""
    import numpy as np
    img = np.ascontiguousarray(img)
    squeeze = img.ndim == 2
    a = img[..., None] if squeeze else img
    Hs, Ws, C = a.shape
    Wd, Hd = (Ws, Hs) if dsize is None else (int(dsize[0]), int(dsize[1]))
    Mc = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(9))
    out = np.zeros((Hd, Wd, C), dtype=a.dtype)
    a = np.ascontiguousarray(a)
    fn = {np.dtype(np.uint8): lib().xo_warp_perspective_u8, np.dtype(np.float32): lib().xo_warp_perspective_f32}[a.dtype]
    fn(_p(a), _p(out), _p(Mc), ctypes.c_int64(Hs), ctypes.c_int64(Ws), ctypes.c_int64(Hd), ctypes.c_int64(Wd), ctypes.c_int64(C),
       ctypes.c_int(1 if inverse_map else 0))
    return out[..., 0] if squeeze else out


def to_u8_image(img01):
    """predict_align_image_pair.py:271: (np.clip(img, 0.0, 1.0) * 255.0).astype(np.uint8) on a float32 image."""
    import numpy as np
    return (np.clip(np.asarray(img01, dtype=np.float32), 0.0, 1.0) * 255.0).astype(np.uint8)
