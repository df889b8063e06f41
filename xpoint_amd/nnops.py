"""Thin tensor-level wrappers over the C ABI for the small conv networks (BASELINE config 1 backbones and the
RegNet head): allocate the output, pass pointers.  Activations are NHWC float32 CUDA tensors."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import ptr

ACT = {"none": 0, "gelu": 1, "relu": 2, "relu_after_affine": 3}


def conv3x3(x, w_ohwi, bias=None, scale=None, shift=None, stride=1, reflect=False, act="none"):
    """x (B,H,W,Ci), w (Co,3,3,Ci) -> (B,Ho,Wo,Co); pad 1 (zero or reflection)."""
    B, H, W, Ci = x.shape
    Co = w_ohwi.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Ho, Wo, Co), device=x.device)
    _lib.call("xp_conv3x3_nhwc", ptr(x), ptr(w_ohwi), ptr(y), ptr(bias), ptr(scale), ptr(shift), B, H, W, Ci, Co, stride,
              int(bool(reflect)), ACT[act], _lib.current_stream())
    return y


def split_h2(w_nk):
    """Offline two-plane fp16 split of a weight matrix (N, K) for the split-fp16 dense engine (xp_split_weights_h2)."""
    import ctypes
    N, K = w_nk.shape
    out = torch.empty(_lib.load().xp_split_weights_h2_bytes(N, K), dtype=torch.uint8, device=w_nk.device)
    _lib.call("xp_split_weights_h2", ptr(w_nk.contiguous()), ctypes.c_void_p(out.data_ptr()), N, K, _lib.current_stream())
    return out


def conv3x3_h2(x, w_split, Co, bias=None, scale=None, shift=None, stride=1, reflect=False, act="none"):
    """conv3x3 on the split-fp16 engine (f32-grade, three fp16 MFMA products): w_split = split_h2(w_ohwi.view(Co, 9 * Ci))."""
    import ctypes
    B, H, W, Ci = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Ho, Wo, Co), device=x.device)
    _lib.call("xp_conv3x3_nhwc_h2", ptr(x), ctypes.c_void_p(w_split.data_ptr()), ptr(y), ptr(bias), ptr(scale), ptr(shift), B, H, W, Ci, Co, stride,
              int(bool(reflect)), ACT[act], _lib.current_stream())
    return y


def costvolume_mean(a_bhc, b_bhc):
    """v[n][p] = a[n][p] . mean_q b[n][q]  (RegNet.py:44-52 cost volume + global average pool, without the volume)."""
    B, hw, C = a_bhc.shape
    v = torch.empty((B, hw), device=a_bhc.device)
    _lib.call("xp_costvolume_mean", ptr(a_bhc), ptr(b_bhc), ptr(v), B, hw, C, _lib.current_stream())
    return v


def _rows_ptr(t):
    """Pointer of a 2-D view whose rows are contiguous (column slices of a wider matrix are fine: lda carries the stride)."""
    import ctypes
    assert t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32
    return ctypes.c_void_p(t.data_ptr())


def linear(x2d, w_nk, bias=None, scale=None, shift=None, act="none", lda=None, out=None, ldc=None):
    """x2d (M, K[..lda]) @ w (N,K)^T -> (M,N)."""
    M = x2d.shape[0]
    N, K = w_nk.shape
    lda = int(lda if lda is not None else x2d.stride(0))
    if out is None:
        out = torch.empty((M, N), device=x2d.device)
    ldc = int(ldc if ldc is not None else N)
    import ctypes
    _lib.call("xp_gemm_nt", _rows_ptr(x2d), ptr(w_nk), ctypes.c_void_p(out.data_ptr()), ptr(bias), ptr(scale), ptr(shift), None, M, N, K,
              lda, ldc, 0, ACT[act], _lib.current_stream())
    return out


def maxpool2(x):
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W // 2, C), device=x.device)
    _lib.call("xp_maxpool2_nhwc", ptr(x), ptr(y), B, H, W, C, _lib.current_stream())
    return y


def l2norm_rows(x2d, eps=1e-12):
    y = torch.empty_like(x2d)
    _lib.call("xp_l2norm_rows", ptr(x2d), ptr(y), x2d.shape[0], x2d.shape[1], float(eps), _lib.current_stream())
    return y


def softmax_shuffle(logits_nhwc, r=8, mode=0):
    B, Hc, Wc, C = logits_nhwc.shape
    p = torch.empty((B, Hc * r, Wc * r), device=logits_nhwc.device)
    _lib.call("xp_softmax_shuffle", ptr(logits_nhwc), ptr(p), B, Hc, Wc, r, C, mode, _lib.current_stream())
    return p


def nchw(x_nhwc):
    B, H, W, C = x_nhwc.shape
    y = torch.empty((B, C, H, W), device=x_nhwc.device)
    _lib.call("xp_nhwc_to_nchw", ptr(x_nhwc), ptr(y), B, H * W, C, _lib.current_stream())
    return y


def bn_affine(sd, pre, eps=1e-5):
    scale = sd[pre + "weight"].double() / torch.sqrt(sd[pre + "running_var"].double() + eps)
    shift = sd[pre + "bias"].double() - sd[pre + "running_mean"].double() * scale
    return scale.float().contiguous(), shift.float().contiguous()


def conv_w(w_oihw, pad_ci_to=None):
    """(Co,Ci,3,3) -> (Co,3,3,Ci') contiguous; Ci padded with zeros to a multiple of 4 when needed."""
    w = w_oihw.permute(0, 2, 3, 1)
    ci = w.shape[-1]
    tgt = pad_ci_to or ((ci + 3) // 4 * 4)
    if tgt != ci:
        w = torch.nn.functional.pad(w, (0, tgt - ci))
    return w.contiguous().float()


def gray_to_nhwc4(img):
    """(B,1,H,W) -> (B,H,W,4) with the image in channel 0 (the 1-input-channel first conv runs as Ci = 4)."""
    B, _, H, W = img.shape
    x = torch.zeros((B, H, W, 4), device=img.device)
    x[..., 0] = img[:, 0]
    return x
