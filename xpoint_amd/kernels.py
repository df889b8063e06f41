"""Operator-level Python surface over the C ABI (thin: argument checks, output allocation, ctypes).

Mirrors the reference's operator interfaces for the hot path so parity tests read like the
reference's own: `selective_scan_fn` (reference vmamba_src/csms6s.py:112-126, pybind op
selective_scan_cuda_oflex.fwd, selective_scan_oflex.cpp:143-231).  Errors from the C ABI surface as
RuntimeError, like TORCH_CHECK failures do in the reference.  `cross_scan_fn` / `cross_merge_fn` mirror
vmamba_src/csm_triton.py:501-517 (same arguments, shapes and layouts)."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import c_i, ptr


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA(HIP) tensor")       # selective_scan_oflex.cpp:152-160
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: this build computes in float32 (got {t.dtype})")
    return t.contiguous()


_ITYPE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def selective_scan_fwd(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True, nrows=1, out_float=True):
    """The pybind op of the reference, `selective_scan_cuda_oflex.fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows,
    out_float) -> [out, x]` (selective_scan_oflex.cpp:143-231): u, delta, B, C of one dtype in {float32, float16, bfloat16}; A, D,
    delta_bias float32; out float32 if out_float else the input dtype; x (B, D, ceil(L / 2048), 2 N) float32, last state =
    x[:, :, -1, 1::2].  nrows is a CUDA launch-shape hint there and is ignored here."""
    for t, n in ((u, "u"), (delta, "delta"), (A, "A"), (B, "B"), (C, "C"), (D, "D"), (delta_bias, "delta_bias")):
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA(HIP) tensor")                     # selective_scan_oflex.cpp:152-160
    if u.dtype not in _ITYPE:
        raise RuntimeError(f"u: dtype must be float32, float16 or bfloat16 (got {u.dtype})")
    for t, n in ((delta, "delta"), (B, "B"), (C, "C")):
        if t.dtype != u.dtype:
            raise RuntimeError(f"{n} must have u's dtype {u.dtype} (got {t.dtype})")
    for t, n in ((A, "A"), (D, "D"), (delta_bias, "delta_bias")):
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"{n} must be float32 (got {t.dtype})")
    if u.dim() != 3 or B.dim() != 4:
        raise RuntimeError("selective_scan_fwd: u must be (B, D, L) and B/C (B, G, N, L)")
    u, delta, A, B, C = u.contiguous(), delta.contiguous(), A.contiguous(), B.contiguous(), C.contiguous()
    D = D.contiguous() if D is not None else None
    delta_bias = delta_bias.contiguous() if delta_bias is not None else None
    batch, dim, L = u.shape
    _, G, N, L2 = B.shape
    if L2 != L or C.shape != B.shape or A.shape != (dim, N) or delta.shape[0] != batch or delta.shape[2] != L:
        raise RuntimeError("selective_scan_fwd: shape mismatch")
    if N > 256:
        raise RuntimeError("selective_scan_fwd: dstate must be <= 256")                # MAX_DSTATE, selective_scan_oflex.cpp:11
    out = torch.empty((batch, dim, L), device=u.device, dtype=torch.float32 if out_float else u.dtype)
    x = torch.empty((batch, dim, (L + 2047) // 2048, 2 * N), device=u.device, dtype=torch.float32)
    _lib.call("xp_selective_scan_fwd_typed", ptr(u), ptr(delta), ptr(A), ptr(B), ptr(C), ptr(D), ptr(delta_bias), ptr(out), ptr(x),
              c_i(_ITYPE[u.dtype]), c_i(int(bool(out_float))), c_i(batch), c_i(dim), c_i(delta.shape[1]), c_i(L), c_i(N), c_i(G),
              c_i(int(bool(delta_softplus))), _lib.current_stream(u))
    return [out, x]


def selective_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True, oflex=True, backend=None,
                      return_last_state=False):
    """u (B, K*C, L); delta (B, Dd, L) with K*C % Dd == 0; A (K*C, N); B, C (B, K, N, L); D, delta_bias
    (K*C)/(Dd).  Returns out (B, K*C, L) float32 (oflex: float output).  float16 / bfloat16 inputs go through
    selective_scan_fwd (the reference's half-input / float-output instantiations)."""
    if u.dtype in (torch.float16, torch.bfloat16):
        out, x = selective_scan_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, 1, oflex)
        return (out, x[:, :, -1, 1::2].contiguous()) if return_last_state else out
    u, delta, A, B, C = (_f32c(t, n) for t, n in ((u, "u"), (delta, "delta"), (A, "A"), (B, "B"), (C, "C")))
    D = _f32c(D, "D"); delta_bias = _f32c(delta_bias, "delta_bias")
    if u.dim() != 3 or B.dim() != 4:
        raise RuntimeError("selective_scan_fn: u must be (B, D, L) and B/C (B, G, N, L)")
    batch, dim, L = u.shape
    _, G, N, L2 = B.shape
    if L2 != L or C.shape != B.shape or A.shape != (dim, N) or delta.shape[0] != batch or delta.shape[2] != L:
        raise RuntimeError("selective_scan_fn: shape mismatch")
    out = torch.empty_like(u)
    last = torch.empty((batch, dim, N), device=u.device, dtype=torch.float32) if return_last_state else None
    with torch.cuda.device(u.device):
        _lib.call("xp_selective_scan_fwd", ptr(u), ptr(delta), ptr(A), ptr(B), ptr(C), ptr(D), ptr(delta_bias), ptr(out),
                  ptr(last), c_i(batch), c_i(dim), c_i(delta.shape[1]), c_i(L), c_i(N), c_i(G), c_i(int(bool(delta_softplus))),
                  _lib.current_stream())
    return (out, last) if return_last_state else out


def _csm_dtype(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: input must be a CUDA(HIP) tensor")
    if t.dtype not in _ITYPE:
        raise RuntimeError(f"{who}: dtype must be float32, float16 or bfloat16 (got {t.dtype})")
    return _ITYPE[t.dtype]


def cross_scan_fn(x, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0, force_torch=False):
    """Reference `cross_scan_fn` (csm_triton.py:501-507).  x: (B,C,H,W) | (B,H,W,C) | one_by_one: (B,4,C,H,W) | (B,H,W,4,C);
    returns (B,4,C,L) if out_channel_first else (B,L,4,C).  scans 0 cross scan, 1 unidirectional, 2 bidirectional.  `force_torch` is accepted and
    ignored (there is one implementation).  Inference only: no autograd graph is recorded.
    Two combinations deliberately implement the INTENDED permutation, not the reference's mis-indexed output (oracle/refharness/make_golden.py documents both):
    one_by_one + channel-last in + scans = 2, and channel-first in / channel-last out with scans = 1."""
    dt = _csm_dtype(x, "cross_scan_fn")
    if scans not in (0, 1, 2):
        raise RuntimeError(f"cross_scan_fn: scans must be 0, 1 or 2 (got {scans})")
    if x.dim() != (5 if one_by_one else 4):
        raise RuntimeError(f"cross_scan_fn: expected a {5 if one_by_one else 4}-d tensor, got {tuple(x.shape)}")
    if one_by_one:
        B, K, C, H, W = x.shape if in_channel_first else (x.shape[0], x.shape[3], x.shape[4], x.shape[1], x.shape[2])
        if K != 4:
            raise RuntimeError("cross_scan_fn: one_by_one input must hold 4 routes")
    else:
        B, C, H, W = x.shape if in_channel_first else (x.shape[0], x.shape[3], x.shape[1], x.shape[2])
    x = x.contiguous()
    y = torch.empty((B, 4, C, H * W) if out_channel_first else (B, H * W, 4, C), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):       # as the reference (csm_triton.py:231): the launch targets x's device, whichever device is current
        _lib.call("xp_cross_scan", ptr(x), ptr(y), c_i(dt), c_i(B), c_i(C), c_i(H), c_i(W), c_i(int(bool(in_channel_first))),
                  c_i(int(bool(out_channel_first))), c_i(int(bool(one_by_one))), c_i(scans), _lib.current_stream(x))
    return y


def cross_merge_fn(y, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0, force_torch=False):
    """Reference `cross_merge_fn` (csm_triton.py:511-517).  y: (B,4,C,H,W) if out_channel_first else (B,H,W,4,C) (the scan's OUT layout);
    returns (B,C,L) if in_channel_first else (B,L,C) — one_by_one: (B,4,C,L) / (B,L,4,C).  Adds associate as the reference's do."""
    dt = _csm_dtype(y, "cross_merge_fn")
    if scans not in (0, 1, 2):
        raise RuntimeError(f"cross_merge_fn: scans must be 0, 1 or 2 (got {scans})")
    if y.dim() != 5:
        raise RuntimeError(f"cross_merge_fn: expected (B,4,C,H,W) or (B,H,W,4,C), got {tuple(y.shape)}")
    B, K, C, H, W = y.shape if out_channel_first else (y.shape[0], y.shape[3], y.shape[4], y.shape[1], y.shape[2])
    if K != 4:
        raise RuntimeError("cross_merge_fn: input must hold 4 routes")
    y = y.contiguous()
    if one_by_one:
        out = torch.empty((B, 4, C, H * W) if in_channel_first else (B, H * W, 4, C), device=y.device, dtype=y.dtype)
    else:
        out = torch.empty((B, C, H * W) if in_channel_first else (B, H * W, C), device=y.device, dtype=y.dtype)
    with torch.cuda.device(y.device):
        _lib.call("xp_cross_merge", ptr(y), ptr(out), c_i(dt), c_i(B), c_i(C), c_i(H), c_i(W), c_i(int(bool(in_channel_first))),
                  c_i(int(bool(out_channel_first))), c_i(int(bool(one_by_one))), c_i(scans), _lib.current_stream(y))
    return out
