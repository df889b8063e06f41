"""Operator-level Python surface over the C ABI (thin: argument checks, output allocation, ctypes).

Mirrors the reference's operator interfaces for the hot path so parity tests read like the
reference's own: `selective_scan_fn` (reference vmamba_src/csms6s.py:112-126, pybind op
selective_scan_cuda_oflex.fwd, selective_scan_oflex.cpp:143-231).  Errors from the C ABI surface as
RuntimeError, like TORCH_CHECK failures do in the reference."""
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


def selective_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True, oflex=True, backend=None,
                      return_last_state=False):
    """u (B, K*C, L); delta (B, Dd, L) with K*C % Dd == 0; A (K*C, N); B, C (B, K, N, L); D, delta_bias
    (K*C)/(Dd).  Returns out (B, K*C, L) float32 (oflex: float output)."""
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
    _lib.call("xp_selective_scan_fwd", ptr(u), ptr(delta), ptr(A), ptr(B), ptr(C), ptr(D), ptr(delta_bias), ptr(out),
              ptr(last), c_i(batch), c_i(dim), c_i(delta.shape[1]), c_i(L), c_i(N), c_i(G), c_i(int(bool(delta_softplus))),
              _lib.current_stream())
    return (out, last) if return_last_state else out
