"""GPU parity of the ring dense engine (csrc/ring_core.h, csrc/gemm_ring.hip): the split-fp16 GEMM on pre-split ("P32") activations and the one-product fp16
instance behind xp_gemm_nt_f16.  Bars: the split class within 2e-5 of fp64 like every f32-grade dense kernel here and BIT-IDENTICAL to xp_gemm_nt_h2's tile
kernel (same partial products, same summation order); the fp16 class bit-identical to the round-4 tile kernel it replaces on the long-K layers; the P32
producers (converter, LayerNorm, SS2D out_norm, the GEMM's own P32 output) reproduce the in-register split of gemm_h2_core.h bit for bit."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from xpoint_amd import synth

pytestmark = pytest.mark.gpu


def _lib():
    from xpoint_amd import _lib as L
    return L


def _u(name, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(synth.uniform(name, shape, lo, hi))


def _vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _split_h2(L, Wd):
    N, K = Wd.shape
    buf = torch.empty(L.load().xp_split_weights_h2_bytes(N, K), dtype=torch.uint8, device="cuda")
    L.call("xp_split_weights_h2", L.ptr(Wd), _vp(buf), N, K, L.current_stream())
    return buf


def _p32(L, Ad):
    M, K = Ad.shape
    out = torch.empty(L.load().xp_p32_bytes(M, K), dtype=torch.uint8, device="cuda")
    L.call("xp_split_activations_h2", L.ptr(Ad), _vp(out), M, K, K, L.current_stream())
    return out


def _p32_planes(buf, M, K):
    """(hi, lo) as float64 (M, K) from a P32 image"""
    p = buf.cpu().view(torch.float16).view(M, K // 32, 2, 32).double()
    return p[:, :, 0].reshape(M, K), p[:, :, 1].reshape(M, K)


def _ref_split(x):
    """the two-way split of gemm_h2_core.h in torch: h0 = fp16(x), h1 = fp16(x - h0)"""
    h0 = x.half()
    h1 = (x - h0.float()).half()
    return h0, h1


def test_split_activations_p32_layout(gpu_lib):
    L = _lib()
    M, K = 77, 96
    A = _u("p32a", (M, K), -3.0, 3.0)
    A[0, 0] = 1e-6; A[1, 5] = 1000.25; A[2, 7] = -0.0; A[3, 31] = 2.0 ** -20
    hi, lo = _p32_planes(_p32(L, A.cuda()), M, K)
    h0, h1 = _ref_split(A)
    assert torch.equal(hi, h0.double()) and torch.equal(lo, h1.double())
    err = (hi + lo - A.double()).abs()
    assert float((err / A.double().abs().clamp_min(2.0 ** -3)).max()) <= 2.0 ** -23


@pytest.mark.parametrize("M,N,K,act,res", [(300, 384, 384, 0, False), (1000, 256, 256, 0, True), (19200, 384, 384, 0, True), (4800, 768, 768, 0, False),
                                           (2500, 1536, 384, 1, False), (2400, 384, 1536, 0, True), (4800, 3072, 768, 1, False), (131, 264, 288, 2, False),
                                           (260, 512, 32, 3, False)])
def test_gemm_nt_h2s_vs_fp64_and_h2_bits(gpu_lib, M, N, K, act, res):
    """within 2e-5 of fp64; where xp_gemm_nt_h2 runs its tile kernel (K < 768 or N < 384: the ping-pong h2p form sums even and odd slabs apart) the
    two engines agree bit for bit; every tile shape of the ring engine gives the same bits (XP_RING_TILE is read once: checked by tools/ring_bench)."""
    L = _lib()
    A = _u(f"rA{M}{N}{K}", (M, K)); Wt = _u(f"rW{M}{N}{K}", (N, K), -0.1, 0.1); bias = _u(f"rb{M}{N}{K}", (N,))
    scale = _u(f"rs{M}{N}{K}", (N,), 0.5, 1.5) if act in (2, 3) else None
    shift = _u(f"rh{M}{N}{K}", (N,), -0.2, 0.2) if act in (2, 3) else None
    R = _u(f"rr{M}{N}{K}", (M, N)) if res else None
    ref = F.linear(A.double(), Wt.double(), bias.double())
    if act == 1:
        ref = F.gelu(ref)
    if act == 2:
        ref = ref.clamp_min(0)
    if scale is not None:
        ref = ref * scale.double() + shift.double()
    if act == 3:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + R.double()
    Ad, Wd, bd = A.cuda(), Wt.cuda(), bias.cuda()
    sd, hd = (scale.cuda(), shift.cuda()) if scale is not None else (None, None)
    Rd = R.cuda() if res else None
    Wx = _split_h2(L, Wd)
    Ap = _p32(L, Ad)
    C = torch.full((M, N), float("nan"), device="cuda")
    L.call("xp_gemm_nt_h2s", _vp(Ap), _vp(Wx), L.ptr(C), 0, L.ptr(bd), L.ptr(sd), L.ptr(hd), L.ptr(Rd), M, N, K, N, N, act, L.current_stream())
    err = float((C.cpu().double() - ref).abs().max())
    assert err <= 2e-5 * max(1.0, float(ref.abs().max())), err
    C2 = torch.empty((M, N), device="cuda")
    L.call("xp_gemm_nt_h2", L.ptr(Ad), _vp(Wx), L.ptr(C2), L.ptr(bd), L.ptr(sd), L.ptr(hd), L.ptr(Rd), M, N, K, K, N, N, act, L.current_stream())
    if K < 768 or N < 384:
        assert torch.equal(C, C2), float((C - C2).abs().max())
    else:
        assert float((C - C2).abs().max()) <= 4e-6 * max(1.0, float(ref.abs().max()))
    # the P32 output (what fc1 hands to fc2) is the split of the f32 output, bit for bit
    if N % 32 == 0:
        Cp = torch.empty(L.load().xp_p32_bytes(M, N), dtype=torch.uint8, device="cuda")
        L.call("xp_gemm_nt_h2s", _vp(Ap), _vp(Wx), _vp(Cp), 2, L.ptr(bd), L.ptr(sd), L.ptr(hd), L.ptr(Rd), M, N, K, N, N, act, L.current_stream())
        hi, lo = _p32_planes(Cp, M, N)
        h0, h1 = _ref_split(C.cpu())
        assert torch.equal(hi, h0.double()) and torch.equal(lo, h1.double())


def test_gemm_nt_h2s_in_place_residual_and_chain(gpu_lib):
    """C aliasing res (the block's `x = x + f(x)`), and fc1 -> P32 -> fc2 chained exactly as the model does"""
    L = _lib()
    M, C_, H4 = 1500, 384, 1536
    X = _u("chx", (M, C_)); W1 = _u("chw1", (H4, C_), -0.08, 0.08); b1 = _u("chb1", (H4,), -0.2, 0.2)
    W2 = _u("chw2", (C_, H4), -0.04, 0.04); b2 = _u("chb2", (C_,), -0.2, 0.2)
    ref = X.double() + F.linear(F.gelu(F.linear(X.double(), W1.double(), b1.double())), W2.double(), b2.double())
    Xd = X.cuda()
    W1x, W2x = _split_h2(L, W1.cuda()), _split_h2(L, W2.cuda())
    b1d, b2d = b1.cuda(), b2.cuda()
    Xp = _p32(L, Xd)
    HB = torch.empty(L.load().xp_p32_bytes(M, H4), dtype=torch.uint8, device="cuda")
    st = L.current_stream()
    L.call("xp_gemm_nt_h2s", _vp(Xp), _vp(W1x), _vp(HB), 2, L.ptr(b1d), None, None, None, M, H4, C_, H4, 0, 1, st)
    L.call("xp_gemm_nt_h2s", _vp(HB), _vp(W2x), L.ptr(Xd), 0, L.ptr(b2d), None, None, L.ptr(Xd), M, C_, H4, C_, C_, 0, st)
    err = float((Xd.cpu().double() - ref).abs().max())
    assert err <= 2e-5 * max(1.0, float(ref.abs().max())), err


def test_gemm_nt_h2s_argument_errors(gpu_lib):
    L = _lib()
    A = torch.zeros(64 * 40 * 4, dtype=torch.uint8, device="cuda"); W = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda"); C = torch.zeros((64, 64), device="cuda")
    st = L.current_stream()
    with pytest.raises(L.XPointHipError):
        L.call("xp_gemm_nt_h2s", _vp(A), _vp(W), L.ptr(C), 0, None, None, None, None, 64, 64, 40, 64, 0, 0, st)          # K % 32
    with pytest.raises(L.XPointHipError):
        L.call("xp_gemm_nt_h2s", _vp(A), _vp(W), L.ptr(C), 0, None, None, None, None, 64, 60, 32, 60, 0, 0, st)          # N % 8
    with pytest.raises(L.XPointHipError):
        L.call("xp_gemm_nt_h2s", _vp(A), _vp(W), L.ptr(C), 1, None, None, None, None, 64, 64, 32, 64, 0, 0, st)          # out_fmt
    with pytest.raises(L.XPointHipError):
        L.call("xp_gemm_nt_h2s", _vp(A), _vp(W), L.ptr(C), 2, None, None, None, None, 64, 40, 32, 40, 0, 0, st)          # P32 output needs N % 32


@pytest.mark.parametrize("rows,C", [(1000, 384), (333, 768), (77, 96), (50, 32)])
def test_layernorm_p32_matches_layernorm_bits(gpu_lib, rows, C):
    L = _lib()
    x = _u(f"lnp{rows}{C}", (rows, C), -4.0, 4.0).cuda(); w = _u(f"lnw{C}", (C,), 0.5, 1.5).cuda(); b = _u(f"lnb{C}", (C,), -0.5, 0.5).cuda()
    y = torch.empty((rows, C), device="cuda")
    L.call("xp_layernorm", L.ptr(x), L.ptr(y), L.ptr(w), L.ptr(b), rows, C, 1e-5, 0, L.current_stream())
    yp = torch.empty(L.load().xp_p32_bytes(rows, C), dtype=torch.uint8, device="cuda")
    L.call("xp_layernorm_p32", L.ptr(x), _vp(yp), L.ptr(w), L.ptr(b), rows, C, 1e-5, L.current_stream())
    hi, lo = _p32_planes(yp, rows, C)
    h0, h1 = _ref_split(y.cpu())
    assert torch.equal(hi, h0.double()) and torch.equal(lo, h1.double())


@pytest.mark.parametrize("B,H,W,C,R", [(2, 15, 20, 384, 24), (2, 8, 10, 768, 48)])
def test_ss2d_core_p32_output_matches_f32_bits(gpu_lib, B, H, W, C, R):
    """the sequential deep-stage form's P32 output = the split of its f32 output (out_norm), bit for bit"""
    L = _lib()
    lib = L.load()
    assert lib.xp_ss2d_core_p32_supported(H, W, C, R) == 1
    M = B * H * W
    u = _u(f"su{C}", (M, C)).cuda(); xdbl = _u(f"sx{C}", (M, 4 * (R + 2)), -0.5, 0.5).cuda()
    wdt = _u(f"sw{C}", (4, R, C), -0.2, 0.2).cuda(); dtb = _u(f"sb{C}", (4, C), -3.0, -1.0).cuda()
    A = (-torch.exp(_u(f"sa{C}", (4, C), -1.0, 1.0))).cuda(); D = _u(f"sd{C}", (4, C)).cuda()
    lw = _u(f"slw{C}", (C,), 0.5, 1.5).cuda(); lb = _u(f"slb{C}", (C,), -0.5, 0.5).cuda()
    ws = torch.empty(lib.xp_ss2d_core_workspace_bytes(B, H, W, C), dtype=torch.uint8, device="cuda")
    out = torch.empty((M, C), device="cuda")
    st = L.current_stream()
    L.call("xp_ss2d_core_fwd_ex", L.ptr(u), L.ptr(xdbl), L.ptr(wdt), L.ptr(dtb), L.ptr(A), L.ptr(D), L.ptr(lw), L.ptr(lb), L.ptr(out), 0,
           _vp(ws), ws.numel(), B, H, W, C, R, 1, 1e-5, st)
    outp = torch.empty(lib.xp_p32_bytes(M, C), dtype=torch.uint8, device="cuda")
    L.call("xp_ss2d_core_fwd_ex", L.ptr(u), L.ptr(xdbl), L.ptr(wdt), L.ptr(dtb), L.ptr(A), L.ptr(D), L.ptr(lw), L.ptr(lb), _vp(outp), 2,
           _vp(ws), ws.numel(), B, H, W, C, R, 1, 1e-5, st)
    hi, lo = _p32_planes(outp, M, C)
    h0, h1 = _ref_split(out.cpu())
    assert torch.equal(hi, h0.double()) and torch.equal(lo, h1.double())
    # the chunked form has no P32 output: refused, not silently f32
    assert lib.xp_ss2d_core_p32_supported(120, 160, 96, 6) == 0
    with pytest.raises(L.XPointHipError):
        L.call("xp_ss2d_core_fwd_ex", L.ptr(u), L.ptr(xdbl), L.ptr(wdt), L.ptr(dtb), L.ptr(A), L.ptr(D), L.ptr(lw), L.ptr(lb), _vp(outp), 2,
               _vp(ws), ws.numel(), 1, 120, 160, 96, 6, 1, 1e-5, st)


@pytest.mark.parametrize("M,N,K,act,res,c_f32", [(19200, 384, 384, 0, False, 0), (2500, 1536, 384, 1, False, 0), (2400, 384, 1536, 0, True, 0),
                                                 (4800, 768, 768, 0, True, 0), (1000, 3072, 768, 1, False, 0), (777, 384, 448, 0, False, 1)])
def test_gemm_f16_long_k_layers_bits(gpu_lib, M, N, K, act, res, c_f32):
    """xp_gemm_nt_f16 on the long-K layers of the deep stages against the rounding recipe evaluated in float64 (round 6: the one-product ring
    instance that could take these layers — bit-identical, slower in the step — was removed from the product; tools/ring_bench.hip still builds it)"""
    L = _lib()
    A = _u(f"fA{M}{N}{K}", (M, K)).half(); Wt = _u(f"fW{M}{N}{K}", (N, K), -0.1, 0.1).half(); bias = _u(f"fb{M}{N}{K}", (N,))
    R = _u(f"fr{M}{N}{K}", (M, N)).half() if res else None
    acc = F.linear(A.double(), Wt.double())
    v = (acc + bias.double()).float().half()                       # f32 accumulate (emulated in f64, then one rounding), + bias, -> half
    if act == 1:
        v = F.gelu(v.float()).half()
    if res:
        v = (v.float() + R.float()).half()
    Ad, Wd, bd = A.cuda(), Wt.cuda(), bias.cuda()
    Rd = R.cuda() if res else None
    C = torch.empty((M, N), device="cuda", dtype=torch.float32 if c_f32 else torch.float16)
    L.call("xp_gemm_nt_f16", _vp(Ad), _vp(Wd), _vp(C), c_f32, L.ptr(bd), None, None, _vp(Rd), M, N, K, K, N, N, act, L.current_stream())
    got = C.cpu().float()
    ref = v.float()
    # A different f32 accumulation order can move the half pre-activation by one ulp; GELU (slope <= 1.13) and the residual add (cancellation) carry that
    # ulp into results of a smaller binade: the bound is one ulp at the magnitude of the pre-activation plus the residual, plus the accumulation's absolute error
    mag = (acc + bias.double()).abs().float() + (R.float().abs() if res else 0.0)
    tol = mag * 2.0 ** -9 + 1e-6
    bad = (got - ref).abs() > tol
    assert int(bad.sum()) == 0, (int(bad.sum()), float((got - ref).abs().max()))
    assert float((got != ref).float().mean()) <= 0.02          # and almost everywhere the same bits
