"""GPU parity: HIP selective scan (C ABI xp_selective_scan_fwd) vs the oracle and the golden KATs
from the real reference.  Tolerance: the reference's own fp32 bar is rtol 6e-4 / atol 2e-3
(test_selective_scan.py:401); the build target is 1e-5 (scaled by the output magnitude)."""
import numpy as np
import pytest
import torch

from oracle import xpoint_oracle as xo
from oracle.refharness.make_golden import SCAN_CASES, scan_inputs

pytestmark = pytest.mark.gpu


def _tol(ref):
    return 1e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("case", SCAN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_scan_vs_golden_and_oracle(gpu_lib, golden, case):
    from xpoint_amd.kernels import selective_scan_fn
    g = golden("g1_selective_scan.npz")
    name = "scan/%d_%d_%d_%d_%d" % case
    cpu = [torch.from_numpy(x) for x in scan_inputs(name, *case)]
    u, delta, A, Bm, Cm, Dv, bias = [t.cuda() for t in cpu]
    out, last = selective_scan_fn(u, delta, A, Bm, Cm, Dv, bias, True, return_last_state=True)
    ref = g[name + "/out"]
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=_tol(ref))
    o_out, o_last = xo.selective_scan(*cpu, True, return_last_state=True)
    np.testing.assert_allclose(last.cpu().numpy(), o_last.numpy(), rtol=0, atol=_tol(o_last.numpy()))
    if name + "/out_plain" in g.files:
        out2 = selective_scan_fn(u, delta, A, Bm, Cm, None, None, False)
        np.testing.assert_allclose(out2.cpu().numpy(), g[name + "/out_plain"], rtol=0, atol=_tol(ref))


@pytest.mark.parametrize("L", [1, 3, 255, 256, 257, 1023])
def test_scan_ragged_lengths(gpu_lib, L):
    from xpoint_amd.kernels import selective_scan_fn
    case = (2, 2, 5, 1, L)
    cpu = [torch.from_numpy(x) for x in scan_inputs(f"scan/ragged{L}", *case)]
    out = selective_scan_fn(*[t.cuda() for t in cpu], True)
    ref = xo.selective_scan(*cpu, True).numpy()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=_tol(ref))


def test_scan_grouped_delta_and_model_shape(gpu_lib):
    """delta grouped over channels (oflex 'delta_group', selective_scan_fwd_kernel_oflex.cuh:92) and the
    XPoint stage-3 call shape (B, 3072, 300)."""
    from xpoint_amd.kernels import selective_scan_fn
    B, K, C, N, L = 1, 4, 768, 1, 300
    cpu = [torch.from_numpy(x) for x in scan_inputs("scan/stage3", B, K, C, N, L)]
    out = selective_scan_fn(*[t.cuda() for t in cpu], True)
    ref = xo.selective_scan(*cpu, True).numpy()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=_tol(ref))
    # grouped delta: 24 delta rows shared by 768 channels each group of 32
    u, delta, A, Bm, Cm, Dv, bias = cpu
    dg = delta[:, ::32].contiguous()
    bg = bias[::32].contiguous()
    out = selective_scan_fn(u.cuda(), dg.cuda(), A.cuda(), Bm.cuda(), Cm.cuda(), Dv.cuda(), bg.cuda(), True)
    ref = xo.selective_scan(u, dg.repeat_interleave(32, dim=1), A, Bm, Cm, Dv, bg.repeat_interleave(32), True).numpy()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=_tol(ref))


def test_scan_linearity_full_size(gpu_lib):
    """Size-independent property at the BASELINE shape (stage 0 of 480x640: B, 384, 19200): with D=0 the
    op is linear in u for fixed delta:  scan(u1 + 2 u2) == scan(u1) + 2 scan(u2)."""
    from xpoint_amd.kernels import selective_scan_fn
    torch.manual_seed(0)
    B, K, C, N, L = 2, 4, 96, 1, 19200
    dev = "cuda"
    u1 = torch.randn(B, K * C, L, device=dev); u2 = torch.randn(B, K * C, L, device=dev)
    delta = 0.5 * torch.rand(B, K * C, L, device=dev); A = -0.5 * torch.rand(K * C, N, device=dev)
    Bm = torch.randn(B, K, N, L, device=dev); Cm = torch.randn(B, K, N, L, device=dev)
    f = lambda u: selective_scan_fn(u, delta, A, Bm, Cm, None, None, True)
    lhs = f(u1 + 2 * u2); rhs = f(u1) + 2 * f(u2)
    assert float((lhs - rhs).abs().max()) < 1e-4 * float(lhs.abs().max())


def test_scan_errors(gpu_lib):
    from xpoint_amd.kernels import selective_scan_fn
    u = torch.zeros(1, 4, 8, device="cuda")
    with pytest.raises(RuntimeError):
        selective_scan_fn(u, u, torch.zeros(4, 1, device="cuda"), torch.zeros(1, 3, 1, 8, device="cuda"),
                          torch.zeros(1, 3, 1, 8, device="cuda"))           # dim % ngroups != 0
    with pytest.raises(RuntimeError):
        selective_scan_fn(u.cpu(), u, torch.zeros(4, 1), torch.zeros(1, 1, 1, 8), torch.zeros(1, 1, 1, 8))


# ------------------------------------------------------------------------------------------------ f16 / bf16 instantiations, x output
def _chunk_states(u, delta, A, Bm, Cm, Dv, bias, softplus=True, chunk=2048):
    """fp64 restatement of the op's second output (selective_scan_oflex.cpp:206-208; kernel :154-162): per 2048-element chunk and state
    the running prefix (product of exp(delta A) since the row start, h at the chunk end)."""
    u, delta, A, Bm, Cm = [t.double() for t in (u, delta, A, Bm, Cm)]
    Bsz, D, L = u.shape
    G, N = Bm.shape[1], Bm.shape[2]
    dl = delta.repeat_interleave(D // delta.shape[1], dim=1) + (bias.double().repeat_interleave(D // bias.shape[0])[None, :, None] if bias is not None else 0.0)
    if softplus:
        dl = torch.where(dl <= 20.0, torch.log1p(torch.exp(dl)), dl)
    Bf = Bm.repeat_interleave(D // G, dim=1); Cf = Cm.repeat_interleave(D // G, dim=1)          # (B, D, N, L)
    h = torch.zeros(Bsz, D, N, dtype=torch.float64); pa = torch.ones(Bsz, D, N, dtype=torch.float64)
    nch = (L + chunk - 1) // chunk
    x = torch.zeros(Bsz, D, nch, 2 * N, dtype=torch.float64)
    out = torch.zeros(Bsz, D, L, dtype=torch.float64)
    for l in range(L):
        a = torch.exp(dl[:, :, l, None] * A[None])
        h = a * h + dl[:, :, l, None] * Bf[:, :, :, l] * u[:, :, l, None]
        pa = pa * a
        out[:, :, l] = (h * Cf[:, :, :, l]).sum(-1) + (Dv.double()[None] * u[:, :, l] if Dv is not None else 0.0)
        if (l + 1) % chunk == 0 or l == L - 1:
            x[:, :, l // chunk, 0::2] = pa; x[:, :, l // chunk, 1::2] = h
    return out, x


@pytest.mark.parametrize("dtype,rtol,atol", [(torch.float32, 2e-5, 2e-5), (torch.float16, 3e-3, 5e-3), (torch.bfloat16, 3e-2, 5e-2)])
@pytest.mark.parametrize("case", [(2, 4, 24, 1, 64), (1, 2, 16, 1, 2049), (1, 1, 4, 8, 512), (2, 2, 8, 16, 4500), (1, 4, 6, 3, 301)],
                         ids=lambda c: "x".join(map(str, c)))
def test_scan_typed_inputs_and_chunk_states(gpu_lib, case, dtype, rtol, atol):
    """The reference's half-input / float-output instantiations (cusoflex/selective_scan_core_fwd.cu:6-10) with the reference's own
    tolerances (test_selective_scan.py:401-403: f16 rtol 3e-3 / atol 5e-3, bf16 3e-2 / 5e-2; the inputs are rounded to the 16-bit type
    first, as its test does), the per-chunk state output x, and last state = x[:, :, -1, 1::2]."""
    from xpoint_amd.kernels import selective_scan_fwd, selective_scan_fn
    name = "scan/typed_%d_%d_%d_%d_%d" % case
    u, delta, A, Bm, Cm, Dv, bias = [torch.from_numpy(x) for x in scan_inputs(name, *case)]
    u, delta, Bm, Cm = [t.to(dtype) for t in (u, delta, Bm, Cm)]
    ref_out, ref_x = _chunk_states(u.float(), delta.float(), A, Bm.float(), Cm.float(), Dv, bias)
    out, x = selective_scan_fwd(u.cuda(), delta.cuda(), A.cuda(), Bm.cuda(), Cm.cuda(), Dv.cuda(), bias.cuda(), True, 1, True)
    assert out.dtype == torch.float32 and x.shape == ref_x.shape
    if dtype == torch.float32:          # f32: the build's own bar, scaled by the output magnitude like the tests above (sums of up to 16 states)
        rtol, atol = 0.0, 2e-5 * max(1.0, float(ref_out.abs().max()))
    np.testing.assert_allclose(out.cpu().numpy(), ref_out.numpy(), rtol=rtol, atol=atol)
    np.testing.assert_allclose(x.cpu().numpy(), ref_x.numpy(), rtol=rtol, atol=atol)
    o2, last = selective_scan_fn(u.cuda(), delta.cuda(), A.cuda(), Bm.cuda(), Cm.cuda(), Dv.cuda(), bias.cuda(), True, return_last_state=True)
    np.testing.assert_allclose(last.cpu().numpy(), ref_x[:, :, -1, 1::2].numpy(), rtol=rtol, atol=atol)
    if dtype != torch.float32:                # output in the input type (out_float False)
        o3, _ = selective_scan_fwd(u.cuda(), delta.cuda(), A.cuda(), Bm.cuda(), Cm.cuda(), Dv.cuda(), bias.cuda(), True, 1, False)
        assert o3.dtype == dtype
        np.testing.assert_allclose(o3.float().cpu().numpy(), ref_out.numpy(), rtol=4 * rtol, atol=4 * atol)


def test_scan_typed_errors(gpu_lib):
    from xpoint_amd.kernels import selective_scan_fwd
    u = torch.zeros(1, 4, 8, device="cuda", dtype=torch.float16)
    A = torch.zeros(4, 1, device="cuda"); B = torch.zeros(1, 1, 1, 8, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError):
        selective_scan_fwd(u, u.float(), A, B, B)                       # mixed input dtypes (selective_scan_oflex.cpp:163-166)
    with pytest.raises(RuntimeError):
        selective_scan_fwd(u, u, A.half(), B, B)                        # A must be float32
    with pytest.raises(RuntimeError):
        selective_scan_fwd(u.to(torch.int32), u, A, B, B)
