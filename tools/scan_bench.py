"""Micro-benchmark of xp_selective_scan_fwd (reference operator boundary) at the XPoint call shapes (16 images)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd.kernels import selective_scan_fn
torch.manual_seed(0)
shapes = [(16, 4, 96, 1, 19200), (16, 4, 192, 1, 4800), (16, 4, 384, 1, 1200), (16, 4, 768, 1, 300), (4, 4, 96, 1, 65536), (8, 4, 96, 16, 4096)]
if os.environ.get("SCAN_ONLY"):
    shapes = [shapes[int(i)] for i in os.environ["SCAN_ONLY"].split(",")]
for (B, K, C, N, L) in shapes:
    D = K * C
    u = torch.randn(B, D, L, device="cuda"); delta = 0.5 * torch.rand(B, D, L, device="cuda")
    A = -0.5 * torch.rand(D, N, device="cuda"); Bm = torch.randn(B, K, N, L, device="cuda"); Cm = torch.randn(B, K, N, L, device="cuda")
    Dv = torch.randn(D, device="cuda"); bias = 0.5 * torch.rand(D, device="cuda")
    for _ in range(3): selective_scan_fn(u, delta, A, Bm, Cm, Dv, bias, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): selective_scan_fn(u, delta, A, Bm, Cm, Dv, bias, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = 12.0 * B * D * L + 8.0 * B * K * N * L
    print(f"B{B} D{D} N{N} L{L}: {ms*1e3:8.1f} us  {by/ms/1e6:7.0f} GB/s algorithmic ({by/1e6:.1f} MB)  frac of 8 TB/s {by/ms/1e6/8000:.3f}", flush=True)
