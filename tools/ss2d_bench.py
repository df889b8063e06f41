"""Micro-benchmark of xp_ss2d_core_fwd at the model's four stage shapes (16 images of 480x640).  usage: ss2d_bench.py [mode ...]
mode: -1 auto (default), 0 chunked three-pass form, 1 sequential form.  Environment knobs (XP_SS2D_SEQ_MFMA, XP_SS2D_TBUDGET ...) are
read once per process: run one process per setting."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L
lib = L.load()
modes = [int(a) for a in sys.argv[1:]] or [-1]
B = int(os.environ.get("SB_BATCH", "16"))
shapes = [(96, 120, 160), (192, 60, 80), (384, 30, 40), (768, 15, 20)]
if os.environ.get("SB_ONLY"): shapes = [shapes[int(i)] for i in os.environ["SB_ONLY"].split(",")]
if os.environ.get("SB_SHAPES"): shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SB_SHAPES"].split(";")]      # "C,H,W;C,H,W"
torch.manual_seed(0)
for (C, H, W) in shapes:
    R = (C + 15) // 16
    M = B * H * W
    u = torch.rand(B, H, W, C, device="cuda") * 1.3 - 0.3
    xdbl = torch.randn(M, 4 * (R + 2), device="cuda") * 0.5
    dtw = (torch.rand(4, R, C, device="cuda") * 2 - 1) * R ** -0.5
    dtb = torch.rand(4, C, device="cuda") * 4.65 - 6.9
    A = -torch.exp(torch.rand(4, C, device="cuda") - 0.5)
    Dd = torch.rand(4, C, device="cuda") + 0.5
    lnw, lnb = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    out = torch.empty(B, H, W, C, device="cuda")
    nbytes = lib.xp_ss2d_core_workspace_bytes(B, H, W, C)
    ws = torch.empty(nbytes // 4 + 4, device="cuda")
    st = L.current_stream()
    outs = {}
    for mode in modes:
        L.call("xp_ss2d_core_set_mode", mode)
        def run():
            L.call("xp_ss2d_core_fwd", L.ptr(u), L.ptr(xdbl), L.ptr(dtw), L.ptr(dtb), L.ptr(A), L.ptr(Dd), L.ptr(lnw), L.ptr(lnb), L.ptr(out),
                   L.ptr(ws), nbytes, B, H, W, C, R, 1, 1e-5, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        outs[mode] = out.clone()
        by = 4.0 * M * (7 * C + 4 * (R + 2))
        import zlib
        crc = zlib.crc32(out.cpu().numpy().tobytes())
        print(f"C{C:4d} {H}x{W} R{R:2d} mode {mode:2d}: {us:7.1f} us  {by/us/1e6:6.2f} TB/s algorithmic   crc {crc:08x}", flush=True)
    L.call("xp_ss2d_core_set_mode", -1)
    if len(outs) > 1:
        ks = list(outs)
        print(f"      max |mode {ks[0]} - mode {ks[1]}| = {(outs[ks[0]] - outs[ks[1]]).abs().max().item():.3e}")
