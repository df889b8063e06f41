"""Micro-benchmark + CRC of xp_dwconv3x3_silu (f32 class) at the model's four stage shapes (16 images of 480 x 640)."""
import os, sys, zlib, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L
B = 16
st = L.current_stream()
torch.manual_seed(0)
tot = 0.0
for (C, H, W) in [(96, 120, 160), (192, 60, 80), (384, 30, 40), (768, 15, 20)]:
    x = torch.randn(B, H, W, C, device="cuda"); w = torch.randn(9, C, device="cuda") * 0.3; y = torch.empty_like(x)
    def run(): L.call("xp_dwconv3x3_silu", L.ptr(x), L.ptr(w), L.ptr(y), B, H, W, C, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    crc = zlib.crc32(y.cpu().numpy().tobytes())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n; tot += ms
    print(f"H {H:4d} W {W:4d} C {C:4d}: {ms*1e3:7.1f} us  {8.0*B*H*W*C/ms/1e6:7.0f} GB/s  crc {crc:08x}", flush=True)
print(f"sum {tot*1e3:.1f} us (x2 blocks per stage = {2*tot:.3f} ms per step)")
