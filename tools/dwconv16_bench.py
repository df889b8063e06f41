"""Micro-benchmark of xp_dwconv3x3_silu_f16 at the model's four stage shapes (16 images of 480 x 640)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L
torch.manual_seed(0)
st = L.current_stream()
tot = 0.0
for (H, W, C) in [(120, 160, 96), (60, 80, 192), (30, 40, 384), (15, 20, 768)]:
    B = 16
    x = torch.randn(B, H, W, C, device="cuda").half(); w = torch.randn(9, C, device="cuda") * 0.2; y = torch.empty_like(x)
    def run(): L.call("xp_dwconv3x3_silu_f16", L.ptr(x), L.ptr(w), L.ptr(y), None, B, H, W, C, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50; tot += ms
    print(f"H {H:4d} W {W:4d} C {C:4d}: {ms*1e3:7.1f} us  {4.0*B*H*W*C/ms/1e9:6.2f} TB/s", flush=True)
print(f"sum {tot*1e3:.1f} us")
