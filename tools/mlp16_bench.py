"""Micro-benchmark of the fast class's fused MLP launches (xp_ln_mlp_fused_f16) at the model's stage-0 / stage-1 shapes (16 images of 480 x 640)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L
torch.manual_seed(0)
st = L.current_stream()
for (M, C) in [(307200, 96), (76800, 192)]:
    H4 = 4 * C
    X = torch.randn(M, C, device="cuda").half(); lw = torch.ones(C, device="cuda"); lb = torch.zeros(C, device="cuda")
    W1 = (torch.randn(H4, C, device="cuda") * 0.05).half(); b1 = torch.randn(H4, device="cuda") * 0.1
    W2 = (torch.randn(C, H4, device="cuda") * 0.05).half(); b2 = torch.randn(C, device="cuda") * 0.1
    def run(): L.call("xp_ln_mlp_fused_f16", L.ptr(X), L.ptr(lw), L.ptr(lb), 1e-5, L.ptr(W1), L.ptr(b1), L.ptr(W2), L.ptr(b2), M, C, H4, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M {M:7d} C {C:4d} H {H4:5d}: ln_mlp_fused_f16 {ms*1e3:8.1f} us  {4.0*M*C*H4/ms/1e9:7.1f} TF/s", flush=True)
    Y = torch.empty(M, C, device="cuda", dtype=torch.float16); W0 = (torch.randn(C, C, device="cuda") * 0.05).half()
    def runp(): L.call("xp_ln_proj_f16", L.ptr(X), L.ptr(lw), L.ptr(lb), 1e-5, L.ptr(W0), L.ptr(Y), M, C, st)
    for _ in range(3): runp()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): runp()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M {M:7d} C {C:4d}        : ln_proj_f16      {ms*1e3:8.1f} us  {4.0*M*C/ms/1e9:7.2f} TB/s", flush=True)
