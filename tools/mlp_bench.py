"""Micro-benchmark: fused VSS-block MLP (xp_mlp_fused_x3) vs the three launches it replaces, at the model's shapes."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L
shapes = [(307200, 96, 384), (76800, 192, 768), (76800, 64, 256), (19200, 96, 384)]
if os.environ.get("MLP_ONLY"):
    shapes = [shapes[int(i)] for i in os.environ["MLP_ONLY"].split(",")]
FUSED_ONLY = os.environ.get("MLP_FUSED_ONLY") == "1"
H2 = os.environ.get("MLP_H2") == "1"            # the split-fp16 instances (default engine of the model)
ENG = "h2" if H2 else "x3"
torch.manual_seed(0)
st = L.current_stream()
def split(W):
    N, K = W.shape
    o = torch.empty(getattr(L.load(), f"xp_split_weights_{ENG}_bytes")(N, K), dtype=torch.uint8, device="cuda")
    L.call(f"xp_split_weights_{ENG}", L.ptr(W), ctypes.c_void_p(o.data_ptr()), N, K, st)
    return o
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, C, H4) in shapes:
    X = torch.randn(M, C, device="cuda"); lw = torch.ones(C, device="cuda"); lb = torch.zeros(C, device="cuda")
    W1 = torch.randn(H4, C, device="cuda") * 0.05; b1 = torch.randn(H4, device="cuda") * 0.1
    W2 = torch.randn(C, H4, device="cuda") * 0.05; b2 = torch.randn(C, device="cuda") * 0.1
    W1x, W2x = split(W1), split(W2)
    T = torch.empty(M, C, device="cuda"); Hb = torch.empty(M, H4, device="cuda")
    p1, p2 = ctypes.c_void_p(W1x.data_ptr()), ctypes.c_void_p(W2x.data_ptr())
    PROJ = os.environ.get("MLP_PROJ", "1") == "1"      # also fold x += t W0^T (out_proj + first residual) into the launch
    W0 = torch.randn(C, C, device="cuda") * 0.05; W0x = split(W0); p0 = ctypes.c_void_p(W0x.data_ptr()); Tin = torch.randn(M, C, device="cuda")
    pack = torch.empty(getattr(L.load(), f"xp_mlp_fused_{ENG}_pack_bytes")(C, H4, int(PROJ)), dtype=torch.uint8, device="cuda")
    pk = ctypes.c_void_p(pack.data_ptr())
    L.call(f"xp_mlp_fused_{ENG}_pack", p1, p2, p0 if PROJ else None, pk, C, H4, st)
    def fused():
        if H2: L.call("xp_mlp_fused_h2", L.ptr(X), L.ptr(Tin) if PROJ else None, L.ptr(lw), L.ptr(lb), pk, p1, p2, p0 if PROJ else None, L.ptr(b1), L.ptr(b2), M, C, H4, 1e-5, st)
        else: L.call("xp_mlp_fused_x3", L.ptr(X), L.ptr(Tin) if PROJ else None, L.ptr(lw), L.ptr(lb), pk, L.ptr(b1), L.ptr(b2), M, C, H4, 1e-5, st)
    def three():
        if PROJ: L.call(f"xp_gemm_nt_{ENG}", L.ptr(Tin), p0, L.ptr(X), None, None, None, L.ptr(X), M, C, C, C, C, C, 0, st)
        L.call("xp_layernorm", L.ptr(X), L.ptr(T), L.ptr(lw), L.ptr(lb), M, C, 1e-5, 0, st)
        L.call(f"xp_gemm_nt_{ENG}", L.ptr(T), p1, L.ptr(Hb), L.ptr(b1), None, None, None, M, H4, C, C, H4, 0, 1, st)
        L.call(f"xp_gemm_nt_{ENG}", L.ptr(Hb), p2, L.ptr(X), L.ptr(b2), None, None, L.ptr(X), M, C, H4, H4, C, C, 0, st)
    if os.environ.get("MLP_CRC") == "1":      # one call on fresh inputs: CRC of the updated rows (schedule A/Bs must not change a bit)
        import zlib
        X0 = X.clone(); fused(); torch.cuda.synchronize()
        print(f"CRC M {M} C {C}: {zlib.crc32(X.cpu().numpy().tobytes()):08x}  finite {bool(torch.isfinite(X).all())}", flush=True)
        X.copy_(X0)
    tf = timeit(fused); t3 = tf if FUSED_ONLY else timeit(three)
    fl = 4.0 * M * C * H4 + (2.0 * M * C * C if PROJ else 0.0)
    print(f"M {M:7d} C {C:4d} H {H4:5d}: fused {tf*1e3:8.1f} us {fl/tf/1e9:7.1f} TF/s | separate launches {t3*1e3:8.1f} us {fl/t3/1e9:7.1f} TF/s | x{t3/tf:.2f}", flush=True)
