"""debug: fused h2 MLP (current schedule) vs the separate launches, by row block"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L
torch.manual_seed(0)
st = L.current_stream()
def split(W):
    N, K = W.shape
    o = torch.empty(L.load().xp_split_weights_h2_bytes(N, K), dtype=torch.uint8, device="cuda")
    L.call("xp_split_weights_h2", L.ptr(W), ctypes.c_void_p(o.data_ptr()), N, K, st)
    return o
for (M, C, H4, PROJ) in [(256, 96, 384, 1), (256, 96, 384, 0), (512, 192, 768, 1), (4096, 96, 384, 1), (65536, 96, 384, 1)]:
    X = torch.randn(M, C, device="cuda"); lw = torch.ones(C, device="cuda"); lb = torch.zeros(C, device="cuda")
    W1 = torch.randn(H4, C, device="cuda") * 0.05; b1 = torch.randn(H4, device="cuda") * 0.1
    W2 = torch.randn(C, H4, device="cuda") * 0.05; b2 = torch.randn(C, device="cuda") * 0.1
    W1x, W2x = split(W1), split(W2)
    T = torch.empty(M, C, device="cuda"); Hb = torch.empty(M, H4, device="cuda")
    p1, p2 = ctypes.c_void_p(W1x.data_ptr()), ctypes.c_void_p(W2x.data_ptr())
    W0 = torch.randn(C, C, device="cuda") * 0.05; W0x = split(W0); p0 = ctypes.c_void_p(W0x.data_ptr()); Tin = torch.randn(M, C, device="cuda")
    pack = torch.empty(L.load().xp_mlp_fused_h2_pack_bytes(C, H4, int(PROJ)), dtype=torch.uint8, device="cuda")
    pk = ctypes.c_void_p(pack.data_ptr())
    L.call("xp_mlp_fused_h2_pack", p1, p2, p0 if PROJ else None, pk, C, H4, st)
    Xa = X.clone(); Xb = X.clone()
    L.call("xp_mlp_fused_h2", L.ptr(Xa), L.ptr(Tin) if PROJ else None, L.ptr(lw), L.ptr(lb), pk, p1, p2, p0 if PROJ else None, L.ptr(b1), L.ptr(b2), M, C, H4, 1e-5, st)
    if PROJ: L.call("xp_gemm_nt_h2", L.ptr(Tin), p0, L.ptr(Xb), None, None, None, L.ptr(Xb), M, C, C, C, C, C, 0, st)
    L.call("xp_layernorm", L.ptr(Xb), L.ptr(T), L.ptr(lw), L.ptr(lb), M, C, 1e-5, 0, st)
    L.call("xp_gemm_nt_h2", L.ptr(T), p1, L.ptr(Hb), L.ptr(b1), None, None, None, M, H4, C, C, H4, 0, 1, st)
    L.call("xp_gemm_nt_h2", L.ptr(Hb), p2, L.ptr(Xb), L.ptr(b2), None, None, L.ptr(Xb), M, C, H4, H4, C, C, 0, st)
    torch.cuda.synchronize()
    d = (Xa - Xb).abs()
    per_block = d.view(M // 32, 32, C).amax(dim=(1, 2))
    bad = (per_block > 1e-4).nonzero().flatten().tolist()
    print(f"M {M} C {C} proj {PROJ}: max diff {float(d.max()):.3e}; bad 32-row blocks {len(bad)} of {M // 32}: {bad[:24]}  (block % 8: {sorted(set(b % 8 for b in bad))})", flush=True)
