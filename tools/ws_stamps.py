"""debug build (-DXP_WS_DBG=64): where the phases of the warp-specialised fused tail go — s_memtime stamps of matrix wave 0 / vector wave 4 of workgroup 0"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L
import numpy as np
torch.manual_seed(0)
st = L.current_stream()
lib = ctypes.CDLL(L.LIB_PATH)
def split(W):
    N, K = W.shape
    o = torch.empty(L.load().xp_split_weights_h2_bytes(N, K), dtype=torch.uint8, device="cuda")
    L.call("xp_split_weights_h2", L.ptr(W), ctypes.c_void_p(o.data_ptr()), N, K, st)
    return o
for (M, C, H4) in [(307200, 96, 384)]:
    X = torch.randn(M, C, device="cuda"); lw = torch.ones(C, device="cuda"); lb = torch.zeros(C, device="cuda")
    W1 = torch.randn(H4, C, device="cuda") * 0.05; b1 = torch.randn(H4, device="cuda") * 0.1
    W2 = torch.randn(C, H4, device="cuda") * 0.05; b2 = torch.randn(C, device="cuda") * 0.1
    W1x, W2x = split(W1), split(W2)
    p1, p2 = ctypes.c_void_p(W1x.data_ptr()), ctypes.c_void_p(W2x.data_ptr())
    W0 = torch.randn(C, C, device="cuda") * 0.05; W0x = split(W0); p0 = ctypes.c_void_p(W0x.data_ptr()); Tin = torch.randn(M, C, device="cuda")
    pack = torch.empty(L.load().xp_mlp_fused_h2_pack_bytes(C, H4, 1), dtype=torch.uint8, device="cuda")
    pk = ctypes.c_void_p(pack.data_ptr())
    L.call("xp_mlp_fused_h2_pack", p1, p2, p0, pk, C, H4, st)
    for _ in range(3):
        L.call("xp_mlp_fused_h2", L.ptr(X), L.ptr(Tin), L.ptr(lw), L.ptr(lb), pk, p1, p2, p0, L.ptr(b1), L.ptr(b2), M, C, H4, 1e-5, st)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (2 * 4096))()
    assert lib.xp_mlp_ws_debug_stamps(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 4096).astype(np.int64)
    NC = H4 // 32; NQ = NC + 2
    m = a[0][:4 * NQ].reshape(NQ, 4); v = a[1][:4 * NQ].reshape(NQ, 4)
    print("phase | M: start->fc1 done, ->fc2 done, ->H written + DMA landed, barrier wait | V: H+scale read, GELU+split, P write, barrier wait   (s_memtime ticks)")
    for q in range(NQ - 1):
        mm = [m[q][1] - m[q][0], m[q][2] - m[q][1], m[q][3] - m[q][2], m[q + 1][0] - m[q][3]]
        vv = [v[q][1] - v[q][0], v[q][2] - v[q][1], v[q][3] - v[q][2], v[q + 1][0] - v[q][3]]
        print(f"{q:3d}   | {mm[0]:6d} {mm[1]:6d} {mm[2]:6d} {mm[3]:6d}  (phase {m[q+1][0]-m[q][0]:6d}) | {vv[0]:6d} {vv[1]:6d} {vv[2]:6d} {vv[3]:6d}  (phase {v[q+1][0]-v[q][0]:6d})   M start - V start {m[q][0]-v[q][0]:6d}")
