"""debug build (XP_MLP_DBG=64): phase durations of the ping-pong fused MLP from s_memtime stamps of workgroup 0"""
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
for (M, C, H4) in [(307200, 96, 384), (76800, 192, 768)]:
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
    buf = (ctypes.c_ulonglong * (8 * 512))()
    assert lib.xp_mlp_debug_stamps(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 512).astype(np.int64)
    NC = H4 // 32
    for w in (0, 4):
        t = a[w]
        # stamps: 2 per barrier (entry, exit).  group 0: barriers after F1, V, F2 per chunk; group 1: one extra first, then F1, V, F2
        off = 0 if w < 4 else 2
        names = ["F1", "V", "F2"]
        work = {k: [] for k in names}; wait = {k: [] for k in names}
        for c in range(2, NC - 2):
            for ph in range(3):
                b = off + 2 * (3 * c + ph)          # entry stamp index of the barrier ending this phase
                prev_exit = b - 1
                work[names[ph]].append(t[b] - t[prev_exit]); wait[names[ph]].append(t[b + 1] - t[b])
        print(f"C {C} wave {w}: " + "  ".join(f"{k}: work {np.mean(work[k]):7.0f} + barrier wait {np.mean(wait[k]):7.0f}" for k in names) + f"   (ticks of s_memtime = 100 MHz?; chunk total {np.mean([sum(x) for x in zip(*[work[k] for k in names], *[wait[k] for k in names])]):.0f})", flush=True)
