"""Micro-benchmark of xp_gemm_nt on the model's shapes (B=16 images, 480x640)."""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L
shapes = [  # (M, N, K, act, res)
    (307200, 96, 96, 0, 0), (307200, 96, 96, 0, 1), (307200, 384, 96, 1, 0), (307200, 384, 96, 0, 0), (307200, 96, 384, 0, 1),
    (307200, 32, 96, 0, 0),
    (76800, 192, 192, 0, 0), (76800, 768, 192, 1, 0), (76800, 768, 192, 0, 0), (76800, 192, 768, 0, 1),
    (19200, 384, 384, 0, 0), (19200, 1536, 384, 1, 0), (19200, 384, 1536, 0, 1),
    (4800, 768, 768, 0, 0), (4800, 3072, 768, 1, 0), (4800, 768, 3072, 0, 1),
    (76800, 65, 256, 0, 0), (76800, 256, 256, 0, 0),
    (19200, 96, 96, 0, 0), (65536, 96, 96, 0, 0),
    (10240, 384, 1536, 0, 0),   # 20: 240 workgroups of 128x128 = one per CU: the single-workgroup K-loop timeline
]
import os
if os.environ.get('GB_SHAPES'):      # "M,N,K,act,res;..."
    shapes = [tuple(int(v) for v in t.split(',')) for t in os.environ['GB_SHAPES'].split(';')]
if os.environ.get('GB_ONLY'):
    idx=[int(i) for i in os.environ['GB_ONLY'].split(',')]; shapes=[shapes[i] for i in idx]
torch.manual_seed(0)
for (M, N, K, act, res) in shapes:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    C = torch.empty(M, N, device="cuda"); R = torch.randn(M, N, device="cuda") if res else None
    st = L.current_stream()
    X3 = os.environ.get("GB_X3", "0") == "1"
    H2 = os.environ.get("GB_H2", "0") == "1"
    F16 = os.environ.get("GB_F16", "0") == "1"      # fp16-storage kernel (csrc/gemm_f16.hip): half A / W / C / res
    if F16:
        A16, W16, C16 = A.half(), W.half(), torch.empty(M, N, device="cuda", dtype=torch.float16)
        R16 = R.half() if res else None
    if H2:
        import ctypes
        Wx = torch.empty(L.load().xp_split_weights_h2_bytes(N, K), dtype=torch.uint8, device="cuda")
        L.call("xp_split_weights_h2", L.ptr(W), ctypes.c_void_p(Wx.data_ptr()), N, K, st)
        wxp = ctypes.c_void_p(Wx.data_ptr())
    elif X3:
        import ctypes
        Wx = torch.empty(L.load().xp_split_weights_x3_bytes(N, K), dtype=torch.uint8, device="cuda")
        L.call("xp_split_weights_x3", L.ptr(W), ctypes.c_void_p(Wx.data_ptr()), N, K, st)
        wxp = ctypes.c_void_p(Wx.data_ptr())
    def run():
        if F16:
            L.call("xp_gemm_nt_f16", L.ptr(A16), L.ptr(W16), L.ptr(C16), 0, L.ptr(b), None, None, L.ptr(R16), M, N, K, K, N, N, act, st)
        elif H2:
            L.call("xp_gemm_nt_h2", L.ptr(A), wxp, L.ptr(C), L.ptr(b), None, None, L.ptr(R), M, N, K, K, N, N, act, st)
        elif X3:
            L.call("xp_gemm_nt_x3", L.ptr(A), wxp, L.ptr(C), L.ptr(b), None, None, L.ptr(R), M, N, K, K, N, N, act, st)
        else:
            L.call("xp_gemm_nt", L.ptr(A), L.ptr(W), L.ptr(C), L.ptr(b), None, None, L.ptr(R), M, N, K, K, N, N, act, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * M * N * K
    by = (2.0 if F16 else 4.0) * (M * K + N * K + M * N * (2 if res else 1))
    print(f"M{M:7d} N{N:5d} K{K:5d} act{act} res{res}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  {by/ms/1e6:7.0f} GB/s", flush=True)
