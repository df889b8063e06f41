"""Are the split-fp16 engines / tiles bit-identical on the same inputs?  Each variant runs in a child process (the knobs are read once); outputs are CRC'd."""
import os, sys, subprocess, zlib
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import ctypes, torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from xpoint_amd import _lib as L
    torch.manual_seed(0); st = L.current_stream(); out = []
    # conv 3x3 stride 2: (B, 30, 40, 384) -> (B, 15, 20, 768) for B = 2 and 16; and stride 1 head conv
    for (B, H, W, Ci, Co, stride) in [(2, 30, 40, 384, 768, 2), (16, 30, 40, 384, 768, 2), (4, 60, 80, 48, 512, 1)]:
        x = torch.randn(B, H, W, Ci, device="cuda"); w = torch.randn(Co, 9 * Ci, device="cuda") * 0.02; b = torch.randn(Co, device="cuda")
        Wx = torch.empty(L.load().xp_split_weights_h2_bytes(Co, 9 * Ci), dtype=torch.uint8, device="cuda")
        L.call("xp_split_weights_h2", L.ptr(w), ctypes.c_void_p(Wx.data_ptr()), Co, 9 * Ci, st)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty(B, Ho, Wo, Co, device="cuda")
        L.call("xp_conv3x3_nhwc_h2", L.ptr(x), ctypes.c_void_p(Wx.data_ptr()), L.ptr(y), L.ptr(b), None, None, B, H, W, Ci, Co, stride, 0, 0, st)
        torch.cuda.synchronize(); out.append(zlib.crc32(y.cpu().numpy().tobytes()))
    for (M, N, K) in [(4800, 200, 768), (600, 200, 768), (19200, 104, 384), (4800, 768, 768), (300, 768, 768)]:
        A = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05
        Wx = torch.empty(L.load().xp_split_weights_h2_bytes(N, K), dtype=torch.uint8, device="cuda")
        L.call("xp_split_weights_h2", L.ptr(w), ctypes.c_void_p(Wx.data_ptr()), N, K, st)
        C = torch.empty(M, N, device="cuda")
        L.call("xp_gemm_nt_h2", L.ptr(A), ctypes.c_void_p(Wx.data_ptr()), L.ptr(C), None, None, None, None, M, N, K, K, N, N, 0, st)
        torch.cuda.synchronize(); out.append(zlib.crc32(C.cpu().numpy().tobytes()))
    print("CRC", " ".join(f"{v:08x}" for v in out)); sys.exit(0)
VARIANTS = [{}, {"XP_H2_NO64": "0"}, {"XP_H2_NO64": "1"}, {"XP_H2_ENGINE": "rs"}, {"XP_H2_ENGINE": "lds"}, {"XP_H2_ENGINE": "lds", "XP_H2_TILE": "4"},
            {"XP_H2_ENGINE": "lds", "XP_H2_TILE": "3"}, {"XP_H2_ENGINE": "lds", "XP_H2_TILE": "1"}, {"XP_H2P": "0"}]


def run_variants():
    """[(env, crc line or None)] — one child process per knob setting (the library reads the knobs once)."""
    res = []
    for env in VARIANTS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("CRC")]
        res.append((env, lines[0] if lines else None, r.stderr[-300:]))
    return res


if __name__ == "__main__":
    for env, crc, err in run_variants():
        print(f"{str(env):55s}", crc or err)
