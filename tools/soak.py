"""Soak run of the overlapped pair pipeline (three encoders in flight): N bursts (default 400) of 1..7 back-to-back calls on the same batch; the keypoint and
match lists fetched after every burst must hash identically.   usage: python tools/soak.py [iterations] [gemm_mode, e.g. amp16f]"""
import sys, zlib, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from xpoint_amd import synth, models
from xpoint_amd.predict import PairPipeline
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
H, W, B = 480, 640, 8
cfg = synth.xpoint_exp1_config(H, W)
net = models.XPoint(cfg); net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True); net = net.to("cuda").eval()
if len(sys.argv) > 2: net.gemm_mode = sys.argv[2]
data = synth.to_torch(synth.make_pair_batch(0, B, H, W), "cuda")
pipe = PairPipeline(net, B, H, W, cap=8192, overlap=True, alternate_encoders=3)
def digest(res):
    h = 0
    for r in res:
        for k in ("kp_optical", "kp_thermal"):
            h = zlib.crc32(np.ascontiguousarray(r[k].numpy()).astype("<i4").tobytes(), h)
        h = zlib.crc32(np.ascontiguousarray(np.stack([r["match_q"], r["match_t"]], 1)).astype("<i4").tobytes(), h)
    return h
ref = None; t0 = time.time(); bad = 0
with torch.no_grad():
    for i in range(N):
        for _ in range(1 + i % 7):          # bursts of 1..7 calls in flight behind each other (the buffer sets rotate), then the last call's results
            pipe.run(data["optical"]["image"], data["thermal"]["image"], data["optical"]["valid_mask"], data["thermal"]["valid_mask"])
        d = digest(pipe.fetch())
        if ref is None: ref = d
        elif d != ref: bad += 1; print(f"iteration {i}: digest {d:#x} != {ref:#x}")
print(f"[{net.gemm_mode}] {N} iterations in {time.time() - t0:.1f} s, digest {ref:#x}, {bad} differing")
sys.exit(1 if bad else 0)
