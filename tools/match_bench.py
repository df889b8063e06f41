"""Matcher micro-benchmark: P pairs of n x n unit descriptors (D = 256) through xp_match_mnn, capacity `cap`.
usage: python tools/match_bench.py [n] [cap] [pairs] [iters]   (run under rocprofv3 --kernel-trace --stats for per-kernel times)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpoint_amd import _lib
from xpoint_amd.utils import MATCH_MODES
from xpoint_amd._lib import ptr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4060
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
base = torch.randn((P, 1, 256), device=dev, generator=g)
d1 = torch.nn.functional.normalize(base + 0.35 * torch.randn((P, cap, 256), device=dev, generator=g), dim=2).contiguous()
d2 = torch.nn.functional.normalize(base + 0.35 * torch.randn((P, cap, 256), device=dev, generator=g), dim=2).contiguous()
counts = torch.full((2 * P,), n, dtype=torch.int32, device=dev)
lib = _lib.load()
res = dict(idx12=torch.empty((P, cap), dtype=torch.int32, device=dev), dist12=torch.empty((P, cap), device=dev),
           idx21=torch.empty((P, cap), dtype=torch.int32, device=dev), dist21=torch.empty((P, cap), device=dev),
           mq=torch.empty((P, cap), dtype=torch.int32, device=dev), mt=torch.empty((P, cap), dtype=torch.int32, device=dev),
           md=torch.empty((P, cap), device=dev), mc=torch.zeros((P,), dtype=torch.int32, device=dev))
ws = torch.empty(lib.xp_match_workspace_bytes(P, cap, cap, 256), dtype=torch.uint8, device=dev)
def run():
    _lib.check(lib.xp_match_mnn(ptr(d1), ptr(d2), ptr(counts), 1, 0, P, P, cap, cap, 256, MATCH_MODES["strict_mnn"], ptr(res["idx12"]), ptr(res["dist12"]),
                                ptr(res["idx21"]), ptr(res["dist21"]), ptr(res["mq"]), ptr(res["mt"]), ptr(res["md"]), ptr(res["mc"]), ptr(ws), ws.numel(),
                                _lib.current_stream()), "xp_match_mnn")
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
st = torch.zeros(4, dtype=torch.int64, device=dev)
_lib.check(lib.xp_match_stats(ptr(ws), ptr(counts), 1, 0, P, P, cap, cap, 256, ptr(st), _lib.current_stream()), "xp_match_stats")
st = [int(v) for v in st.cpu()]
print(f"candidates per row / column: mean {st[0] / max(st[3], 1):.2f}, max {st[1]}, overflowed lists {st[2]} of {st[3]} (inline capacity {lib.xp_match_cand_cap()})")
print(f"n={n} cap={cap} pairs={P}: {dt * 1e6:.1f} us per call, {int(res['mc'].sum())} mutual matches, "
      f"{2 * P * n * n * 256 / dt / 1e12:.1f} TFLOP/s algorithmic")
