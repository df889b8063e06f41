"""Latency of ONE 480x640 pair through the whole path (the reference's scripts run one pair per call): eager predict_align_image_pair (per-stage host
round trips, like the reference), PairPipeline eager, PairPipeline replayed from hipGraphs.   python tools/single_pair_latency.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import models, synth
from xpoint_amd.predict import PairPipeline, predict_align_image_pair
H, W = 480, 640
cfg = synth.xpoint_exp1_config(H, W)
net = models.XPoint(cfg); net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True); net = net.to("cuda").eval()
d = synth.to_torch(synth.make_pair_batch(0, 1, H, W), "cuda")
a = (d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"])
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    t_script = timeit(lambda: predict_align_image_pair(net, d), 10)
    pipe = PairPipeline(net, 1, H, W, cap=8192)
    t_pipe = timeit(lambda: (pipe.run(*a), torch.cuda.synchronize()))
    pipe_g = PairPipeline(net, 1, H, W, cap=8192)
    step = pipe_g.capture(*a)
    t_graph = timeit(lambda: (step(*a), torch.cuda.synchronize()))
print(f"one 480x640 pair, latency per call: predict_align_image_pair (host lists, DMatch objects) {t_script:.2f} ms | PairPipeline eager {t_pipe:.2f} ms | "
      f"PairPipeline hipGraph replay {t_graph:.2f} ms   (throughput path, 8 pairs per step, 3 streams: 0.57 ms per pair)")
