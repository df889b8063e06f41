import os, sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import xpoint_oracle as xo
from xpoint_amd import synth
H, W = 480, 640
cfg = synth.xpoint_exp1_config(H, W)
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}
data = synth.to_torch(synth.make_pair_batch(0, 1, H, W))
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        t0 = time.perf_counter(); xo.predict_align_image_pair(data, sd); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); xo.predict_align_image_pair(data, sd); dt2 = time.perf_counter() - t0
    print(th, "threads:", round(dt, 2), round(dt2, 2), "s/pair", flush=True)
