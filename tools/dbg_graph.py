import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpoint_amd import models, synth
from xpoint_amd.predict import PairPipeline
H, W, B = 96, 128, 2
cfg = synth.xpoint_exp1_config(H, W)
net = models.XPoint(cfg); net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True); net.to("cuda").eval()
d = [synth.to_torch(synth.make_pair_batch(s, B, H, W), "cuda") for s in (0, 9, 4)]
args = lambda x: (x["optical"]["image"], x["thermal"]["image"], x["optical"]["valid_mask"], x["thermal"]["valid_mask"])
for split in (0, 2):
    with torch.no_grad():
        pipe = PairPipeline(net, B, H, W, cap=2048, overlap=True, split_encoder=split)
        rep = pipe.capture(*args(d[0]))
        for x in d[1:]:
            rep(*args(x))
            torch.cuda.synchronize()
            print("split", split, "counts", pipe.counts.tolist(), "prob sum", float(pipe.raw["prob"].sum()), flush=True)
            try:
                pipe.verify(); print("  verify ok")
            except Exception as e:
                print("  verify:", e)
print("---- counters")
import numpy as np
def counters(pipe):
    n = (2 * B * H * W + 255) // 256 * 256
    nt = (2 * B * ((H + 31) // 32) * ((W + 31) // 32) + 255) // 256 * 256
    c = pipe.nms_ws[n + nt: n + nt + 64 * 4].cpu().numpy().view(np.int32)
    return c[:7].tolist(), int(c[63])
with torch.no_grad():
    pipe = PairPipeline(net, B, H, W, cap=2048, overlap=True, split_encoder=2)
    rep = pipe.capture(*args(d[0]))
    for i in range(6):
        rep(*args(d[i % 3])); torch.cuda.synchronize()
        print("replay", i, "k", (pipe._call - 1) & 1, counters(pipe))
    for i in range(4):
        pipe.run(*args(d[i % 3])); torch.cuda.synchronize()
        print("eager", i, counters(pipe))
