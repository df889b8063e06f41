"""How long does the host take to ENQUEUE one pipeline step (vs the GPU time per step)?  If the two are close the GPU starves."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import models, synth
from xpoint_amd.predict import PairPipeline
H, W, B = 480, 640, 8
cfg = synth.xpoint_exp1_config(H, W)
net = models.XPoint(cfg); net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True); net = net.to("cuda").eval()
d = synth.to_torch(synth.make_pair_batch(0, B, H, W), "cuda")
o, t, mo, mt = d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"]
for ov, S in ((False, 0), (True, 0), (True, 2)):
    pipe = PairPipeline(net, B, H, W, cap=8192, overlap=ov, split_encoder=S)
    with torch.no_grad():
        for _ in range(3):
            pipe.run(o, t, mo, mt)
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            pipe.run(o, t, mo, mt)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    print(f"overlap={ov} split={S}: host enqueue {t_enq / n * 1e3:.2f} ms/step, wall {t_all / n * 1e3:.2f} ms/step")
