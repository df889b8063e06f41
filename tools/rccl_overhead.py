"""Where does a live RCCL communicator cost the overlapped pipeline its 5-9 %?  One process, same pipeline, rate measured
(1) before init_process_group, (2) after init (no collective yet), (3) after the first collective, (4) after destroy."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from xpoint_amd import models, synth
from xpoint_amd.predict import PairPipeline

H, W, B = 480, 640, 8
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = synth.xpoint_exp1_config(H, W)
net = models.XPoint(cfg).eval()
net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True)
net.to(dev)
data = synth.to_torch(synth.make_pair_batch(0, B, H, W), dev)
opt, thr = data["optical"]["image"], data["thermal"]["image"]
mo, mt = data["optical"]["valid_mask"], data["thermal"]["valid_mask"]
pipe = PairPipeline(net, B, H, W, cap=8192, overlap=True, split_encoder=2)


def rate(tag, steps=30):
    with torch.no_grad():
        for _ in range(3):
            pipe.run(opt, thr, mo, mt)
        torch.cuda.synchronize()
        best = 0
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                pipe.run(opt, thr, mo, mt)
            torch.cuda.synchronize()
            best = max(best, B * steps / (time.perf_counter() - t0))
    print(f"{tag:40s} {best:8.1f} pairs/s", flush=True)


rate("before init")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=dev)
rate("after init_process_group")
x = torch.zeros(1024, device=dev)
dist.broadcast(x, 0); torch.cuda.synchronize()
rate("after first broadcast")
dist.barrier(); torch.cuda.synchronize()
rate("after barrier")
dist.destroy_process_group()
rate("after destroy")
