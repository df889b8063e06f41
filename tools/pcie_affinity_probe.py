"""Which part of the PCIe-inclusive step slows down when the CPU affinity is restricted before torch loads?   usage: python tools/pcie_affinity_probe.py [cpulist|none]"""
import os, sys, time
cpus = sys.argv[1] if len(sys.argv) > 1 else "none"
if cpus != "none":
    out = []
    for part in cpus.split(","):
        a, _, b = part.partition("-"); out += list(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, out)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xpoint_amd import models, synth
from xpoint_amd.predict import PairPipeline
H, W, B = 480, 640, 8
cfg = synth.xpoint_exp1_config(H, W)
net = models.XPoint(cfg); net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True); net = net.to("cuda").eval()
d = synth.to_torch(synth.make_pair_batch(0, B, H, W), "cuda")
o, t, mo, mt = d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"]
ho, ht = o.cpu().pin_memory(), t.cpu().pin_memory()
pipe = PairPipeline(net, B, H, W, cap=8192, overlap=True, alternate_encoders=3)
def loop(host_in, download, n=60):
    prev = None
    with torch.no_grad():
        for _ in range(5): pipe.run(o, t, mo, mt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            pipe.run(ho if host_in else o, ht if host_in else t, mo, mt)
            if download:
                bufs, ev = pipe.download_async()
                if prev is not None: prev.synchronize()
                prev = ev
        torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)
print(f"affinity {cpus} ({len(os.sched_getaffinity(0))} cpus, torch threads {torch.get_num_threads()}): device in, no download {loop(False, False):7.1f} | host in {loop(True, False):7.1f} | "
      f"download {loop(False, True):7.1f} | both {loop(True, True):7.1f} pairs/s")
