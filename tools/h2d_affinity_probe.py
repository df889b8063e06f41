"""Does restricting the CPU affinity BEFORE the HIP runtime loads change pinned host <-> device copy rates?  (bench.py's per-rank pinning halved the
PCIe-inclusive rate on the pool: this isolates the copy.)   usage: python tools/h2d_affinity_probe.py [cpulist | none] [before | after]"""
import os, sys, time
cpus = sys.argv[1] if len(sys.argv) > 1 else "none"
when = sys.argv[2] if len(sys.argv) > 2 else "before"
def pin():
    if cpus != "none":
        out = []
        for part in cpus.split(","):
            a, _, b = part.partition("-"); out += list(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, out)
if when == "before": pin()
import torch
torch.cuda.init()
if when == "after": pin()
dev = torch.device("cuda")
for mb in (20, 64):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"affinity {cpus} set {when} runtime load: {mb} MB  H2D {20 * mb / 1024 / (t1 - t0):6.1f} GB/s  D2H {20 * mb / 1024 / (t2 - t1):6.1f} GB/s  (mask now {len(os.sched_getaffinity(0))} cpus)")
