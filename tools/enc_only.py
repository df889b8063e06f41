"""How much of the overlapped step is the encoder?  16-image forwards (C2) on 1 / 2 / 3 alternating streams, no detection / matching, eager launches
enqueued without any host synchronisation inside the timed loop (round 4's version read the forward's status word after every call, so it timed the host's
enqueue + wait, not the GPU: 1 592 "encoder-only" pairs/s BELOW the full step's 1 749)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import models, synth
H, W, n = 480, 640, 16
cfg = synth.xpoint_exp1_config(H, W)
net = models.XPoint(cfg); net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True); net.to("cuda").eval()
img = torch.rand(n, 1, H, W, device="cuda")
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    ws = [torch.empty(net.workspace_bytes(n, H, W), dtype=torch.uint8, device="cuda") for _ in range(S)]
    outs = [None] * S
    def step(i):
        k = i % S
        with torch.cuda.stream(streams[k]):
            outs[k] = net.forward_raw(img, want_prob=True, want_desc=True, out=outs[k], workspace=ws[k], check=False)      # stream-ordered: no host read of the status word per call
    with torch.no_grad():
        for i in range(2 * S): step(i)
        torch.cuda.synchronize()
        K = 60
        t0 = time.perf_counter()
        for i in range(K): step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
    print(f"{S} stream(s): {dt*1e3:.3f} ms per 16-image forward = {8/dt:.1f} pairs/s encoder-only", flush=True)
