"""xp_box_nms on the model's own heat maps (480x640, 16 images): time per call and the finisher's phase stamps (debug words of the workspace)."""
import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L, models, synth
H, W, B = 480, 640, 8
cfg = synth.xpoint_exp1_config(H, W)
net = models.XPoint(cfg); net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True); net.to("cuda").eval()
d = synth.to_torch(synth.make_pair_batch(0, B, H, W), "cuda")
with torch.no_grad():
    raw = net.forward_raw(torch.cat([d["optical"]["image"], d["thermal"]["image"]], 0))
prob = raw["prob"].contiguous(); n = 2 * B
lib = L.load()
ws = torch.zeros(lib.xp_box_nms_workspace_bytes(n, H, W, 8192), dtype=torch.uint8, device="cuda")
out = torch.empty_like(prob)
st = L.current_stream()
def run():
    L.check(lib.xp_box_nms(L.ptr(prob), L.ptr(out), L.ptr(ws), ws.numel(), n, H, W, 8.0, 0.015, 0.1, 0, 8192, int(os.environ.get("NMS_SWEEPS", "8")), None, st), "nms")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"xp_box_nms {n} images {H}x{W}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call; kept per image {int((out > 0).sum()) / n:.0f}")
info = ws[-(8 * n + 64) * 4:].view(torch.int32)[: 8 * n].cpu().numpy().reshape(n, 8)
print("per image [undecided after pass 1, table in global?, rounds, t_load, t_first, t_prefix, t_table, t_done] (100 MHz ticks = 10 ns):")
print(info[:4])
