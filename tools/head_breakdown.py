"""Per-kernel HIP-event breakdown of the C5 head's forward alone (16 crops of 256 x 256 through the encoder + RegNet head): where its ~110 small launches spend
their time.   python tools/head_breakdown.py [pairs]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import _lib as L, models, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = synth.xpoint_exp1_config(256, 256, hm_head=True)
net = models.XPoint(cfg).eval(); net.load_state_dict(synth.make_torch_state_dict(cfg), strict=True); net.to("cuda")
d = synth.to_torch(synth.make_pair_batch(0, B, 256, 256), "cuda")
o, t = d["optical"]["image"], d["thermal"]["image"]
lib = L.load()
with torch.no_grad():
    for _ in range(3): net.predict_homography(o, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): net.predict_homography(o, t)
    torch.cuda.synchronize()
    print(f"head forward, {B} pairs of 256x256 crops: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call (eager, one stream)")
    lib.xp_prof_reset(); lib.xp_prof_filter(None); lib.xp_prof_enable(1)
    net.predict_homography(o, t); torch.cuda.synchronize()
    lib.xp_prof_enable(0)
name = ctypes.create_string_buffer(64); ms = ctypes.c_double(); cnt = ctypes.c_int(); fl = ctypes.c_double(); by = ctypes.c_double()
rows = []
for i in range(lib.xp_prof_count()):
    lib.xp_prof_get(i, name, 64, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by))
    rows.append((ms.value, name.value.decode(), cnt.value, fl.value))
tot = sum(r[0] for r in rows)
for m, n, c, f in sorted(rows, reverse=True):
    print(f"  {n:30s} {m * 1e3:8.1f} us {100 * m / tot:5.1f}%  x{c:3d}" + (f"  {f / m / 1e9:7.1f} TF/s" if f > 0 and m > 0 else ""))
print(f"  {'sum':30s} {tot * 1e3:8.1f} us")
