import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd import synth
from xpoint_amd.utils import box_nms
from oracle import xpoint_oracle as xo
for name, shape, size, levels in [("ties", (1, 1, 40, 56), 8, 16), ("size4", (1, 1, 33, 47), 4, 0), ("batch", (3, 1, 32, 48), 8, 64)]:
    p = synth.uniform("nms/" + name, shape, 0.0, 1.0)
    if levels:
        p = (np.floor(p * levels) / levels).astype(np.float32)
    pt = torch.from_numpy(p)
    ref = xo.box_nms(pt, size, 0.3)
    out = box_nms(pt.cuda(), size, 0.3).cpu()
    d = (out != ref)
    print(name, "mismatch", int(d.sum()), "kept ref", int((ref > 0).sum()), "kept out", int((out > 0).sum()))
    idx = torch.nonzero(d)
    for i in idx[:8]:
        b, _, y, x = i.tolist()
        print("  at", (y, x), "ref", float(ref[b, 0, y, x]), "out", float(out[b, 0, y, x]), "p", float(pt[b, 0, y, x]))
