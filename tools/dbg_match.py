import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpoint_amd.utils import match_descriptors
from oracle import xpoint_oracle as xo
n1, n2, D = 257, 311, 256
rng = np.random.default_rng(0)
d1 = rng.standard_normal((n1, D)).astype(np.float32); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
d2 = rng.standard_normal((n2, D)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
res = match_descriptors(torch.from_numpy(d1).cuda().unsqueeze(0), torch.from_numpy(d2).cuda().unsqueeze(0))
torch.cuda.synchronize()
ws = res["_ws"].cpu().numpy()
a, b = n1, n2
aP, bP = 512, 320
off = 8 * a
rowkey = ws[off:off + 4 * aP].view(np.uint32); off += 4 * aP
colkey = ws[off:off + 4 * bP].view(np.uint32); off += 4 * bP
na = ws[off:off + 4 * a].view(np.float32); off += 4 * a
nb = ws[off:off + 4 * b].view(np.float32); off += 4 * b
rcnt = ws[off:off + 4 * a].view(np.int32); off += 4 * a
ccnt = ws[off:off + 4 * b].view(np.int32); off += 4 * b
off += 64 * (a + b)
base = res["_ws"].data_ptr()
off = ((base + off + 255) & ~255) - base
maxbits = ws[off:off + 4].view(np.float32); off += 256
imgA = ws[off:off + aP * 560].reshape(aP, 560); off += aP * 560
imgB = ws[off:off + bP * 560].reshape(bP, 560)
def unord(u):
    u = u.astype(np.uint32)
    pos = (u & 0x80000000) != 0
    out = np.where(pos, u & 0x7fffffff, ~u).astype(np.uint32)
    return out.view(np.float32)
print("maxbits", maxbits, "na[:4]", na[:4], "nb[:4]", nb[:4])
A16 = imgA[:, :544].copy().view(np.float16).astype(np.float64); B16 = imgB[:, :544].copy().view(np.float16).astype(np.float64)
print("A row0 first vals", A16[0, :4], "vs", d1[0, :4], "ext", A16[0, 256:262], "dead ext", A16[300, 256:262])
print("B row0 ext", B16[0, 256:262])
E = A16[:n1] @ B16[:n2].T
print("expected rowmax[:4]", E.max(1)[:4], "got", unord(rowkey[:4]), "raw", rowkey[:4])
print("expected colmax[:4]", E.max(0)[:4], "got", unord(colkey[:4]))
print("rcnt hist", np.bincount(rcnt), "ccnt hist", np.bincount(ccnt))
idx12, dist12, gap12, idx21, dist21 = xo.nn_both(d1, d2)
print("idx12 ok", np.array_equal(res["idx12"][0].cpu().numpy(), idx12), "idx21 ok", np.array_equal(res["idx21"][0].cpu().numpy(), idx21))
