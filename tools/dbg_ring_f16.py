import ctypes, torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch.nn.functional as F
from xpoint_amd import _lib as L, synth
M, N, K = 19200, 384, 384
u = lambda n, s, lo=-1.0, hi=1.0: torch.from_numpy(synth.uniform(n, s, lo, hi))
A = u(f"fA{M}{N}{K}", (M, K)).half(); Wt = u(f"fW{M}{N}{K}", (N, K), -0.1, 0.1).half(); bias = u(f"fb{M}{N}{K}", (N,))
acc = F.linear(A.double(), Wt.double())
v = (acc + bias.double()).float().half()
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
Ad, Wd, bd = A.cuda(), Wt.cuda(), bias.cuda()
C = torch.empty((M, N), device="cuda", dtype=torch.float16)
L.call("xp_gemm_nt_f16", vp(Ad), vp(Wd), vp(C), 0, L.ptr(bd), None, None, None, M, N, K, K, N, N, 0, L.current_stream())
got = C.cpu().float(); ref = v.float()
d = (got - ref).abs()
bad = (d > ref.abs().clamp_min(2.0**-14) * 2.0**-9).nonzero()
print(len(bad)); 
for r, c in bad[:40].tolist(): print(r, c, float(got[r, c]), float(ref[r, c]), float(acc[r, c] + bias[c].double()))
