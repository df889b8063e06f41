import numpy as np, torch
from xpoint_amd import synth
from xpoint_amd.utils import interpolate_descriptors
desc = torch.from_numpy(synth.uniform("kern/interp/big", (256, 60, 80), -1, 1))
ys = torch.arange(0, 480, 7); xs = torch.arange(0, 640, 11)
kp = torch.stack(torch.meshgrid(ys, xs, indexing="ij"), -1).reshape(-1, 2)
kp = torch.cat([kp, torch.tensor([[479, 639], [0, 639], [479, 0]])])
out = interpolate_descriptors(kp.cuda(), desc.cuda(), 480, 640)
np.save("gpurun_out/interp_gpu.npy", out.cpu().numpy())
