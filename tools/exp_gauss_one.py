"""Tuning experiment (GPU box): three Gaussian passes along one axis of a 24 x 4096 x 4096 stack, for
rocprofv3 counter runs:  python tools/exp_gauss_one.py <axis> <sigma>."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nd_amd import kernels
ax = int(sys.argv[1]); sg = float(sys.argv[2])
x = torch.rand((24, 4096, 4096), device='cuda') + 0.5
out = torch.empty_like(x)
sig = [0, 0, 0]; sig[ax] = sg
for _ in range(3): kernels.gaussian_filter(x, sig, out=out)
torch.cuda.synchronize()
