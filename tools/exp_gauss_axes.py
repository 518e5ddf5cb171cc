"""Tuning experiment (GPU box): time of one Gaussian pass per axis (correlate1d kernels)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nd_amd import kernels, _lib
x = torch.rand((24, 4096, 4096), device='cuda') + 0.5
out = torch.empty_like(x)
for sig in [(0, 1, 0), (0, 0, 1), (0, 2.5, 0), (0, 0, 2.5), (1, 0, 0)]:
    for _ in range(2): kernels.gaussian_filter(x, sig, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): kernels.gaussian_filter(x, sig, out=out)
    torch.cuda.synchronize(); print(sig, (time.perf_counter() - t0) / 5 * 1e3, 'ms')
