"""Tuning experiment (GPU box): host <-> device copy rates (pageable, pinned, staged) and the cost of
a host-side transpose -- the numbers behind the host-dataset path of nd_amd/change.py and filters.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
dev = torch.device('cuda:0')
a = np.random.default_rng(0).random((2048, 2048, 24), dtype=np.float32)      # 403 MB
def T(name, fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print('%-40s %.1f ms  %.1f GB/s' % (name, dt * 1e3, a.nbytes / dt / 1e9))
T('torch.from_numpy(a).to(dev) pageable', lambda: torch.from_numpy(a).to(dev))
p = torch.from_numpy(a).pin_memory()
T('pinned .to(dev)', lambda: p.to(dev, non_blocking=True))
T('np.ascontiguousarray(transpose)', lambda: np.ascontiguousarray(a.transpose(2, 0, 1)))
d = torch.from_numpy(a).to(dev)
T('device .cpu() (D2H pageable)', lambda: d.cpu())
from nd_amd.streaming import _parallel_copy, _pinned_like
stage = _pinned_like(a.shape, torch.float32)
src = torch.from_numpy(a)
def staged():
    n = a.shape[0]; step = n // 16
    _parallel_copy([(stage[i:i + step], src[i:i + step]) for i in range(0, n, step)])
    return stage.to(dev, non_blocking=True)
T('threaded pageable->pinned + H2D', staged)
