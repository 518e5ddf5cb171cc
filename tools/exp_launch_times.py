"""Per-launch kernel times of the register-window filters over 40 back-to-back launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from nd_amd import kernels, _lib
dev = torch.device('cuda')
x = torch.rand((24, 4096, 4096), device=dev) + 0.5
y = torch.empty_like(x)
for name, fn in (('boxcar3', lambda: kernels.convolve(x, np.ones((1, 3, 3)) / 9.0, out=y)),
                 ('boxcar5', lambda: kernels.convolve(x, np.ones((1, 5, 5)) / 25.0, out=y)),
                 ('gauss1', lambda: kernels.gaussian_filter(x, (0, 1.0, 1.0), out=y)),
                 ('add', None)):
    if fn is None:
        ts = []
        for _ in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); torch.add(x, 1.0, out=y); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    else:
        _lib.timing_enable(256)
        for _ in range(40): fn()
        torch.cuda.synchronize()
        ts = [ms for _, ms in _lib.timing_collect()]
        _lib.timing_enable(0)
    print(name, ' '.join('%.3f' % t for t in ts), flush=True)
