import sys, time, torch
sys.path.insert(0, '.')
from nd_amd import kernels, _lib
x = torch.rand((24, 4096, 4096), device='cuda') + 0.5
y = torch.empty_like(x)
for mode in ('plain', 'timing'):
    if mode == 'timing': _lib.timing_enable(1000)
    for _ in range(20): kernels.gaussian_filter(x, (0, 1.0, 1.0), out=y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): kernels.gaussian_filter(x, (0, 1.0, 1.0), out=y)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(mode, 'host ms/call', (t1 - t0) / 20 * 1e3, 'total ms/call', (t2 - t0) / 20 * 1e3)
