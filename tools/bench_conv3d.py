"""ConvolutionFilter with a kernel that extends along time: the tiled kernel's walk over the window's planes
(round 5) against the generic footprint kernel (ND_AMD_NO_TILED=1, the route until round 4).
    python tools/bench_conv3d.py ; ND_AMD_NO_TILED=1 python tools/bench_conv3d.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from nd_amd import _lib, kernels
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand((24, 4096, 4096), generator=g, device=dev) + 0.5
y = torch.empty_like(x)
rng = np.random.default_rng(0)
for name, k in (('boxcar 3x3x3', np.ones((3, 3, 3)) / 27.0), ('random 3x3x3', rng.normal(size=(3, 3, 3))),
                ('random 3x5x5', rng.normal(size=(3, 5, 5))), ('boxcar 5x5x5', np.ones((5, 5, 5)) / 125.0)):
    fn = lambda: kernels.convolve(x, k, out=y)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    _lib.timing_enable(64); t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    by = {}
    for n_, ms in _lib.timing_collect(): by.setdefault(n_, []).append(ms)
    _lib.timing_enable(0)
    print(json.dumps({'kernel': name, 'route': 'generic' if os.environ.get('ND_AMD_NO_TILED') else 'tiled', 'ms': round(dt * 1e3, 3),
                      'kernels_ms': {n_: round(sum(v) / len(v), 3) for n_, v in by.items()},
                      'M_px_t_per_s': round(x.numel() / dt / 1e6, 1)}), flush=True)
