"""Secondary measurement: full-pol C3 omnibus on one GPU's share of BASELINE config 4
(48 dates x 8192 x 8192 split over 8 GPUs = 1024 rows x 8192 columns per GPU)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
k, ny, nx = 48, 1024, 8192
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(3)
st = synth.empty_stack(9, k, ny, nx, dev)
# diagonally dominant Hermitian matrices with Wishart-like noise
for t in range(k):
    for c in range(9):
        noise = torch.randn((ny, nx), generator=g, device=dev)
        st[c, t] = (1.0 + 0.3 * noise) if c < 3 else 0.1 * noise
mask = torch.rand((ny, nx), generator=g, device=dev) < 0.01
st[:3, k // 2:] = torch.where(mask, st[:3, k // 2:] * 4.0, st[:3, k // 2:])
planes = [st[c] for c in range(9)]
for _ in range(2): out = kernels.change_detection_c3(planes, alpha=0.99, n=9)
torch.cuda.synchronize()
_lib.timing_enable(64)
t0 = time.perf_counter()
for _ in range(5): out = kernels.change_detection_c3(planes, alpha=0.99, n=9)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
kt = _lib.timing_collect(); by = {}
for n_, ms in kt: by.setdefault(n_, []).append(ms)
avg = {n_: sum(v) / len(v) for n_, v in by.items()}
bytes_ = ny * nx * k * 9 * 4
print(json.dumps({'workload': 'omnibus C3 %dt x %d x %d f32' % (k, ny, nx), 'ms': dt * 1e3, 'Mpx_per_s': ny * nx / dt / 1e6,
                  'kernels_ms': avg, 'passA_GBps': bytes_ / (avg['omnibus_c2_global'] * 1e-3) / 1e9,
                  'flagged': float((out.sum(dim=2) > 0).float().mean().item())}))
