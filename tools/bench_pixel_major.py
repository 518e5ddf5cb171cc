"""Secondary measurement: nd_amd_omnibus_c2_pixel_major (inputs in the reference's (y, x, time) layout,
C12 interleaved complex) against the planar entry point on the same data."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
ap = argparse.ArgumentParser(); ap.add_argument('--alpha', type=float, default=0.99)
ALPHA = ap.parse_args().alpha
dev = torch.device('cuda:0')
k, ny, nx = 24, 4096, 4096
st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1234, device=dev, change_frac=0.01)
yxt = [st[v].permute(1, 2, 0).contiguous() for v in range(4)]
c12 = torch.complex(yxt[1], yxt[2])
ref = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=ALPHA, n=9)
run = lambda: kernels.change_detection_pixel_major(yxt[0], c12.real, c12.imag, yxt[3], alpha=ALPHA, n=9)
for _ in range(3): out = run()
torch.cuda.synchronize()
_lib.timing_enable(64); t0 = time.perf_counter()
for _ in range(10): out = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
by = {}
for n_, ms in _lib.timing_collect(): by.setdefault(n_, []).append(ms)
km = {n_: sum(v) / len(v) for n_, v in by.items()}
gb = ny * nx * k * 4 * 4 / 1e9
print(json.dumps({'workload': 'omnibus C2 pixel-major, 24t x 4096 x 4096 f32, C12 complex64', 'alpha': ALPHA, 'ms': dt * 1e3,
                  'Mpx_per_s': ny * nx / dt / 1e6, 'kernels_ms': km, 'passA_TBps': gb / km['omnibus_c2_global'],
                  'equal_to_planar': bool(torch.equal(out, ref))}))
