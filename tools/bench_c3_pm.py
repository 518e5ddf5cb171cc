"""Full-pol OmnibusTest on data in the reference's (y, x, time) layout (48 dates x 1024 x 8192, config 4's share):
nd_amd_omnibus_c3_pixel_major (round 5) against the planar entry point on the same values and against the former
route for such data (transpose every variable, then the planar entry point).
    python tools/bench_c3_pm.py [--alpha 0.99]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
ap = argparse.ArgumentParser(); ap.add_argument('--alpha', type=float, default=0.99)
ap.add_argument('--k', type=int, default=48); ap.add_argument('--ny', type=int, default=1024); ap.add_argument('--nx', type=int, default=8192)
a = ap.parse_args()
dev = torch.device('cuda:0')
k, ny, nx = a.k, a.ny, a.nx
st = synth.wishart_c3_stack(k, ny, nx, looks=9, seed=4321, device=dev, change_frac=0.01)
planar = [st[c] for c in range(9)]
ref = kernels.change_detection_c3(planar, alpha=a.alpha, n=9)
yxt = [st[c].permute(1, 2, 0).contiguous() for c in range(9)]
cplx = [torch.complex(yxt[c], yxt[c + 1]) for c in (3, 5, 7)]
joint = yxt[:3] + [h for z in cplx for h in (z.real, z.imag)]


def timed(fn, reps=5):
    for _ in range(2): out = fn()
    torch.cuda.synchronize()
    _lib.timing_enable(256); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    by = {}
    for n_, ms in _lib.timing_collect(): by.setdefault(n_, []).append(ms)
    _lib.timing_enable(0)
    return dt * 1e3, {n_: round(sum(v) / reps, 4) for n_, v in by.items()}, out


def transposed():
    pl = synth.empty_stack(9, k, ny, nx, dev, torch.float32)
    for c in range(9):
        assert kernels.relayout_planar(yxt[c], pl[c])
    return kernels.change_detection_c3([pl[c] for c in range(9)], alpha=a.alpha, n=9)


gb = ny * nx * k * 36 / 1e9
for name, fn in (('planar entry point (planar data)', lambda: kernels.change_detection_c3(planar, alpha=a.alpha, n=9)),
                 ('pixel-major entry point, nine real arrays', lambda: kernels.change_detection_c3_pixel_major(yxt, alpha=a.alpha, n=9)),
                 ('pixel-major entry point, interleaved complex off-diagonals', lambda: kernels.change_detection_c3_pixel_major(joint, alpha=a.alpha, n=9)),
                 ('transpose + planar (the former route for this layout)', transposed)):
    ms, km, out = timed(fn)
    print(json.dumps({'route': name, 'k': k, 'ny': ny, 'nx': nx, 'alpha': a.alpha, 'ms': round(ms, 3), 'kernels_ms_per_call': km,
                      'input_GB': round(gb, 2), 'equal_to_planar_map': bool(out is not None and torch.equal(out, ref))}), flush=True)
