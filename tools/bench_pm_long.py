"""OmnibusTest on data in the reference's (y, x, time) layout BEYOND 24 dates: the pixel-major entry point
(LDS images folded in place, round 5) against the former route (transpose kernels, then the planar path).
    python tools/bench_pm_long.py [--k 48 --ny 2048 --nx 4096 --alpha 0.99]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=48); ap.add_argument('--ny', type=int, default=2048)
ap.add_argument('--nx', type=int, default=4096); ap.add_argument('--alpha', type=float, default=0.99)
a = ap.parse_args()
dev = torch.device('cuda:0')
k, ny, nx = a.k, a.ny, a.nx
st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1234, device=dev, change_frac=0.01)
yxt = [st[v].permute(1, 2, 0).contiguous() for v in range(4)]
c12 = torch.complex(yxt[1], yxt[2])
ref = kernels.change_detection(st[0], st[1], st[2], st[3], alpha=a.alpha, n=9)
del st


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


def new():
    return kernels.change_detection_pixel_major(yxt[0], c12.real, c12.imag, yxt[3], alpha=a.alpha, n=9)


def old():
    planar = synth.empty_stack(4, k, ny, nx, dev, torch.float32)
    assert kernels.relayout_planar(yxt[0], planar[0]) and kernels.relayout_planar(yxt[3], planar[3])
    assert kernels.relayout_planar_complex(c12.real, c12.imag, planar[1], planar[2])
    return kernels.change_detection(planar[0], planar[1], planar[2], planar[3], alpha=a.alpha, n=9)


gb = ny * nx * k * 16 / 1e9
for name, fn in (('pixel-major entry point', new), ('transpose + planar', old)):
    ms, km, out = timed(fn)
    print(json.dumps({'route': name, 'k': k, 'ny': ny, 'nx': nx, 'alpha': a.alpha, 'ms': round(ms, 3), 'kernels_ms_per_call': km,
                      'input_GB': round(gb, 2), 'input_TBps_whole_call': round(gb / ms, 3),
                      'equal_to_planar_map': bool(out is not None and torch.equal(out, ref))}))
