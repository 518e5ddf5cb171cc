"""OmnibusTest C2 on the benchmark stack over a range of thresholds (the dense regime: the
reference's default alpha = 0.01, the tutorial's 1e-4), each compared with the CPU oracle on the
whole raster (byte for byte).  Prints one JSON line per alpha."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from nd_amd import _lib, kernels, synth
ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=24); ap.add_argument('--ny', type=int, default=4096); ap.add_argument('--nx', type=int, default=4096)
ap.add_argument('--alphas', default='1e-4,0.01,0.5,0.9,0.99'); ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--cpu-rows', type=int, default=4096); ap.add_argument('--dtype', default='f32')
ap.add_argument('--layouts', default='planar', help="planar and/or pm (the reference's (y, x, time) layout, C12 complex)")
ap.add_argument('--want-dir', default=None,
                help='directory of oracle maps shared by several runs on the same stack (the synthesis is seeded): '
                     'want_<dtype>_<k>_<ny>_<nx>_<rows>_<alpha>.npy is read if present, computed and written otherwise')
a = ap.parse_args()
dev = torch.device('cuda:0')
dt_ = torch.float32 if a.dtype == 'f32' else torch.float64
st = synth.wishart_c2_stack(a.k, a.ny, a.nx, looks=9, seed=1234, device=dev, change_frac=0.01, dtype=dt_)
host = None
pm = None
for layout, alpha in [(l, float(x)) for l in a.layouts.split(',') for x in a.alphas.split(',')]:
    if layout == 'pm':
        if pm is None:
            yxt = [st[v].permute(1, 2, 0).contiguous() for v in range(4)]
            c12 = torch.complex(yxt[1], yxt[2])
            pm = (yxt[0], c12.real, c12.imag, yxt[3])
        fn = lambda: kernels.change_detection_pixel_major(*pm, alpha=alpha, n=9)
    else:
        fn = lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
    for _ in range(2): out = fn()
    torch.cuda.synchronize()
    _lib.timing_enable(8 * a.steps + 8)
    t0 = time.perf_counter()
    for _ in range(a.steps): out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    kt = _lib.timing_collect(); _lib.timing_enable(0)
    by = {}
    for n_, ms in kt: by.setdefault(n_, []).append(ms)
    res = {'layout': layout, 'alpha': alpha, 'ms': dt * 1e3, 'Mpx_per_s': a.ny * a.nx / dt / 1e6,
           'kernels_ms': {n_: round(sum(v) / len(v), 4) for n_, v in by.items()},
           'flagged': float((out.sum(dim=2) > 0).float().mean().item()),
           'changes_per_px': float(out.sum().item()) / (a.ny * a.nx)}
    if a.cpu_rows > 0:
        from oracle import oracle as O
        rows = min(a.cpu_rows, a.ny)
        if host is None:
            host = st[:, :, :rows].cpu().numpy()
        planes = [np.moveaxis(host[v], 0, -1) for v in range(4)]
        t0 = time.perf_counter()
        wf = (os.path.join(a.want_dir, 'want_%s_%d_%d_%d_%d_%r.npy' % (a.dtype, a.k, a.ny, a.nx, rows, alpha))
              if a.want_dir else None)
        if wf and os.path.exists(wf):
            want = np.load(wf)
        else:
            want = O.change_detection_planes(planes, alpha, 9, njobs=len(os.sched_getaffinity(0)))
            if wf:
                tmp = '%s.%d.tmp.npy' % (wf, os.getpid())
                np.save(tmp, want)
                os.replace(tmp, wf)            # (several runs may compute the same map at once: last one wins, all equal)
        res['cpu_s'] = time.perf_counter() - t0
        res['bytes_differing'] = int((out[:rows].cpu().numpy() != want).sum())
        res['compared_px'] = rows * a.nx
    print(json.dumps(res)); sys.stdout.flush()
