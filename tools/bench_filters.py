"""Secondary measurements (not the headline bench): boxcar / convolution and non-local means on
device-resident planar stacks.  Prints one JSON line per workload."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from nd_amd import _lib, kernels

def timed(fn, steps, warmup):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    _lib.timing_enable(4 * steps + 8)
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kt = _lib.timing_collect(); _lib.timing_enable(0)
    by = {}
    for n, ms in kt: by.setdefault(n, []).append(ms)
    return dt, {n: sum(v) / len(v) for n, v in by.items()}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--what', default='boxcar')
    ap.add_argument('--k', type=int, default=24); ap.add_argument('--ny', type=int, default=4096); ap.add_argument('--nx', type=int, default=4096)
    ap.add_argument('--w', type=int, default=5); ap.add_argument('--r', type=int, default=10); ap.add_argument('--f', type=int, default=3)
    ap.add_argument('--patch-mode', type=int, default=1); ap.add_argument('--sigma', type=float, default=1.0)
    ap.add_argument('--steps', type=int, default=5); ap.add_argument('--warmup', type=int, default=2)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev); g.manual_seed(7)
    if a.what in ('boxcar', 'conv'):
        x = torch.rand((a.k, a.ny, a.nx), generator=g, device=dev) + 0.5
        out = torch.empty_like(x)
        if a.what == 'boxcar':
            kern = np.ones((1, a.w, a.w)) / (a.w * a.w)
        else:
            kern = np.random.default_rng(0).normal(size=(1, a.w, a.w))
        dt, km = timed(lambda: kernels.convolve(x, kern, out=out), a.steps, a.warmup)
        n = x.numel()
        print(json.dumps({'workload': '%s %dx%d on %dt x %d x %d f32' % (a.what, a.w, a.w, a.k, a.ny, a.nx),
                          'ms': dt * 1e3, 'Mpx_t_per_s': n / dt / 1e6, 'GBps_algorithmic': 8 * n / dt / 1e9,
                          'frac_hbm_peak': 8 * n / dt / 8e12, 'kernels_ms': km,
                          'tiled': os.environ.get('ND_AMD_NO_TILED') is None}))
    elif a.what == 'gaussian':
        x = torch.rand((a.k, a.ny, a.nx), generator=g, device=dev) + 0.5
        out = torch.empty_like(x)
        sig = (0.0, float(a.sigma), float(a.sigma))
        dt, km = timed(lambda: kernels.gaussian_filter(x, sig, out=out), a.steps, a.warmup)
        n = x.numel()
        print(json.dumps({'workload': 'gaussian sigma=%g (y, x) on %dt x %d x %d f32' % (a.sigma, a.k, a.ny, a.nx),
                          'ms': dt * 1e3, 'Mpx_t_per_s': n / dt / 1e6, 'GBps_algorithmic': 16 * n / dt / 1e9,
                          'kernels_ms': km}))
    elif a.what == 'nlmeans':
        x = torch.empty((1, a.k, a.ny, a.nx), device=dev)
        x.copy_(torch.distributions.Gamma(4.0, 4.0).sample((1, a.k, a.ny, a.nx)).to(dev))
        arr = x.permute(2, 3, 1, 0)             # (y, x, time, var) view of planar memory
        out = torch.empty_like(x)
        outv = out.permute(2, 3, 1, 0)
        dt, km = timed(lambda: kernels.pixelwise_nlmeans_3d(arr, outv, (a.r, a.r, 0), (a.f, a.f, 0), 0.5, 0.5, -1,
                                                             patch_mode=a.patch_mode), a.steps, a.warmup)
        n = x.numel()
        nq = (2 * a.r + 1) ** 2 - 1; P = (2 * a.f + 1) ** 2
        flop = n * nq * (P * 3 + 8) if a.patch_mode else n * nq * 2
        print(json.dumps({'workload': 'nlmeans r=%d f=%d patch_mode=%d on %dt x %d x %d f32' % (a.r, a.f, a.patch_mode, a.k, a.ny, a.nx),
                          'ms': dt * 1e3, 'Mpx_t_per_s': n / dt / 1e6, 'TFLOPs_algorithmic': flop / dt / 1e12,
                          'GBps_algorithmic': 8 * n / dt / 1e9, 'kernels_ms': km}))

if __name__ == '__main__':
    main()
