"""Probe: the headline test on plain contiguous torch tensors (what a user has) of various raster shapes, against the
padded benchmark stack (nd_amd.synth.empty_stack pads the date planes by 256 B)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
def t_ms(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
k = 24
for ny, nx in ((4096, 4096), (4000, 4000), (5000, 5000), (2048, 8192), (8192, 2048), (4096, 4100), (3000, 6000)):
    pad = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1, device=dev, change_frac=0.01)
    plain = pad.contiguous() if not pad.is_contiguous() else pad.clone()
    for name, st in (('padded', pad), ('plain ', plain)):
        for alpha in (0.99, 0.01):
            ms = t_ms(lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9))
            print('%s %dx%d alpha=%g: %.3f ms  %.1f Mpx/s  (%.3f of 8 TB/s on 408 B/px)' % (name, ny, nx, alpha, ms, ny * nx / ms / 1e3, ny * nx * 408 / (ms * 1e-3) / 8e12), flush=True)
    del pad, plain; torch.cuda.empty_cache()
