"""z / P rasters on top of the change map: ms per call with and without them (24 x 2048 x 4096 and
24 x 4096 x 4096 float32, thresholds below the sparse regime), rasters compared with the separate-pass form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd_amd import synth, kernels
dev = torch.device('cuda:0')
k = 24
for ny, nx in ((2048, 4096), (4096, 4096)):
    st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=77, device=dev, change_frac=0.01)
    for alpha in (1e-4, 0.01, 0.2, 0.99):
        ms = {}
        for stats in (False, True):
            fn = lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9, stats=stats)
            for _ in range(5):
                out = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                out = fn()
            torch.cuda.synchronize()
            ms[stats] = (time.perf_counter() - t0) / 20 * 1e3
        print('%d x %d alpha %g: map only %.3f ms, with z / P %.3f ms (+%.0f %%)'
              % (ny, nx, alpha, ms[False], ms[True], 100 * (ms[True] / ms[False] - 1)), flush=True)
    del st
