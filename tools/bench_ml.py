"""OmnibusTest(ml=w): the fused multilooking kernel against the two-step path (boxcar over the
stack, then the test) on 24 x 4096^2 float32 -- times per call, kernel times from the library's
events, whole-raster comparison of the two maps.
    python tools/bench_ml.py [--ny 4096 --nx 4096 --k 24 --reps 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nd_amd import _lib, kernels, synth          # noqa: E402


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def kernel_times(fn):
    _lib.timing_enable(64)
    fn()
    torch.cuda.synchronize()
    out = _lib.timing_collect()
    _lib.timing_enable(0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ny', type=int, default=4096)
    ap.add_argument('--nx', type=int, default=4096)
    ap.add_argument('--k', type=int, default=24)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--alphas', default='0.99,0.01,1e-4')
    ap.add_argument('--mls', default='3,5')
    ap.add_argument('--no-two-step', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    stack = synth.wishart_c2_stack(a.k, a.ny, a.nx, looks=1, seed=1234, device=dev, change_frac=0.01)
    for ml in [int(m) for m in a.mls.split(',')]:
        kern = (np.ones((ml, ml)) / ml ** 2).reshape(1, 1, ml, ml)
        for alpha in [float(x) for x in a.alphas.split(',')]:
            fused = lambda: kernels.change_detection_multilooked(stack[0], stack[1], stack[2], stack[3],
                                                                 alpha=alpha, ml=ml)
            got = fused()
            assert got is not None
            line = 'ml=%d alpha=%g fused %.3f ms' % (ml, alpha, timed(fused, a.reps))
            kt = kernel_times(fused)
            line += '  kernels ' + ' '.join('%s=%.3f' % (n, ms) for n, ms in kt)
            if not a.no_two_step:
                def two():
                    m = kernels.convolve(stack, kern)
                    return kernels.change_detection(m[0], m[1], m[2], m[3], alpha=alpha, n=ml * ml)
                want = two()
                line += ' | two-step %.3f ms | maps equal: %s, changes/px %.3f' % (
                    timed(two, max(2, a.reps // 3)), bool(torch.equal(got, want)),
                    float(want.sum()) / (a.ny * a.nx))
            print(line, flush=True)


if __name__ == '__main__':
    main()
