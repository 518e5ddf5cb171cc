import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nd_amd import kernels
from tests import synth
from tests.test_omnibus_ml_gpu import SHAPES, _two_step
dev = torch.device('cuda:0')
ml = int(sys.argv[1]) if len(sys.argv) > 1 else 3
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
for i, (k, ny, nx) in enumerate(SHAPES):
    if ny <= ml - 1 or nx <= ml - 1: continue
    planes = synth.omnibus_stack(100 + i, k, ny, nx, looks=1, dtype=np.float32, change_frac=0.05)
    d = [torch.from_numpy(p).to(dev) for p in planes]
    got = kernels.change_detection_multilooked(*d, alpha=alpha, ml=ml)
    if got is None:
        print((k, ny, nx), 'not covered'); continue
    two = _two_step(kernels, torch, d, ml, alpha)
    bad = (got != two).any(dim=2)
    n = int(bad.sum())
    print((k, ny, nx), 'bad pixels', n)
    if n:
        ys, xs = torch.nonzero(bad, as_tuple=True)
        print('  rows', sorted(set(ys.tolist()))[:20], ' cols', sorted(set(xs.tolist()))[:40])
