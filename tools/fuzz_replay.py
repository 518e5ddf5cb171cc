"""Replay one tools/fuzz_parity.py case and print where the HIP path and the oracle differ."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import tools.fuzz_parity as F
from nd_amd import kernels
from oracle import oracle as O
name, seed, i = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng([seed, i])
captured = {}
if name == 'nlmeans':
    orig = kernels.pixelwise_nlmeans_3d
    def spy(t, out, *a, **k):
        orig(t, out, *a, **k); captured['got'] = out.cpu().numpy(); captured['in'] = t.cpu().numpy()
    kernels.pixelwise_nlmeans_3d = spy
    oorig = O.pixelwise_nlmeans_3d
    def ospy(a, want, *aa, **k):
        oorig(a, want, *aa, **k); captured['want'] = want
    O.pixelwise_nlmeans_3d = ospy
ok, desc = F.CASES[name](rng)
print(ok, desc)
if 'got' in captured:
    g, w = captured['got'], captured['want']
    fin = captured['in'][np.isfinite(captured['in'])]
    bad = np.argwhere(~np.isclose(g, w, rtol=1e-5, atol=2e-6 * (np.abs(fin).max() if fin.size else 1.0), equal_nan=True))
    print('n bad', len(bad), 'of', g.size)
    for idx in bad[:10]:
        idx = tuple(idx); print(idx, g[idx], w[idx], captured['in'][idx])
