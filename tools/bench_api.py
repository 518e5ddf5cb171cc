"""Secondary measurement: the drop-in surface end to end -- OmnibusTest(...).apply(ds) on a
device-resident dataset in the reference's own layout, variables (y, x, time) with time fastest and a
complex64 C12 -- i.e. including the re-layout into the planar stack the kernels read."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import synth, xr_lite
from nd_amd.change import OmnibusTest
ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=24); ap.add_argument('--ny', type=int, default=4096); ap.add_argument('--nx', type=int, default=4096)
ap.add_argument('--ml', type=int, default=0); ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--what', default='omnibus'); ap.add_argument('--host', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda:0')
st = synth.wishart_c2_stack(a.k, a.ny, a.nx, looks=9, seed=1234, device=dev, change_frac=0.01)
ds = xr_lite.Dataset()
yxt = [st[v].permute(1, 2, 0).contiguous() for v in range(4)]            # reference layout
ds['C11'] = (('y', 'x', 'time'), yxt[0])
ds['C12'] = (('y', 'x', 'time'), torch.complex(yxt[1], yxt[2]))
ds['C22'] = (('y', 'x', 'time'), yxt[3])
del st, yxt
if a.host:
    hds = xr_lite.Dataset()
    for name in ds.data_vars:
        hds[name] = (('y', 'x', 'time'), ds[name].values.cpu().numpy())
    ds = hds
if a.what == 'omnibus':
    algo = OmnibusTest(ml=a.ml or None, n=9, alpha=0.99)
elif a.what == 'boxcar':
    from nd_amd.filters import BoxcarFilter
    algo = BoxcarFilter(w=a.ml or 3)
elif a.what == 'gaussian':
    from nd_amd.filters import GaussianFilter
    algo = GaussianFilter(sigma=1.0)
else:
    from nd_amd.filters import NLMeansFilter
    algo = NLMeansFilter(dims=('time', 'y', 'x'), r=(1, 3, 3), f=1, sigma=0.5, h=0.5, n_eff=50)
out = algo.apply(ds); out = algo.apply(ds); out = algo.apply(ds); torch.cuda.synchronize()      # allocator warm
t0 = time.perf_counter()
for _ in range(a.steps): out = algo.apply(ds)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({'workload': '%s(%s).apply(ds), ds on %s in (y, x, time) layout, C12 complex64, %dt x %d x %d' % (type(algo).__name__, a.ml or '', 'host (numpy)' if a.host else 'device', a.k, a.ny, a.nx),
                  'ms': dt * 1e3, 'Mpx_per_s': a.ny * a.nx / dt / 1e6}))
