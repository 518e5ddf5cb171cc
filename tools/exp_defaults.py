"""Cliff probe: every reference class with its DEFAULT parameters on a float64 (y, x, time) dataset -- what
nd.testing.generate_test_dataset produces, scaled up -- and on float32; ms per .apply()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nd_amd import synth, xr_lite
from nd_amd.change import OmnibusTest
from nd_amd.filters import BoxcarFilter, GaussianFilter, NLMeansFilter, ConvolutionFilter
dev = torch.device('cuda:0')
def t_ms(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
k, ny, nx = 10, 2048, 2048
for dt in (torch.float64, torch.float32):
    st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=5, device=dev, change_frac=0.01).to(dt)
    for lay in (('y', 'x', 'time'), ('time', 'y', 'x')):
        v = [st[i].permute(1, 2, 0).contiguous() if lay[0] == 'y' else st[i].contiguous() for i in range(4)]
        for split in (True, False):
            ds = xr_lite.Dataset()
            if split:
                for name, a in zip(('C11', 'C12__re', 'C12__im', 'C22'), v): ds[name] = (lay, a)
            else:
                ds['C11'] = (lay, v[0]); ds['C12'] = (lay, torch.complex(v[1], v[2])); ds['C22'] = (lay, v[3])
            for name, algo in (('OmnibusTest()', OmnibusTest()), ('OmnibusTest(n=9)', OmnibusTest(n=9)), ('NLMeansFilter()', NLMeansFilter()),
                               ('BoxcarFilter()', BoxcarFilter()), ('GaussianFilter()', GaussianFilter()), ('ConvolutionFilter()', ConvolutionFilter()),
                               ('NLMeansFilter(r=3, f=1)', NLMeansFilter(r=3, f=1))):
                try:
                    ms = t_ms(lambda: algo.apply(ds))
                    print('%s %s %s %-26s: %8.2f ms' % (str(dt)[6:], ''.join(d[0] for d in lay), 'split  ' if split else 'complex', name, ms), flush=True)
                except Exception as e:
                    print('%s %s %s %-26s: FAILED %r' % (str(dt)[6:], ''.join(d[0] for d in lay), 'split  ' if split else 'complex', name, e), flush=True)
            del ds
        del v; torch.cuda.empty_cache()
    del st; torch.cuda.empty_cache()
