"""Cliff probe 2: the tutorial's and other plausible parameter sets through the classes, float64 / float32, both layouts."""
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
k, ny, nx = 12, 1024, 2048
algos = [('OmnibusTest(ml=3)', lambda: OmnibusTest(ml=3)), ('OmnibusTest(ml=5, alpha=0.99)', lambda: OmnibusTest(ml=5, alpha=0.99)),
         ('OmnibusTest(ml=7)', lambda: OmnibusTest(ml=7)),
         ('NLMeans tutorial (t,y,x) r=(1,3,3) f=1 n_eff=50', lambda: NLMeansFilter(dims=('time', 'y', 'x'), r=(1, 3, 3), f=1, n_eff=50)),
         ('NLMeans (y,x,t) r=(3,3,1)', lambda: NLMeansFilter(dims=('y', 'x', 'time'), r=(3, 3, 1), f=1)),
         ('NLMeans r=5 f=2 sigma=.5', lambda: NLMeansFilter(r=5, f=2, sigma=0.5, h=0.5)),
         ('NLMeans r=3 f=0', lambda: NLMeansFilter(r=3, f=0, sigma=0.5, h=0.5)),
         ('Boxcar (y,x,time) w=3', lambda: BoxcarFilter(dims=('y', 'x', 'time'), w=3)), ('Boxcar (time,) w=3', lambda: BoxcarFilter(dims=('time',), w=3)),
         ('Boxcar w=7', lambda: BoxcarFilter(w=7)), ('Boxcar w=4', lambda: BoxcarFilter(w=4)),
         ('Gaussian (time,) s=1', lambda: GaussianFilter(dims=('time',), sigma=1)), ('Gaussian (y,x,time) s=(1,1,.5)', lambda: GaussianFilter(dims=('y', 'x', 'time'), sigma=(1, 1, 0.5))),
         ('Gaussian s=2.5', lambda: GaussianFilter(sigma=2.5)), ('Gaussian s=(1,2)', lambda: GaussianFilter(sigma=(1, 2))),
         ('Convolution sobel 3x3', lambda: ConvolutionFilter(kernel=np.array([[1., 0, -1], [2, 0, -2], [1, 0, -1]]))),
         ('Convolution (x,y) order', lambda: ConvolutionFilter(dims=('x', 'y'), kernel=np.ones((3, 5)) / 15))]
for dt in (torch.float64, torch.float32):
    st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=5, device=dev, change_frac=0.01).to(dt)
    for lay in (('y', 'x', 'time'), ('time', 'y', 'x')):
        v = [st[i].permute(1, 2, 0).contiguous() if lay[0] == 'y' else st[i].contiguous() for i in range(4)]
        ds = xr_lite.Dataset()
        ds['C11'] = (lay, v[0]); ds['C12'] = (lay, torch.complex(v[1], v[2])); ds['C22'] = (lay, v[3])
        for name, mk in algos:
            try:
                algo = mk()
                ms = t_ms(lambda: algo.apply(ds))
                print('%s %s %-48s: %8.2f ms' % (str(dt)[6:], ''.join(d[0] for d in lay), name, ms), flush=True)
            except Exception as e:
                print('%s %s %-48s: FAILED %r' % (str(dt)[6:], ''.join(d[0] for d in lay), name, str(e)[:120]), flush=True)
        del ds, v; torch.cuda.empty_cache()
    del st; torch.cuda.empty_cache()
