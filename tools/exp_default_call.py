import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nd_amd import kernels, synth, xr_lite
from nd_amd.change import OmnibusTest
from oracle import oracle as O
dev = torch.device('cuda:0')
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for k in (24, 12, 48):
    st = synth.wishart_c2_stack(k, 2048, 4096, looks=9, seed=12, device=dev, change_frac=0.01)
    v = [st[i].permute(1, 2, 0).contiguous() for i in range(4)]
    ds = xr_lite.Dataset()
    lay = ('y', 'x', 'time')
    ds['C11'] = (lay, v[0]); ds['C12'] = (lay, torch.complex(v[1], v[2])); ds['C22'] = (lay, v[3])
    for name, algo in (('OmnibusTest()  [n=1, alpha=0.01: the defaults]', OmnibusTest()), ('OmnibusTest(n=9)', OmnibusTest(n=9)), ('OmnibusTest(n=1, alpha=0.99)', OmnibusTest(n=1, alpha=0.99))):
        ms = t_ms(lambda: algo.apply(ds))
        out = algo.apply(ds)
        print('api k=%d yxt %s: %.2f ms  changes %d' % (k, name, ms, int(out.values.sum().item() if hasattr(out.values, 'sum') else 0)), flush=True)
    # oracle check on a crop, default parameters
    crop = [x[:16, :256].cpu().numpy() for x in v]
    want = O.change_detection_planes(crop, 0.01, 1, njobs=8)
    dsc = xr_lite.Dataset()
    dsc['C11'] = (lay, v[0][:16, :256].contiguous()); dsc['C12'] = (lay, torch.complex(v[1], v[2])[:16, :256].contiguous()); dsc['C22'] = (lay, v[3][:16, :256].contiguous())
    got = OmnibusTest().apply(dsc)
    g = got.values.cpu().numpy() if hasattr(got.values, 'cpu') else np.asarray(got.values)
    print('   crop equals oracle:', bool(np.array_equal(g.astype(bool), want.astype(bool))), flush=True)
    del st, v, ds; torch.cuda.empty_cache()
