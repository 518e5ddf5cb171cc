"""Secondary measurement: OmnibusTest C2 pass A / pass B times for other series lengths and dtypes
(device-resident planar stacks, same synthetic recipe as bench.py)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
ap = argparse.ArgumentParser()
ap.add_argument('--cases', default='24:f32,24:f64,12:f64,48:f32,32:f32,16:f32')
ap.add_argument('--ny', type=int, default=4096); ap.add_argument('--nx', type=int, default=4096)
ap.add_argument('--steps', type=int, default=5)
a = ap.parse_args()
dev = torch.device('cuda:0')
for case in a.cases.split(','):
    k, dt = case.split(':'); k = int(k)
    dtype = torch.float32 if dt == 'f32' else torch.float64
    st = synth.wishart_c2_stack(k, a.ny, a.nx, looks=9, seed=1234, device=dev, change_frac=0.01)
    if dtype != st.dtype:
        st64 = synth.empty_stack(4, k, a.ny, a.nx, dev, dtype); st64.copy_(st); st = st64
    run = lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9)
    run(); torch.cuda.synchronize()
    _lib.timing_enable(64); t0 = time.perf_counter()
    for _ in range(a.steps): out = run()
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / a.steps
    by = {}
    for n_, ms in _lib.timing_collect(): by.setdefault(n_, []).append(ms)
    _lib.timing_enable(0)
    km = {n_: sum(v) / len(v) for n_, v in by.items()}
    gb = a.ny * a.nx * k * 4 * st.element_size() / 1e9
    print(json.dumps({'k': k, 'dtype': dt, 'ms': dtm * 1e3, 'Mpx_per_s': a.ny * a.nx / dtm / 1e6, 'kernels_ms': km,
                      'passA_TBps': gb / km['omnibus_c2_global'], 'passA_frac': gb / km['omnibus_c2_global'] / 8.0}))
    del st, out
    torch.cuda.empty_cache()
