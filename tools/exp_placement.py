"""Does pass A's duration depend on WHERE the stack lies?  One process, the same values copied into freshly
allocated stacks at different addresses (earlier ones kept alive), pass A timed on each.
    python tools/exp_placement.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
dev = torch.device('cuda:0')
k, ny, nx = 24, 4096, 4096
base = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1234, device=dev, change_frac=0.01)
keep = []


def time_passA(st, reps=12):
    for _ in range(3):
        kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9)
    torch.cuda.synchronize()
    _lib.timing_enable(256)
    for _ in range(reps):
        kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9)
    torch.cuda.synchronize()
    ts = [ms for n_, ms in _lib.timing_collect() if n_ == 'omnibus_c2_global']
    _lib.timing_enable(0)
    ts.sort()
    return ts[0], ts[len(ts) // 2], ts[-1]


print(json.dumps({'trial': 'base', 'ptr': hex(base.data_ptr()), 'mod_2MB': base.data_ptr() % (1 << 21), 'mod_1GB': base.data_ptr() % (1 << 30),
                  'passA_ms_min_med_max': [round(x, 4) for x in time_passA(base)]}), flush=True)
for trial in range(10):
    # a filler of varying size moves the next allocation
    keep.append(torch.empty((trial * 37 + 5) << 20, dtype=torch.uint8, device=dev))
    st = synth.empty_stack(4, k, ny, nx, dev)
    st.copy_(base)
    keep.append(st)
    print(json.dumps({'trial': trial, 'ptr': hex(st.data_ptr()), 'mod_2MB': st.data_ptr() % (1 << 21), 'mod_1GB': st.data_ptr() % (1 << 30),
                      'passA_ms_min_med_max': [round(x, 4) for x in time_passA(st)]}), flush=True)
# the first one again (did the device change state meanwhile?)
print(json.dumps({'trial': 'base again', 'passA_ms_min_med_max': [round(x, 4) for x in time_passA(base)]}), flush=True)
