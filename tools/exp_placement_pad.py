"""Pass A of the headline against (padding between date planes) x (where the stack lies): the same values copied
into freshly allocated stacks, several placements per padding, one process.
    python tools/exp_placement_pad.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
dev = torch.device('cuda:0')
k, ny, nx = 24, 4096, 4096
base = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1234, device=dev, change_frac=0.01)


def time_passA(st, reps=8):
    for _ in range(2):
        kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9)
    torch.cuda.synchronize()
    _lib.timing_enable(256)
    for _ in range(reps):
        kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9)
    torch.cuda.synchronize()
    ts = sorted(ms for n_, ms in _lib.timing_collect() if n_ == 'omnibus_c2_global')
    _lib.timing_enable(0)
    return ts[len(ts) // 2]


pads = [int(p) for p in (sys.argv[1].split(',') if len(sys.argv) > 1 else
                         '64,192,1088,2112,4160,8256,16448,65600,262208,1048640'.split(','))]
nplace = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for pad in pads:
    res = []
    keep = []
    for trial in range(nplace):
        keep.append(torch.empty((trial * 53 + 7) << 20, dtype=torch.uint8, device=dev))
        st = synth.empty_stack(4, k, ny, nx, dev, date_pad=pad)
        st.copy_(base)
        res.append(round(time_passA(st), 4))
        keep.append(st)
        if len(keep) > 6:            # keep memory bounded: drop the oldest pair
            del keep[:2]
    del keep
    torch.cuda.empty_cache()
    print(json.dumps({'date_pad_elements': pad, 'passA_ms_by_placement': res, 'mean': round(sum(res) / len(res), 4),
                      'min': min(res), 'max': max(res)}), flush=True)
