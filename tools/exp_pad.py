"""Tuning experiment (GPU box): pass A of the headline against the padding between date planes
(elements of float32; the package's stacks use 64) -- and against the variable planes' spacing."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import _lib, kernels, synth
dev = torch.device('cuda:0')
base = synth.wishart_c2_stack(24, 4096, 4096, seed=1234, device=dev, change_frac=0.01)
for pad in [int(a) for a in sys.argv[1:]] or [0, 16, 64, 128, 256, 512, 1024, 4096, 4160, 16448]:
    st = synth.empty_stack(4, 24, 4096, 4096, dev, torch.float32, date_pad=pad) if pad else \
        torch.empty((4, 24, 4096, 4096), device=dev)
    st.copy_(base)
    fn = lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    _lib.timing_enable(400); _lib.timing_select(['omnibus_c2_global'])
    for _ in range(30): fn()
    torch.cuda.synchronize()
    a = sorted(ms for n, ms in _lib.timing_collect())
    _lib.timing_enable(0)
    print(json.dumps({'date_pad_elements': pad, 'plane_stride': st.stride(1), 'passA_ms_min': round(a[0], 4),
                      'median': round(a[len(a) // 2], 4), 'max': round(a[-1], 4)}), flush=True)
    del st
