"""Which part of the full-pol traffic run upsets rocprofv3's counter pass: the synthesis or the kernels?
    python tools/exp_c3_prof.py synth | kernels"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nd_amd import kernels, synth
dev = torch.device('cuda:0')
what = sys.argv[1]
if what == 'synth':
    st = synth.wishart_c3_stack(48, 256, 2048, looks=9, seed=1, device=dev, change_frac=0.01)
    torch.cuda.synchronize(); print('synth ok', tuple(st.shape))
else:
    k, ny, nx = 48, 256, 2048
    st = torch.rand((9, k, ny, nx), device=dev) * 0.1
    st[:3] += 1.0
    out = kernels.change_detection_c3([st[c] for c in range(9)], alpha=0.99, n=9)
    torch.cuda.synchronize(); print('kernels ok', int(out.sum()))
