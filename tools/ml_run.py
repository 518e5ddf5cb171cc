"""A few launches of the fused multilooking test and of the plain test on the same stack (profiling target).
    python tools/ml_run.py [ml] [alpha] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nd_amd import kernels, synth
ml = int(sys.argv[1]) if len(sys.argv) > 1 else 3
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 0.99
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
st = synth.wishart_c2_stack(24, 4096, 4096, looks=1, seed=1234, device=dev, change_frac=0.01)
for _ in range(reps):
    kernels.change_detection_multilooked(st[0], st[1], st[2], st[3], alpha=alpha, ml=ml)
    kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
torch.cuda.synchronize()
