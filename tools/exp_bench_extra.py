"""Run single `extra` workloads of bench.py with their timing and checks (not the traffic mode):
    python tools/exp_bench_extra.py gauss1 boxcar5,gauss1"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
class A: pass
a = A(); a.k, a.ny, a.nx, a.looks, a.alpha, a.change_frac, a.scaling, a.patch_mode = 24, 4096, 4096, 9, 0.99, 0.01, 'weak', 0
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
w = bench.OmnibusC2(a, 0, 1, dev)
def barrier(): torch.cuda.synchronize()
for group in sys.argv[1:]:
    keys = group.split(',')
    out = []
    for k in keys:
        out += bench.extras(w, barrier, dev, only=k)
    for e in out:
        print(group, e['key'], round(e['ms'], 3), {n: round(v, 3) for n, v in e['kernels_ms'].items()})
