"""Fused y-then-x Gaussian (nd_amd_correlate1d_yx): float32 ring vs float64 window, rows per wave.
One child process per setting (the switches are read once per process)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    from nd_amd import kernels
    x = torch.rand((24, 4096, 4096), device='cuda')
    out = torch.empty_like(x)
    res = {}
    for sg in (0.3, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0):
        sig = (0, sg, sg)
        for _ in range(3): kernels.gaussian_filter(x, sig, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): kernels.gaussian_filter(x, sig, out=out)
        e1.record(); torch.cuda.synchronize()
        res[sg] = round(e0.elapsed_time(e1) / 10, 4)
    print(json.dumps(res))
else:
    for dw in ('0', '1'):
        for rpc in ('0', '64'):
            env = dict(os.environ, ND_AMD_YX_DWIN=dw, ND_AMD_YX_RPC=rpc)
            fn = '/tmp/gyx_%s_%s.out' % (dw, rpc)
            with open(fn, 'w') as fo:
                subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, stdout=fo,
                               stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=300)
            print('dwin', dw, 'rpc', rpc, open(fn).read().strip().splitlines()[-1], flush=True)
