"""Streaming fused search (ND_AMD_FUSED_FORM=0) against the chain form (=2) at low thresholds, 24 x 4096^2 float32:
where should the default switch from one to the other?  (same process order for both, maps compared by hash)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import time, hashlib, torch
    from nd_amd import synth, kernels, _lib
    dev = torch.device('cuda:0')
    K = int(os.environ.get('ND_EXP_K', '24'))
    st = synth.wishart_c2_stack(K, 4096, 4096 if K <= 24 else 2048, looks=9, seed=1234, device=dev, change_frac=0.01)
    for alpha in [float(v) for v in os.environ.get('ND_EXP_ALPHAS', '1e-4,1e-3,5e-3,0.01,0.02').split(',')]:
        fn = lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
        for _ in range(5):
            out = fn()
        torch.cuda.synchronize()
        _lib.timing_enable(512)
        t0 = time.perf_counter()
        for _ in range(20):
            out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        by = {}
        for n_, ms in _lib.timing_collect():
            by.setdefault(n_, []).append(ms)
        _lib.timing_enable(0)
        h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]
        print('alpha %g: %.3f ms per call, fused kernel %.3f, map %s' % (alpha, dt, sum(by['omnibus_c2_fused']) / 20, h), flush=True)
else:
    for form in ('0', '2'):
        print('ND_AMD_FUSED_FORM =', form, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, ND_AMD_FUSED_FORM=form), check=True)
