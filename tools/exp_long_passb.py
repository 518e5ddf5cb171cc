"""Long series in the sparse regime: pass B forms (ND_AMD_SEARCH_MODE unset / 0 / 1) at k = 64, 96, 128 on
8.4 Mpx float32, alpha = 0.99: ms per call and per kernel, maps compared between the forms."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import time, hashlib, torch
    from nd_amd import synth, kernels, _lib
    dev = torch.device('cuda:0')
    for k in [int(v) for v in os.environ.get("ND_EXP_KS", "64,96,128").split(",")]:
        st = synth.wishart_c2_stack(k, 2048, 4096, looks=9, seed=3, device=dev, change_frac=0.01)
        fn = lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=0.99, n=9)
        for _ in range(2):
            out = fn()
        torch.cuda.synchronize()
        _lib.timing_enable(256)
        t0 = time.perf_counter()
        for _ in range(5):
            out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        by = {}
        for n_, ms in _lib.timing_collect():
            by.setdefault(n_, []).append(ms)
        _lib.timing_enable(0)
        h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
        print('k=%d: %.2f ms per call, kernels %s, map %s' % (k, dt, {n_: round(sum(v) / 5, 3) for n_, v in by.items()}, h), flush=True)
        del st, out
else:
    for mode in (None, '0', '1'):
        env = dict(os.environ)
        if mode is not None:
            env['ND_AMD_SEARCH_MODE'] = mode
        print('ND_AMD_SEARCH_MODE =', mode, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, check=True)
