"""patch_mode 1 with a search extent along time (the tutorial's window, signed patch distances): the tiled kernel of
round 6 (nlmeans_patch3_kernel) against the per-pixel kernel (ND_AMD_NLM_NOPATCH3=1), 4 variables, 6 x 1024 x 2048.
    python tools/exp_nlm_patch3.py            # both forms, one child process each"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def child(tag):
    import torch
    from nd_amd import kernels, _lib
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(4)
    for nv, r, f, ne in ((4, (1, 3, 3), (1, 1, 1), -1), (4, (1, 3, 3), (1, 1, 1), 50.0), (1, (1, 3, 3), (1, 1, 1), -1),
                         (4, (1, 3, 3), (0, 1, 1), -1), (4, (2, 3, 3), (1, 1, 1), -1)):
        k, ny, nx = 6, 1024, 2048
        x = torch.rand((nv, k, ny, nx), generator=g, device=dev) + 0.5
        y = torch.empty_like(x)
        run = lambda: kernels.pixelwise_nlmeans_3d(x.permute(1, 2, 3, 0), y.permute(1, 2, 3, 0), r, f, 0.5, 0.5, ne,   # noqa: E731
                                                   patch_mode=1, neff_policy=0)
        run(); torch.cuda.synchronize()
        _lib.timing_enable(16)
        t0 = time.perf_counter()
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        by = {}
        for n_, m_ in _lib.timing_collect():
            by.setdefault(n_, []).append(m_)
        _lib.timing_enable(0)
        print(json.dumps({'form': tag, 'nv': nv, 'r': r, 'f': f, 'n_eff': ne, 'shape': [k, ny, nx], 'ms': round(ms, 3),
                          'kernels_ms': {n_: round(sum(v) / len(v), 3) for n_, v in by.items()},
                          'checksum': float(y.double().sum().item())}), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'child':
        child(sys.argv[2])
    else:
        for tag, env in (('tiled (nlmeans_patch3_kernel)', {}), ('per-pixel kernel', {'ND_AMD_NLM_NOPATCH3': '1'})):
            e = dict(os.environ); e.update(env)
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), 'child', tag], env=e)
            if rc:
                sys.exit(rc)
