"""Dual-pol series beyond the register-retaining lengths in the sparse regime (96 dates x 2048 x 4096 by
default), one build, the environment deciding the form (read once per process: one child per form):

    python tools/exp_c2_split.py               # driver
    python tools/exp_c2_split.py child TAG     # one measurement under the current environment

Per form: the call's time, per-kernel times, the candidate count and a checksum of the change map."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tag):
    import time
    import torch
    from nd_amd import _lib, kernels, synth
    k = int(os.environ.get('EXP_K', '96'))
    ny = int(os.environ.get('EXP_NY', '2048'))
    nx = int(os.environ.get('EXP_NX', '4096'))
    alpha = float(os.environ.get('EXP_ALPHA', '0.99'))
    dev = torch.device('cuda:0')
    tdt = torch.float64 if os.environ.get('EXP_DTYPE', 'f32') == 'f64' else torch.float32
    st = synth.wishart_c2_stack(k, ny, nx, looks=9, seed=1234, device=dev, change_frac=0.01, dtype=tdt)
    fn = lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)   # noqa: E731
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    _lib.timing_enable(256)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    by = {}
    for n_, ms in _lib.timing_collect():
        by.setdefault(n_, []).append(ms)
    avg = {n_: round(sum(v) / len(v), 4) for n_, v in by.items()}
    digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({'tag': tag, 'dtype': str(tdt)[6:], 'k': k, 'ny': ny, 'nx': nx, 'alpha': alpha, 'ms': round(dt * 1e3, 4), 'kernels_ms': avg,
                      'changes': int(out.sum().item()), 'changed_px_frac': round(float((out.sum(dim=2) > 0).float().mean().item()), 6),
                      'map_sha1': digest}), flush=True)


def main():
    forms = [('plain_pass_A_then_gather', {'ND_AMD_C2_SPLIT': '0'}),
             ('time_split_pass_A_lockstep_rounds_on_the_blocked_dump', {'ND_AMD_C2_SPLIT': '1'})]
    for tag, env in forms:
        e = dict(os.environ)
        e.update(env)
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), 'child', tag], env=e)
        if rc != 0:
            sys.exit(rc)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'child':
        child(sys.argv[2])
    else:
        main()
