"""Tuning experiment (GPU box): whole-step time of the headline OmnibusTest call (24 x 4096^2 f32,
alpha = 0.99) under environment switches of the library, one child process per setting (the
switches are read once per process).  Per-step times from one torch event per step boundary on
the launch stream; kernel times from the library's own events in a second loop.

    python tools/exp_step.py [--alpha 0.99] [--steps 40] NAME=VAL,NAME=VAL ...   (each arg = one setting)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(alpha, steps):
    import torch
    from nd_amd import _lib, kernels, synth
    dev = torch.device('cuda:0')
    if os.environ.get('EXP_WORKLOAD') == 'c3':       # one GPU's share of config 4, as bench.py's extra
        st = synth.wishart_c3_stack(48, 1024, 8192, looks=9, seed=4321, device=dev, change_frac=0.01)
        fn = lambda: kernels.change_detection_c3([st[c] for c in range(9)], alpha=alpha, n=9)   # noqa: E731
    else:
        st = synth.wishart_c2_stack(24, 4096, 4096, seed=1234, device=dev, change_frac=0.01)
        fn = lambda: kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)   # noqa: E731
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        out = fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    _lib.timing_enable(16 * steps)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    by = {}
    for name, ms in _lib.timing_collect():
        by.setdefault(name, []).append(ms)
    _lib.timing_enable(0)
    res = {'step_ms': {'min': round(per[0], 4), 'median': round(per[len(per) // 2], 4), 'max': round(per[-1], 4)},
           'step_ms_with_kernel_events': round(wall, 4),
           'kernels_ms': {n: round(sum(v) / len(v), 4) for n, v in by.items()},
           'changes': int(out.sum().item())}
    print(json.dumps(res))


if __name__ == '__main__':
    args = sys.argv[1:]
    alpha, steps = 0.99, 40
    if args and args[0] == 'child':
        child(float(args[1]), int(args[2]))
        sys.exit(0)
    while args and args[0].startswith('--'):
        if args[0] == '--alpha':
            alpha = float(args[1])
        elif args[0] == '--steps':
            steps = int(args[1])
        args = args[2:]
    for setting in args or ['']:
        env = dict(os.environ)
        for kv in filter(None, setting.split(',')):
            k, v = kv.split('=')
            env[k] = v
        r = subprocess.run([sys.executable, __file__, 'child', str(alpha), str(steps)], env=env,
                           capture_output=True, text=True)
        print('%-40s %s' % (setting or '(default)', r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-1500:]))
        sys.stdout.flush()
