"""Full-pol C3 at the benchmark's threshold on one GPU's share of config 4 (48 x 1024 x 8192), one build,
the environment deciding the form of pass A (the switches are read once per process, so the driver
starts one child per form):

    python tools/exp_c3_retain.py              # driver: old form, time-split form (1 and 2 pixel groups)
    python tools/exp_c3_retain.py child TAG    # one measurement under the current environment

Per form: the call's time, per-kernel times, the candidate fraction (entries of the lists of pass A)
and a checksum of the change map (all forms must agree)."""
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
    k = int(os.environ.get('EXP_K', '48'))
    ny = int(os.environ.get('EXP_NY', '1024'))
    nx = int(os.environ.get('EXP_NX', '8192'))
    alpha = float(os.environ.get('EXP_ALPHA', '0.99'))
    dev = torch.device('cuda:0')
    st = synth.wishart_c3_stack(k, ny, nx, looks=9, seed=4321, device=dev, change_frac=0.01)
    planes = [st[c] for c in range(9)]
    for _ in range(2):
        out = kernels.change_detection_c3(planes, alpha=alpha, n=9)
    torch.cuda.synchronize()
    _lib.timing_enable(256)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        out = kernels.change_detection_c3(planes, alpha=alpha, n=9)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    by = {}
    for n_, ms in _lib.timing_collect():
        by.setdefault(n_, []).append(ms)
    avg = {n_: round(sum(v) / len(v), 4) for n_, v in by.items()}
    # candidates: run once more with a workspace of our own and read the counters of pass A
    L = _lib.lib()
    nbytes = L.nd_amd_omnibus_c3_workspace_bytes(ny, nx, k)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    import ctypes as C
    ch = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
    ptrs = (C.c_void_p * 9)(*[t.data_ptr() for t in planes])
    _lib.check(L.nd_amd_omnibus_c3(ptrs, 0, ny, nx, k, planes[0].stride(1), planes[0].stride(2), planes[0].stride(0),
                                   9, alpha, ch.data_ptr(), None, None, ws.data_ptr(), nbytes,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    counts = ws[:128 * 32 * 4].view(torch.int32).view(128, 32)[:, 0]
    cand = int(counts.sum().item())
    digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
    assert torch.equal(out, ch)
    print(json.dumps({'tag': tag, 'k': k, 'ny': ny, 'nx': nx, 'alpha': alpha, 'ms': round(dt * 1e3, 4), 'kernels_ms': avg,
                      'candidates': cand, 'candidate_frac': round(cand / (ny * nx), 6),
                      'max_list': int(counts.max().item()), 'workspace_MB': round(nbytes / 2**20, 1),
                      'changes': int(out.sum().item()), 'map_sha1': digest}), flush=True)


def main():
    forms = [('planar_pass_A_then_gather', {'ND_AMD_C3_RETAIN': '0'}),
             ('time_split_retain_pg1', {'ND_AMD_C3_RETAIN': '1', 'ND_AMD_C3_RETAIN_PG': '1'}),
             ('time_split_retain_pg1_128regs', {'ND_AMD_C3_RETAIN': '1', 'ND_AMD_C3_RETAIN_OCC': '4'}),
             ('time_split_retain_pg2', {'ND_AMD_C3_RETAIN': '1', 'ND_AMD_C3_RETAIN_PG': '2'})]
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
