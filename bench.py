#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the nd_amd hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Mpixels/s of OmnibusTest on a dual-pol C2 stack,
24 dates x 4096 x 4096 float32 per GPU, plus the achieved HBM GB/s of the
dominant kernel against the chip's peak.  One "step" = one full OmnibusTest
pass (both kernels) over the rank's device-resident stack.  With N ranks each
rank owns one y-tile of a (N*4096) x 4096 raster (weak scaling; the omnibus test
is per pixel, there is no data-path collective).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# HBM bytes per launch of the dominant kernel from the separate rocprofv3 PMC passes
# (2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_omnibus_rocprof.txt), keyed by (k, ny, nx, alpha, frac)
MEASURED_TRAFFIC = {(24, 4096, 4096, 0.99, 0.01): 6.9769e9}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--k', type=int, default=24)
    ap.add_argument('--ny', type=int, default=4096)
    ap.add_argument('--nx', type=int, default=4096)
    ap.add_argument('--looks', type=int, default=9)
    ap.add_argument('--alpha', type=float, default=0.99)
    ap.add_argument('--change-frac', type=float, default=0.01)
    ap.add_argument('--cpu-rows', type=int, default=4096,
                    help='rows of the stack the CPU baseline is timed on (0 = skip)')
    ap.add_argument('--traffic-bytes', type=float, default=None,
                    help='HBM bytes per launch of the dominant kernel from a separate '
                         'rocprofv3 --pmc pass (profiles/), copied into roofline.traffic')
    return ap.parse_args()


def _usable_cores():
    """Host cores this process may actually use: CPU affinity, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = max(1, min(n, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(stack, args, npix_rows):
    """Time the CPU oracle (oracle/, a port of the reference's Cython path) on
    the first `npix_rows` rows of the same stack, all host cores."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    rows = min(npix_rows, stack.shape[2])
    host = stack[:, :, :rows, :].cpu().numpy()            # (4, k, rows, nx)
    planes = [np.moveaxis(host[v], 0, -1) for v in range(4)]   # (rows, nx, k) strided views
    cores = _usable_cores()
    # warm (page-in + thread pool) on a sliver, then time
    O.change_detection_planes([p[:8] for p in planes], args.alpha, args.looks, njobs=cores)
    t0 = time.perf_counter()
    ch = O.change_detection_planes(planes, args.alpha, args.looks, njobs=cores)
    dt = time.perf_counter() - t0
    npx = rows * stack.shape[3]
    return {
        'value': npx / dt / 1e6, 'unit': 'Mpixels/s', 'cores': int(cores), 'kind': 'port',
        'sample': 'oracle/nd_oracle.c (C port of nd/_change.pyx, reference-order arithmetic, '
                  'OpenMP over rows) on the first %d rows x %d cols x %d dates of the same '
                  'stack: %.2f s wall' % (rows, stack.shape[3], stack.shape[1], dt),
        'flagged_fraction': float((ch.sum(axis=2) > 0).mean()),
    }, ch


def main():
    args = parse()
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU (nd_amd has no CPU path)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or 'RANK' in os.environ:      # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from nd_amd import _lib, kernels, synth

    k, ny, nx = args.k, args.ny, args.nx
    # rank r owns rows [r*ny, (r+1)*ny) of the (world*ny) x nx raster
    stack = synth.wishart_c2_stack(k, ny, nx, looks=args.looks, seed=1234 + rank, device=dev,
                                   change_frac=args.change_frac)
    torch.cuda.synchronize()

    def step():
        return kernels.change_detection(stack[0], stack[1], stack[2], stack[3],
                                        alpha=args.alpha, n=args.looks,
                                        dims=('time', 'y', 'x'))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    _lib.timing_enable(2 * (args.steps + args.warmup) + 8)
    for _ in range(args.warmup):
        out = step()
    barrier()
    _lib.timing_collect()          # drop the warm-up launches; the events themselves are reused
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    kt = _lib.timing_collect()
    _lib.timing_enable(0)

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    npix = ny * nx
    value = world * npix * args.steps / dt / 1e6
    flagged = float((out.sum(dim=2) > 0).float().mean().item())

    if rank == 0:
        by = {}
        for name, ms in kt:
            by.setdefault(name, []).append(ms)
        avg = {n: sum(v) / len(v) for n, v in by.items()}
        dom = 'omnibus_c2_global'
        alg_bytes = npix * k * 4 * stack.element_size()        # 384 B/px at k=24 f32
        achieved = alg_bytes / (avg[dom] * 1e-3) / 1e9
        res = {
            'metric': 'Mpixels/s OmnibusTest dual-pol %dt x %d x %d' % (k, ny, nx),
            'value': value, 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': 'OmnibusTest dual-pol C2, synthetic %dt x %d x %d float32 per GPU '
                            '(BASELINE.json configs[1]), n=%d looks, alpha=%g, %.3g of pixels '
                            'with a x4 step; inputs resident in HBM'
                            % (k, ny, nx, args.looks, args.alpha, args.change_frac),
                'arithmetic': 'float32 planes and running sums; float64 product of determinants, '
                              'logs and chi-square pair (the reference\'s rounding points)',
                'flagged_pixel_fraction': flagged,
                'sharding': 'y-tiles, one per rank, no collective',
            },
            'kernels_ms': avg,
            'roofline': {
                'kernel': dom, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                'algorithmic_bytes_per_launch': alg_bytes,
                'traffic': args.traffic_bytes if args.traffic_bytes is not None else
                MEASURED_TRAFFIC.get((k, ny, nx, args.alpha, args.change_frac)),
            },
        }
        if world == 1 and args.cpu_rows > 0:
            cb, ch_cpu = cpu_baseline(stack, args, args.cpu_rows)
            rows = ch_cpu.shape[0]
            same = bool((out[:rows].cpu().numpy() == ch_cpu).all())
            cb['gpu_matches_cpu_on_sample'] = same
            res['cpu_baseline'] = cb
        print(json.dumps(res))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
