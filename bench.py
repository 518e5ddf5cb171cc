#!/usr/bin/env python3
"""
bench.py -- benchmark of the nd_amd hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload omnibus|c3|pipeline]
                    [--scaling weak|strong] [--no-extra]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Headline (default): BASELINE.json's metric, Mpixels/s of OmnibusTest on a dual-pol C2 stack of
24 dates x 4096 x 4096 float32 (configs[1]), inputs resident in HBM, plus the achieved HBM GB/s of
the dominant kernel against the chip's peak (kernel durations from HIP events recorded on the launch
stream inside the timed region).  One "step" = one full pass of the workload over the rank's
device-resident rows.

Workloads (--workload), all row-sharded over the ranks (nd_amd/tiles.py):
  omnibus   OmnibusTest C2, 24t x 4096 x 4096                       no collective
  c3        OmnibusTest full-pol C3 (extension), 48t x 1024 x 8192  no collective (config 4's share)
  pipeline  NLMeansFilter -> OmnibusTest with the tutorial's parameters, 24t x 2048 x 16384 x 4
            variables (config 5's share); with N > 1 every step exchanges the halo rows with the
            neighbouring ranks over RCCL (point-to-point), then filters tile+halo and tests.
Scaling (--scaling): weak (default) = every rank owns a raster of the stated size, the job's raster
is N times as tall; strong = the stated raster is split over the ranks by tiles.row_partition.

At N = 1 with the default workload the line also carries `cpu_baseline` (the C oracle on the host's
cores).  `--extras` (its own invocation) adds the other kernels of the path (dense-threshold omnibus,
multilooking, pixel-major, C3, boxcar, non-local means in both patch modes, Gaussian, the pipeline),
each with ms, Mpx/s, roofline and a sampled oracle check: they are written to bench_extras.json, the
long form of the headline (notes, per-step durations, N > 1 per-rank records) to bench_detail.json.

`--gpus N` with N > 1 and no launcher around it starts the N ranks itself (the launch line above, as a
child process) and relays rank 0's line; a WORLD_SIZE that differs from --gpus is an error.

Prints ONE JSON line on rank 0: the headline object only, strict JSON, under 4 KB.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The CPU baselines run the oracle on every host core (OpenMP).  libgomp's workers spin for a while
# after a parallel region; on a box whose CPU quota equals its core count they then starve the
# thread that launches the next workload's kernels (GaussianFilter right behind the boxcar
# baseline: 3.8 ms per step on the host against 0.73 ms on the device).  Passive waiting puts them
# to sleep at once.  (Read by libgomp when the oracle library is first loaded.)
os.environ.setdefault('OMP_WAIT_POLICY', 'passive')
os.environ.setdefault('GOMP_SPINCOUNT', '0')

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes per launch (fetch_factor x FETCH_SIZE + WRITE_SIZE: the guide's gfx950 correction, calibrated
# per access pattern in tools/summarize_traffic.py) of every
# workload's kernels, from separate rocprofv3 --pmc passes over `bench.py --traffic-run KEY`
# (tools/collect_traffic.sh writes the file, with the commit it was taken at).  Reported with its
# source; a workload or kernel the file does not hold gets traffic = null.
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'r06_traffic.json')
FULL_SYNTH = os.environ.get('ND_AMD_TRAFFIC_FULL_SYNTH', '') == '1'


def csrc_sha():
    """Hash of the kernel sources the traffic figures were measured on (tools/summarize_traffic.py
    stores it in the file): figures of other sources are not reported as measurements."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'nd_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.hpp')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def profiled_traffic(key, hint):
    """-> (bytes per launch, source) of the kernel of workload `key` whose symbol contains `hint`
    (the largest one if several do), or (None, None)."""
    try:
        tab = json.load(open(TRAFFIC_FILE))
    except Exception:
        return None, None
    rows = [r for r in tab.get('workloads', {}).get(key, []) if hint in r['kernel']]
    if not rows:
        return None, None
    if tab.get('csrc_sha') != csrc_sha():
        return None, 'stale: %s@%s is of other kernel sources' % (os.path.relpath(TRAFFIC_FILE, ROOT), tab.get('commit', '?'))
    r = max(rows, key=lambda r: r['traffic_bytes'])
    return r['traffic_bytes'], '%s@%s' % (os.path.relpath(TRAFFIC_FILE, ROOT), tab.get('commit', '?'))

DEFAULTS = {'omnibus': (24, 4096, 4096), 'c3': (48, 1024, 8192), 'pipeline': (24, 2048, 16384)}
# tutorial parameters (examples/tutorial_s1.ipynb cells 11, 15; NLMeansFilter defaults sigma=h=f=1)
TUT = dict(r=(1, 3, 3), f=(1, 1, 1), sigma=1.0, h=1.0, n_eff=50.0, n=50, alpha=1e-4)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', choices=sorted(DEFAULTS), default='omnibus')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--k', type=int, default=None)
    ap.add_argument('--ny', type=int, default=None)
    ap.add_argument('--nx', type=int, default=None)
    ap.add_argument('--looks', type=int, default=9)
    ap.add_argument('--alpha', type=float, default=None,
                    help='omnibus / c3: 0.99 (SURVEY 8d); pipeline: the tutorial\'s 1e-4')
    ap.add_argument('--change-frac', type=float, default=0.01)
    ap.add_argument('--patch-mode', type=int, default=0, help='pipeline: 0 reference, 1 signed')
    ap.add_argument('--cpu-rows', type=int, default=4096,
                    help='rows of the stack the all-core CPU baseline is timed on (0 = skip)')
    ap.add_argument('--extras', action='store_true',
                    help='N = 1, default workload: also run the secondary workloads (dense thresholds, '
                         'multilooking, pixel-major, C3, boxcar, Gaussian, non-local means, the pipeline), '
                         'device clocks and transfer rates; they go to bench_extras.json / '
                         'bench_detail.json, never into the line')
    ap.add_argument('--no-extra', action='store_true', help='(accepted, ignored: extras are opt-in now)')
    ap.add_argument('--no-secondary', action='store_true',
                    help='N = 1, default workload: skip the `secondary` block of the line (BASELINE configs 3 - 5 '
                         'at one GPU\'s share: a few steps and a sampled oracle check each, ~30 s of wall)')
    ap.add_argument('--traffic-bytes', type=float, default=None,
                    help='HBM bytes per launch of the dominant kernel from a separate '
                         'rocprofv3 --pmc pass, copied into roofline.traffic')
    ap.add_argument('--traffic-run', default=None, metavar='KEY',
                    help='run only workload KEY (headline or one of the extras) for a few steps, '
                         'no timing events, no checks: the command rocprofv3 --pmc is pointed at')
    a = ap.parse_args()
    dk, dy, dx = DEFAULTS[a.workload]
    a.k, a.ny, a.nx = a.k or dk, a.ny or dy, a.nx or dx
    if a.alpha is None:
        a.alpha = TUT['alpha'] if a.workload == 'pipeline' else 0.99
    return a


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


# ------------------------------------------------------------------------------------------
# timing helpers
# ------------------------------------------------------------------------------------------
def _kernel_avgs(kt):
    by = {}
    for name, ms in kt:
        by.setdefault(name, []).append(ms)
    return {n: sum(v) / len(v) for n, v in by.items()}


TRAFFIC_MODE = False       # --traffic-run: plain loops, no events, no checks


def timed(fn, steps, warmup, barrier, launches_per_step=32, only=None, per_step=None, settle=True):
    """W untimed calls, then EXACTLY `steps` calls between two barriers.  -> (seconds, kernel
    averages from the library's HIP events on the launch stream, last result).
    only: names of the kernels to time (an event pair costs a few microseconds of stream time; the
    headline times its dominant kernel only).  per_step: a list that receives the duration of
    every step in ms (one torch event per step boundary on the launch stream)."""
    from nd_amd import _lib
    if TRAFFIC_MODE:
        out = None
        for _ in range(max(1, min(steps, 3))):
            out = fn()
        barrier()
        return 1.0, {}, out
    import torch
    # The checks and CPU baselines in front of a timed loop run the oracle on every core of the box's CPU
    # share; a cgroup that has burnt its quota for the current scheduler period is paused as a whole until
    # the next one, and that pause (tens of milliseconds) then falls into the loop that follows -- the
    # Gaussian line, right behind the boxcar's all-core baseline, showed 4 ms "per step" around a 0.7 ms
    # kernel in its first batch whenever the baseline had just run, never without it.  Let the period pass
    # (settle; not in front of the headline's region, which follows the device-side synthesis directly and
    # whose W warm-up steps are the caller's).
    if settle:
        time.sleep(0.25)
    _lib.timing_enable(launches_per_step * (steps + warmup) + 16)
    _lib.timing_select(only)
    out = None
    for _ in range(warmup):
        out = fn()
    barrier()
    _lib.timing_collect()          # drop the warm-up launches; the events themselves are reused
    _lib.timing_dropped()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step is not None else None
    t0 = time.perf_counter()
    if ev:
        ev[0].record()
    for i in range(steps):
        out = fn()
        if ev:
            ev[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    kt = _lib.timing_collect()
    dropped = _lib.timing_dropped()
    _lib.timing_enable(0)
    if dropped:
        raise RuntimeError('%d kernel launches were not timed: timing ring too small' % dropped)
    if ev:
        per_step.extend(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return dt, _kernel_avgs(kt), out


def timed_extra(fn, steps, warmup, barrier):
    """The secondary workloads' timing, as the headline's: a short loop with every kernel's event pair
    finds the dominant kernel (and the others' durations), then the timed loop carries that kernel's pair
    only -- an event pair costs ~3 us of stream time, and six around the small kernels of a 1.5 ms call
    lengthened it by 1 - 2 %.  -> (seconds, kernel averages, last result) like timed()."""
    if TRAFFIC_MODE:
        return timed(fn, steps, warmup, barrier)
    _, km_all, _ = timed(fn, 3, warmup, barrier)
    dom = max(km_all, key=km_all.get) if km_all else None
    dt, km_dom, out = timed(fn, steps, warmup, barrier, only=[dom] if dom else None)
    km = dict(km_all)
    km.update(km_dom)
    return dt, km, out


def device_state(index):
    """Clocks, temperature and power of the device as rocm-smi reports them (None where it does not)."""
    import subprocess
    # under rocprofv3 the profiler's preloaded library initialises the GPU in every child process as well,
    # and rocm-smi (a script: env -> python3) would then replace a GPU-initialised program: not started there
    if 'rocprof' in os.environ.get('LD_PRELOAD', '') or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ):
        return {'skipped': 'running under rocprofv3'}
    try:
        r = subprocess.run(['rocm-smi', '-d', str(index), '--showclocks', '--showtemp', '--showpower', '--json'],
                           capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
    except Exception as e:              # noqa: BLE001
        return {'error': repr(e)[:120]}
    keep = {}
    for key, val in card.items():
        kl = key.lower()
        if any(t in kl for t in ('sclk', 'mclk', 'fclk', 'socclk', 'temperature', 'power')):
            keep[key] = val
    return keep


def transfer_rates(dev, nbytes=1 << 30):
    """host <-> device rates, reported once and never part of `value` (inputs are resident
    when the timed region starts).  One plane-sized block through page-locked memory, best of three."""
    import torch
    try:
        host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    except Exception as e:              # noqa: BLE001
        return {'error': repr(e)[:120]}
    devt = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    best = {}
    for name, fn in (('h2d', lambda: devt.copy_(host, non_blocking=True)),
                     ('d2h', lambda: host.copy_(devt, non_blocking=True))):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        best[name + '_GBs'] = nbytes / min(ts) / 1e9
    best['block_bytes'] = nbytes
    best['note'] = 'page-locked host memory, one 1 GiB copy, best of three; not part of `value`'
    del devt, host
    return best


VALU_PK_F32_TADD = 78.6        # packed float32 additions per second, 10^12 (MI355X_MICROARCH.md: 157.3 TFLOP/s FMA rate)


def valu_roofline(r, adds_per_launch, what):
    """Turns an HBM roofline block into that of a kernel bound by vector issue (non-local
    means is not HBM-bound): `bound` = valu, achieved = dependent float32 additions per second against the
    packed-float32 add rate; the HBM figures stay as `hbm`."""
    t = r['kernel_ms'] * 1e-3
    out = dict(r)
    out['hbm'] = {'achieved': r['achieved'], 'peak': r['peak'], 'unit': r['unit'], 'frac': r['frac']}
    out.update({'bound': 'valu', 'achieved': adds_per_launch / t / 1e12, 'peak': VALU_PK_F32_TADD,
                'unit': 'T float32 additions/s', 'frac': adds_per_launch / t / 1e12 / VALU_PK_F32_TADD,
                'algorithmic_additions_per_launch': int(adds_per_launch), 'valu_note': what})
    return out


def boundary_check(w, out, dist, rank, world, rdev, core_rows=32, cols=2048):
    """N > 1: the rows either side of every shard boundary, recomputed UNSHARDED on rank 0 and compared
    with what the two neighbour ranks produced -- byte for byte (change map) and bit for bit (filtered
    values of the pipeline, whose rows next to a boundary depend on the other rank's rows: the halo
    exchange).  Every rank sends its first and last rows (inputs with the filter's halo, outputs) to
    rank 0 point to point.  -> dict on rank 0 (None elsewhere); raises on any difference."""
    import torch
    halo = w.halo_rows
    hx = halo                                   # the filter's reach along x equals its reach along y here
    cols = min(cols, w.nx - hx) if w.nx > hx + 8 else w.nx
    cin = min(cols + hx, w.nx)
    nin = min(core_rows + halo, w.rows)
    nco = min(core_rows, w.rows)
    filt = getattr(w, 'filtered', None) if w.name == 'pipeline' else None

    def pack(top):
        sl = slice(0, nco) if top else slice(w.rows - nco, w.rows)
        d = {'in': w.edge_inputs(top, nin, cin), 'map': out[sl, :cols].contiguous()}
        if filt is not None:
            d['filt'] = filt[:, :, sl, :cols].contiguous()
        return d

    mine = {'top': pack(True), 'bot': pack(False)}
    names = ['in', 'map'] + (['filt'] if filt is not None else [])
    if rank != 0:
        for side in ('top', 'bot'):
            for n_ in names:
                dist.send(mine[side][n_].to(rdev), dst=0)
        return None
    edges = {0: mine}
    for r in range(1, world):
        edges[r] = {}
        for side in ('top', 'bot'):
            edges[r][side] = {}
            for n_ in names:
                t = torch.empty_like(mine[side][n_], device=rdev)
                dist.recv(t, src=r)
                edges[r][side][n_] = t.to(w.dev)
    bad_map = bad_filt = compared = 0
    for r in range(1, world):
        up, dn = edges[r - 1]['bot'], edges[r]['top']
        band = torch.cat([up['in'], dn['in']], dim=2)            # 2 (core + halo) rows around the boundary
        ch, fl = w.band_result(band, (nin - nco, 2 * nco), cols)
        want_map = torch.cat([up['map'], dn['map']], dim=0)
        bad_map += int((ch != want_map).sum().item())
        compared += int(want_map.numel())
        if fl is not None:
            want_f = torch.cat([up['filt'], dn['filt']], dim=2)
            bad_filt += int((fl.view(torch.int32) != want_f.view(torch.int32)).sum().item())
    res = {'boundaries': world - 1, 'rows_each_side': nco, 'columns': cols, 'map_bytes_compared': compared,
           'map_bytes_differing': bad_map, 'filtered_values_differing': bad_filt if filt is not None else None,
           'note': 'rank 0 recomputes the rows around every shard boundary unsharded from the neighbours\' '
                   'inputs and compares with their outputs'}
    if bad_map or bad_filt:
        raise RuntimeError('sharded result differs from the unsharded one at a shard boundary: %r' % res)
    return res


def roofline(kernel, avg_ms, alg_bytes, note=None, traffic=None, traffic_source=None):
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    r = {'kernel': kernel, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'kernel_ms': avg_ms,
         'algorithmic_bytes_per_launch': int(alg_bytes), 'traffic': traffic}
    if traffic_source:
        r['traffic_source'] = traffic_source
    if note:
        r['note'] = note
    return r


# ------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------
class Workload:
    """rows [r0, r1) of the job's raster on this rank."""

    def __init__(self, a, rank, world, dev):
        from nd_amd import tiles
        self.a, self.rank, self.world, self.dev = a, rank, world, dev
        self.k, self.nx = a.k, a.nx
        self.global_ny = a.ny if a.scaling == 'strong' else a.ny * world
        self.r0, self.r1 = tiles.row_partition(self.global_ny, world)[rank]
        self.rows = self.r1 - self.r0
        self.npix = self.rows * self.nx


    # ---- N > 1 self-check (boundary_check below): the rows either side of a shard boundary ----
    halo_rows = 0

    def edge_inputs(self, top, nrows, cols):
        """this rank's first (top) or last `nrows` rows of every input plane, columns [0, cols)"""
        st = self.inputs()
        sl = slice(0, nrows) if top else slice(self.rows - nrows, self.rows)
        return st[:, :, sl, :cols].contiguous()

    def band_result(self, band, core, cols):
        """the unsharded computation on a band of input rows (var, k, rows, cols + halo): the change map
        of its `core` = (first row, rows) restricted to columns [0, cols) -- and, for the pipeline, the
        filtered values there"""
        raise NotImplementedError


class OmnibusC2(Workload):
    name, dom = 'omnibus', 'omnibus_c2_global'

    def inputs(self):
        return self.stack

    def band_result(self, band, core, cols):
        from nd_amd import kernels
        ch = kernels.change_detection(band[0], band[1], band[2], band[3], alpha=self.a.alpha, n=self.a.looks)
        return ch[core[0]:core[0] + core[1], :cols], None

    def __init__(self, a, rank, world, dev):
        super().__init__(a, rank, world, dev)
        from nd_amd import synth
        self.stack = synth.wishart_c2_stack(self.k, self.rows, self.nx, looks=a.looks,
                                            seed=1234 + rank, device=dev, change_frac=a.change_frac)
        # 384 B/px read (k = 24 float32 x 4 planes) + k B/px change map written by the same kernel
        self.alg_bytes = self.npix * self.k * (4 * self.stack.element_size() + 1)
        self.read_bytes = self.npix * self.k * 4 * self.stack.element_size()

    def step(self):
        from nd_amd import tiles
        return tiles.omnibus_rows(self.stack, self.a.alpha, self.a.looks)

    def metric(self):
        return 'Mpixels/s OmnibusTest dual-pol %dt x %d x %d' % (self.k, self.a.ny, self.nx)

    def config(self):
        return {'workload': 'OmnibusTest C2 %dt x %d x %d f32 %s (BASELINE configs[1]), resident in HBM'
                            % (self.k, self.a.ny, self.nx, 'per GPU' if self.a.scaling == 'weak' else 'in all'),
                'looks': self.a.looks, 'alpha': self.a.alpha, 'change_frac': self.a.change_frac}

    def describe(self):
        return ('OmnibusTest dual-pol C2, synthetic %dt x %d x %d float32 %s (BASELINE.json '
                'configs[1]), n=%d looks, alpha=%g, %.3g of pixels with a x4 step; inputs resident '
                'in HBM' % (self.k, self.a.ny, self.nx,
                            'per GPU' if self.a.scaling == 'weak' else 'in all, rows split over the GPUs',
                            self.a.looks, self.a.alpha, self.a.change_frac))

    def check(self, out, nsample=0):
        from oracle import checks
        return checks.omnibus_sample(self.stack, out, self.a.alpha, self.a.looks, nsample=nsample,
                                     rows=(0, self.rows - 1), seed=4)


class OmnibusC3(Workload):
    name, dom = 'c3', 'omnibus_c2_global'          # the C3 pass A reports under the same id

    def inputs(self):
        return self.stack

    def band_result(self, band, core, cols):
        from nd_amd import kernels
        ch = kernels.change_detection_c3([band[c] for c in range(9)], alpha=self.a.alpha, n=self.a.looks)
        return ch[core[0]:core[0] + core[1], :cols], None

    def __init__(self, a, rank, world, dev):
        super().__init__(a, rank, world, dev)
        from nd_amd import synth
        # (under the profiler's counter pass only six dates are drawn and repeated: the full synthesis is
        # 17 000 small launches, at which rocprofv3 --pmc segfaults -- unless the counters are restricted to
        # this library's kernels (--kernel-include-regex nd_amd), which tools/collect_traffic.sh does since
        # round 6 with ND_AMD_TRAFFIC_FULL_SYNTH=1: six repeated dates are NOT the workload -- 575 527
        # candidates instead of 168 025 at the benchmark's threshold, 3.4 x the traffic of pass B)
        self.stack = synth.wishart_c3_stack(self.k, self.rows, self.nx, looks=a.looks,
                                            seed=4321 + rank, device=dev, change_frac=a.change_frac,
                                            cycle=6 if (TRAFFIC_MODE and not FULL_SYNTH) else 0)
        self.alg_bytes = self.npix * self.k * (9 * self.stack.element_size() + 1)

    def step(self):
        from nd_amd import tiles
        return tiles.omnibus_c3_rows(self.stack, self.a.alpha, self.a.looks)

    def metric(self):
        return 'Mpixels/s OmnibusTest full-pol C3 %dt x %d x %d' % (self.k, self.a.ny, self.nx)

    def config(self):
        return {'workload': 'OmnibusTest C3 %dt x %d x %d f32 x 9 planes %s (share of BASELINE configs[3])'
                            % (self.k, self.a.ny, self.nx, 'per GPU' if self.a.scaling == 'weak' else 'in all'),
                'looks': self.a.looks, 'alpha': self.a.alpha, 'change_frac': self.a.change_frac}

    def describe(self):
        return ('OmnibusTest full-pol C3 (extension, no reference implementation), synthetic '
                '%dt x %d x %d float32 x 9 planes %s (one GPU\'s share of BASELINE.json configs[3]), '
                'n=%d looks, alpha=%g' % (self.k, self.a.ny, self.nx,
                                          'per GPU' if self.a.scaling == 'weak' else 'in all',
                                          self.a.looks, self.a.alpha))

    def check(self, out, nsample=20000):
        from oracle import checks
        return checks.omnibus_sample(self.stack, out, self.a.alpha, self.a.looks, nsample=nsample,
                                     rows=(), seed=4, pol=3)


class Pipeline(Workload):
    name, dom = 'pipeline', 'nlmeans_tiled'
    halo_rows = TUT['r'][1] + TUT['f'][1]

    def inputs(self):
        return self.shard.core

    def band_result(self, band, core, cols):
        import torch
        from nd_amd import kernels, tiles
        # the kernel itself on the whole band (no process group: this IS the unsharded computation)
        filt = torch.empty_like(band)
        kernels.pixelwise_nlmeans_3d(band.permute(1, 2, 3, 0), filt.permute(1, 2, 3, 0), TUT['r'], TUT['f'],
                                     TUT['sigma'], TUT['h'], TUT['n_eff'], patch_mode=self.a.patch_mode)
        filt = filt[:, :, core[0]:core[0] + core[1], :].contiguous()
        ch = tiles.omnibus_rows(filt, self.a.alpha, TUT['n'])
        return ch[:, :cols], filt[..., :cols]

    def __init__(self, a, rank, world, dev):
        super().__init__(a, rank, world, dev)
        from nd_amd import synth, tiles
        self.halo = TUT['r'][1] + TUT['f'][1]
        self.shard = tiles.empty_shard((4, self.k), self.global_ny, self.nx, self.halo, dev,
                                       rank=rank, world=world)
        if TRAFFIC_MODE and not FULL_SYNTH:
            # (under the profiler's counter passes the Wishart synthesis -- thousands of small launches, each
            #  with its counters read out -- takes longer than the run may stay silent; the filter's traffic does
            #  not depend on the values: uniform planes, three launches per date)
            import torch
            g_ = torch.Generator(device=dev).manual_seed(99 + rank)
            for t_ in range(self.k):
                self.shard.core[:, t_] = torch.rand((4, self.rows, self.nx), generator=g_, device=dev) + 0.5
                self.shard.core[1:3, t_] -= 1.0
                self.shard.core[1:3, t_] *= 0.3
        else:
            synth.wishart_c2_stack(self.k, self.rows, self.nx, looks=a.looks, seed=99 + rank,
                                   change_frac=a.change_frac, out=self.shard.core)
        if world == 1:
            self.stack = self.shard.core
        # the filter reads and writes every (variable, date, pixel) once: 8 B per px.t.var
        ext_px = self.shard.ext.shape[2] * self.nx
        self.alg_bytes = ext_px * self.k * 4 * 8
        self.filtered = None
        import torch
        # find_weight's 'No solution' flag of n_eff = 50: collected on the device, read once after
        # the timed region (a step has no host synchronisation)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.status_any = torch.zeros(1, dtype=torch.int32, device=dev)

    overlap = None            # None: the library's default form; first_contact() may force False

    def filter_once(self, overlap=None):
        from nd_amd import tiles
        return tiles.nlmeans_rows(self.shard, self.global_ny, TUT['r'], TUT['f'],
                                  TUT['sigma'], TUT['h'], n_eff=TUT['n_eff'],
                                  patch_mode=self.a.patch_mode, status=self.status, overlap=overlap)

    def step(self):
        from nd_amd import tiles
        self.filtered = self.filter_once(self.overlap)
        self.status_any.bitwise_or_(self.status)
        return tiles.omnibus_rows(self.filtered, self.a.alpha, TUT['n'])

    def comm(self, steps=5):
        """halo bytes this rank sends per step and the duration of the exchange by itself (events on
        the launch stream around `steps` exchanges with nothing to overlap them)"""
        import torch
        from nd_amd import tiles
        h = tiles.exchange_halo_begin(self.shard)
        nbytes = h.nbytes
        h.wait()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            tiles.exchange_halo_(self.shard)
        e1.record()
        torch.cuda.synchronize()
        return {'halo_bytes_sent_per_step': int(nbytes), 'exchange_ms_alone': e0.elapsed_time(e1) / steps}

    def metric(self):
        return 'Mpixels/s NLMeans->OmnibusTest %dt x %d x %d' % (self.k, self.a.ny, self.nx)

    def config(self):
        return {'workload': 'NLMeansFilter(r=(1,3,3), f=1, sigma=1, h=1, n_eff=50) -> OmnibusTest(n=50) on '
                            '%dt x %d x %d f32 x 4 variables %s (share of BASELINE configs[4])'
                            % (self.k, self.a.ny, self.nx, 'per GPU' if self.a.scaling == 'weak' else 'in all'),
                'alpha': self.a.alpha, 'patch_mode': self.a.patch_mode, 'change_frac': self.a.change_frac}

    def describe(self):
        return ('NLMeansFilter(dims=(time,y,x), r=(1,3,3), f=1, sigma=1, h=1, n_eff=50, patch '
                'distances %s) -> OmnibusTest(n=50, alpha=%g) (examples/tutorial_s1.ipynb cells 11, '
                '15) on synthetic %dt x %d x %d float32 x 4 variables %s (one GPU\'s share of '
                'BASELINE.json configs[4]); %s'
                % ('as compiled (reference)' if self.a.patch_mode == 0 else 'signed', self.a.alpha,
                   self.k, self.a.ny, self.nx, 'per GPU' if self.a.scaling == 'weak' else 'in all',
                   'halo rows exchanged with the neighbour ranks over RCCL every step'
                   if self.world > 1 else 'single rank: no exchange'))

    def check(self, out, nsample=0, light=False):
        from nd_amd import kernels
        from oracle import checks
        kernels.raise_if_no_solution(self.status_any)
        if light:
            # the `secondary` block: the corner crops of the two edge bands and one interior crop
            res = checks.nlmeans_crops(self.stack, self.filtered, TUT['r'], TUT['f'], TUT['sigma'],
                                       TUT['h'], TUT['n_eff'], self.a.patch_mode,
                                       [(0, 0), (self.rows, 0), (self.rows // 2, self.nx // 3)],
                                       size=(8, 256), then_omnibus=(self.a.alpha, TUT['n']), change=out)
            res['bands'] = 'three 8 x 256 crops: the first and last rows at column 0, one in the interior'
            return res
        # the two edge bands of the tile whole (rows 0 .. 7 and the last 8, every column: with neighbour
        # ranks these are the rows the separate edge launches compute) and a crop from the interior
        res = checks.nlmeans_crops(self.stack, self.filtered, TUT['r'], TUT['f'], TUT['sigma'],
                                   TUT['h'], TUT['n_eff'], self.a.patch_mode, [(0, 0), (self.rows, 0)],
                                   size=(8, self.nx), then_omnibus=(self.a.alpha, TUT['n']), change=out)
        mid = checks.nlmeans_crops(self.stack, self.filtered, TUT['r'], TUT['f'], TUT['sigma'],
                                   TUT['h'], TUT['n_eff'], self.a.patch_mode, [(self.rows // 2, self.nx // 3)],
                                   size=(8, 64), then_omnibus=(self.a.alpha, TUT['n']), change=out)
        for key in ('bad', 'compared', 'change_bad', 'change_compared'):
            res[key] += mid[key]
        res['max_rel'] = max(res['max_rel'], mid['max_rel'])
        res['bands'] = 'rows 0-7 and the last 8, full width; one 8 x 64 interior crop'
        return res


WORKLOADS = {'omnibus': OmnibusC2, 'c3': OmnibusC3, 'pipeline': Pipeline}


# ------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle timed on the host's cores, bounded samples
# ------------------------------------------------------------------------------------------
def cpu_baseline_omnibus(w, out):
    import numpy as np
    from oracle import oracle as O
    O.build()
    a = w.a
    stack = w.stack
    rows = min(a.cpu_rows, stack.shape[2])
    host = stack[:, :, :rows, :].cpu().numpy()                 # (4, k, rows, nx)
    planes = [np.moveaxis(host[v], 0, -1) for v in range(4)]   # (rows, nx, k) strided views
    cores = _usable_cores()
    # warm (page-in + thread pool) on a sliver, then time
    O.change_detection_planes([p[:8] for p in planes], a.alpha, a.looks, njobs=cores)
    t0 = time.perf_counter()
    ch = O.change_detection_planes(planes, a.alpha, a.looks, njobs=cores)
    dt = time.perf_counter() - t0
    # one thread = the reference as shipped (its prange is serial: setup.py never passes -fopenmp)
    rows1 = min(rows, 1024)
    t0 = time.perf_counter()
    O.change_detection_planes([p[:rows1] for p in planes], a.alpha, a.looks, njobs=1)
    dt1 = time.perf_counter() - t0
    npx = rows * stack.shape[3]
    res = {
        'value': npx / dt / 1e6, 'unit': 'Mpixels/s', 'cores': int(cores), 'kind': 'port',
        'sample_short': 'oracle/nd_oracle.c (OpenMP) on the first %d rows of the same stack, %.2f s; '
                        'one_thread_value: %d rows, %.2f s' % (rows, dt, rows1, dt1),
        'sample': 'oracle/nd_oracle.c (C port of nd/_change.pyx, reference-order arithmetic, '
                  'OpenMP over rows) on the first %d rows x %d cols x %d dates of the same '
                  'stack: %.2f s wall' % (rows, stack.shape[3], stack.shape[1], dt),
        'one_thread': {'value': rows1 * stack.shape[3] / dt1 / 1e6, 'unit': 'Mpixels/s', 'cores': 1,
                       'sample': 'first %d rows, %.2f s wall (the reference as shipped runs its '
                                 'prange serially)' % (rows1, dt1)},
        'flagged_fraction': float((ch.sum(axis=2) > 0).mean()),
        'gpu_matches_cpu_on_sample': bool((out[:rows].cpu().numpy() == ch).all()),
    }
    return res


# ------------------------------------------------------------------------------------------
# secondary workloads (rank 0, N = 1)
# ------------------------------------------------------------------------------------------
def _free():
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()


EXTRA_KEYS = ('omnibus_a0.01', 'omnibus_a0.0001', 'omnibus_a0.2', 'ml3', 'ml5', 'pm_a0.99', 'pm_a0.01', 'c3_a0.99', 'c3_a0.01',
              'c3_pm_a0.99',
              'boxcar3', 'boxcar5', 'gauss1', 'nlm_pm0', 'nlm_pm1', 'pipeline')


def extras(main, barrier, dev, only=None, light=False):
    """The other kernels of the path.  only = one of EXTRA_KEYS: just that workload (--traffic-run).
    light: the `secondary` block of the default run -- fewer steps, no CPU baselines, smaller oracle samples."""
    import numpy as np
    import torch
    from nd_amd import kernels
    from oracle import checks
    from oracle import oracle as O
    out = []
    cores = _usable_cores()
    quick = TRAFFIC_MODE

    def want(*keys):
        if only is None or isinstance(only, str):
            return only is None or only in keys
        return any(k_ in only for k_ in keys)

    def roof(key, hint, dom, km, alg_bytes, note=None):
        traffic, source = profiled_traffic(key, hint)
        return roofline(dom, km[dom], alg_bytes, note=note, traffic=traffic, traffic_source=source)

    def entry(key, workload, dt, steps, npix, km, roof_, match, **more):
        e = {'key': key, 'workload': workload, 'ms': dt / steps * 1e3, 'Mpx_per_s': npix * steps / dt / 1e6,
             'kernels_ms': km, 'roofline': roof_, 'matches_oracle_on_sample': match}
        e.update(more)
        out.append(e)

    a = main.a
    # -- OmnibusTest at the thresholds users actually pass: the reference default and the tutorial's
    for alpha in (0.01, 1e-4, 0.2):
        key = 'omnibus_a%g' % alpha
        if not want(key):
            continue
        fn = lambda: kernels.change_detection(*main.stack, alpha=alpha, n=a.looks)       # noqa: E731
        dt, km, ch = timed_extra(fn, 10, 10, barrier)
        if quick:
            continue
        res = checks.omnibus_sample(main.stack, ch, alpha, a.looks, nsample=20000, seed=8)
        dom = max(km, key=km.get)
        entry(key, 'OmnibusTest C2 %dt x %d x %d f32, alpha=%g (dense regime: %.3f of pixels change)'
              % (main.k, main.rows, main.nx, alpha, res['flagged_fraction']), dt, 10, main.npix, km,
              roof(key, 'chain_kernel' if alpha > 0.007 else 'stream_kernel', dom, km, main.alg_bytes,
                   note=('search fused into pass A, two linear passes over the retained series (dense_chain)'
                         if alpha > 0.007 else
                         'search fused into the streaming pass over the planes (vector issue and HBM both '
                         'near their floors)') + '; bytes = planes read once + change map written once'),
              res['bad'] == 0, sample=res)
        del ch

    # -- OmnibusTest(ml=w): boxcar multilooking fused into the test (nd/change.py:61-69), the planes read once
    for mlw in (3, 5):
        key = 'ml%d' % mlw
        if not want(key):
            continue
        alpha = 0.99                 # the benchmark's threshold (sparse regime: the fused kernel's)
        fn = lambda: kernels.change_detection_multilooked(*main.stack, alpha=alpha, ml=mlw)       # noqa: E731
        dt, km, ch = timed_extra(fn, 10, 10, barrier)
        if quick:
            continue
        kern = (np.ones((mlw, mlw)) / mlw ** 2).reshape(1, 1, mlw, mlw)
        mlk = kernels.convolve(main.stack, kern)
        two = kernels.change_detection(mlk[0], mlk[1], mlk[2], mlk[3], alpha=alpha, n=mlw * mlw)
        torch.cuda.synchronize()
        same = bool(torch.equal(ch, two))                 # whole raster against boxcar kernel + plain test
        del mlk, two
        ny_ = main.rows
        res = checks.omnibus_ml_bands(main.stack, ch, mlw, alpha,
                                      [(0, 12), (ny_ // 3, 12), (ny_ // 2 + 5, 12), (ny_ - 12, 12)])
        dom = max(km, key=km.get)
        # window sums: w^2 dependent float64 additions per value (scipy's order), 96 values per pixel
        adds = main.npix * main.k * 4 * mlw * mlw
        entry(key, 'OmnibusTest(ml=%d) C2 %dt x %d x %d f32, alpha=%g: %d x %d boxcar multilooking fused into '
              'the test (n = %d looks)' % (mlw, main.k, main.rows, main.nx, alpha, mlw, mlw, mlw * mlw), dt, 10,
              main.npix, km,
              roof(key, 'ml_kernel', dom, km, main.alg_bytes,
                   note='planes read once through LDS (16-byte LDS-DMA), window sums in scipy\'s order, series '
                        'retained in registers; bound by vector issue, not memory: %.1f G dependent float64 '
                        'additions per launch = %.2f T/s against 39.3 T/s (one per lane and 4 cycles)'
                        % (adds / 1e9, adds / (km[dom] * 1e-3) / 1e12)),
              res['bad'] == 0 and same, sample=res, equals_two_step_path_on_whole_raster=same,
              valu={'bound': 'f64 add issue', 'achieved_Tadd_per_s': adds / (km[dom] * 1e-3) / 1e12,
                    'peak_Tadd_per_s': 39.3, 'frac': adds / (km[dom] * 1e-3) / 1e12 / 39.3})
        del ch

    # -- the same test on data in the reference's own layout ((y, x, time), C12 interleaved complex):
    #    what OmnibusTest.apply(ds) receives from a reference-layout dataset
    if want('pm_a0.99', 'pm_a0.01'):
        yxt = [main.stack[v].permute(1, 2, 0).contiguous() for v in range(4)]
        c12 = torch.complex(yxt[1], yxt[2])
        pmv = (yxt[0], c12.real, c12.imag, yxt[3])
        del yxt
        for alpha in (0.99, 0.01):
            key = 'pm_a%g' % alpha
            if not want(key):
                continue
            fn = lambda: kernels.change_detection_pixel_major(*pmv, alpha=alpha, n=a.looks)      # noqa: E731
            dt, km, ch = timed_extra(fn, 10, 10, barrier)
            if quick:
                continue
            ref = kernels.change_detection(*main.stack, alpha=alpha, n=a.looks)
            same = bool(torch.equal(ch, ref))
            res = checks.omnibus_sample(main.stack, ch, alpha, a.looks, nsample=20000, seed=9)
            dom = max(km, key=km.get)
            entry(key, 'OmnibusTest C2 %dt x %d x %d f32 in the reference layout (y, x, time), C12 complex64, '
                  'alpha=%g' % (main.k, main.rows, main.nx, alpha), dt, 10, main.npix, km,
                  roof(key, 'pm_dma', dom, km, main.alg_bytes,
                       note='LDS-DMA staging of pixel-major spans (global_load_lds_dwordx4)' if alpha > 0.5 else
                       'dense_chain behind the staging: C12 through an LDS-DMA image, C11 / C22 straight '
                       'into registers (half the LDS, twice the waves per CU)'),
                  res['bad'] == 0 and same, sample=res, equal_to_planar_map=same)
            del ch, ref
        del c12, pmv
        _free()

    # -- full-pol C3, config 4's single-GPU share
    class A:
        pass
    if want('c3_a0.99', 'c3_a0.01', 'c3_pm_a0.99'):
        a3 = A()
        a3.__dict__.update(a.__dict__)
        a3.k, a3.ny, a3.nx, a3.alpha, a3.scaling = 48, 1024, 8192, 0.99, 'weak'
        w = OmnibusC3(a3, 0, 1, dev)
        for alpha in (0.99, 0.01):      # the benchmark's threshold, then the reference's default
            key = 'c3_a%g' % alpha
            if not want(key):
                continue
            a3.alpha = alpha
            nst = 5 if light else 10
            dt, km, ch = timed_extra(w.step, nst, 3 if light else 10, barrier)
            if quick:
                continue
            res = w.check(ch, nsample=5000 if light else 20000)
            dom = w.dom if (alpha > 0.5 and w.dom in km) else max(km, key=km.get)
            entry(key, w.describe(), dt, nst, w.npix, km,
                  roof(key, 'omnibus_c3_retain' if alpha > 0.5 else 'omnibus_c3_stream', dom, km, w.alg_bytes,
                       note=None if alpha > 0.5 else 'search fused into the streaming pass (omnibus_c3_stream_kernel)'),
                  res['bad'] == 0, sample=res, alg_bytes=w.alg_bytes)
            del ch
        # -- the same test on data in the reference's layout: nine (y, x, time) variables, the off-diagonals as
        #    interleaved complex arrays (what OmnibusTest(pol='full').apply(ds) receives)
        if want('c3_pm_a0.99'):
            yxt = [w.stack[c].permute(1, 2, 0).contiguous() for c in range(9)]
            cplx = [torch.complex(yxt[c], yxt[c + 1]) for c in (3, 5, 7)]
            pmv = yxt[:3] + [h for z_ in cplx for h in (z_.real, z_.imag)]
            del yxt
            fn = lambda: kernels.change_detection_c3_pixel_major(pmv, alpha=0.99, n=a.looks)      # noqa: E731
            dt, km, ch = timed_extra(fn, 10, 10, barrier)
            if not quick:
                a3.alpha = 0.99
                ref = w.step()
                same = bool(torch.equal(ch, ref))
                res = w.check(ch)
                dom = max(km, key=km.get)
                entry('c3_pm_a0.99', 'OmnibusTest full-pol C3 48t x 1024 x 8192 f32 in the reference layout (y, x, time), '
                      'C12 / C13 / C23 complex64, alpha=0.99', dt, 10, w.npix, km,
                      roof('c3_pm_a0.99', 'omnibus_c3_pm', dom, km, w.alg_bytes,
                           note='LDS images of the contiguous per-pixel runs folded in place (16 pixels per wave); pass B '
                                'reads the listed series where they lie'),
                      res['bad'] == 0 and same, sample=res, equal_to_planar_map=same)
                del ref
            del ch, pmv, cplx
        del w
        _free()

    g = torch.Generator(device=dev).manual_seed(7)
    # -- boxcar 3x3 / 5x5 (the multilooking step in front of the test) and the fused Gaussian, 24t x 4096 x 4096
    if want('boxcar3', 'boxcar5', 'gauss1'):
        x = torch.rand((24, 4096, 4096), generator=g, device=dev) + 0.5
        y = torch.empty_like(x)
        for wdt in (3, 5):
            key = 'boxcar%d' % wdt
            if not want(key):
                continue
            kern = np.ones((1, wdt, wdt)) / float(wdt * wdt)
            # (sub-millisecond kernels behind host-side checks: the device needs ~10 ms of work to be
            # back at its sustained clocks -- 40 launches back to back are flat from the second on,
            # tools/exp_launch_times.py)
            dt, km, _ = timed_extra(lambda: kernels.convolve(x, kern, out=y), 20, 20, barrier)
            if quick:
                continue
            res = checks.convolve_bands(x, y, kern[0], [(0, 40), (2030, 2070), (4056, 4096)], [0, 23])
            dom = max(km, key=km.get)
            e = dict(sample=res)
            if wdt == 5:
                crop = np.ascontiguousarray(x[:, :2048, :2048].cpu().numpy())
                t0 = time.perf_counter()
                O.convolve_reflect_mt(crop, kern, njobs=cores)
                dtc = time.perf_counter() - t0
                e['cpu_baseline'] = {'value': crop.size / dtc / 1e6, 'unit': 'M px.t/s', 'cores': cores,
                                     'kind': 'port', 'sample': '24 x 2048 x 2048 crop, %.2f s' % dtc}
            entry(key, 'BoxcarFilter %dx%d on 24t x 4096 x 4096 f32 (scipy.ndimage.convolve arithmetic)'
                  % (wdt, wdt), dt, 20, x.numel(), km, roof(key, 'correlate', dom, km, 8 * x.numel()),
                  res['bad'] == 0, unit_note='Mpx_per_s counts px.t', **e)
        # -- GaussianFilter(dims=('y', 'x'), sigma=1): both passes in one kernel
        if want('gauss1'):
            import scipy.ndimage as ndi
            # (warm-up launches first: the host-side baseline just above leaves the GPU at idle clocks)
            # (two batches of 20, BOTH reported; `ms` is their mean.  The pause of tens of milliseconds that
            # used to fall into the first batch was the box's CPU quota after the all-core baseline just
            # above, see timed().)
            batches = [timed(lambda: kernels.gaussian_filter(x, (0, 1.0, 1.0), out=y), 20, 20, barrier)
                       for i in range(2)]
            dt = sum(b_[0] for b_ in batches) / len(batches)
            km = batches[0][1]
            if not quick:
                res = checks.gaussian_bands(x, y, 1.0, [(0, 40), (2030, 2070), (4056, 4096)], [0, 23])
                dom = max(km, key=km.get)
                crop = np.ascontiguousarray(x[:4, :2048, :2048].cpu().numpy())
                t0 = time.perf_counter()
                ndi.gaussian_filter(crop, (0, 1.0, 1.0))
                dtc = time.perf_counter() - t0
                entry('gauss1', 'GaussianFilter sigma=1 (9 taps along y, then along x) on 24t x 4096 x 4096 f32 '
                      '(scipy.ndimage.gaussian_filter arithmetic, float32 intermediate)', dt, 20, x.numel(), km,
                      roof('gauss1', 'correlate1d', dom, km, 8 * x.numel(),
                           note='one read and one write of the array for both passes'),
                      res['bad'] == 0, unit_note='Mpx_per_s counts px.t', sample=res,
                      batches_ms=[b_[0] / 20 * 1e3 for b_ in batches],
                      cpu_baseline={'value': crop.size / dtc / 1e6, 'unit': 'M px.t/s', 'cores': 1,
                                    'kind': 'reference', 'sample': 'scipy.ndimage.gaussian_filter (the reference\'s '
                                    'own arithmetic for this filter) on a 4 x 2048 x 2048 crop, %.2f s' % dtc})
        del x, y
        _free()

    # -- non-local means, BASELINE config 3: 7x7 patch / 21x21 search, 12t x 4096 x 4096
    if want('nlm_pm0', 'nlm_pm1'):
        k, ny, nx = 12, 4096, 4096
        x = torch.empty((1, k, ny, nx), device=dev)
        for t in range(k):
            u = torch.rand((4, ny, nx), generator=g, device=dev)
            # Gamma(4, 0.25).  torch.rand draws from [0, 1): a zero (one draw in 2^24) would put an
            # infinity into a variate that has none, and every pixel within r + f of it would take the
            # kernel's exact per-pixel path
            x[0, t] = -0.25 * torch.log(u.clamp_min_(2.0 ** -25)).sum(dim=0)
        y = torch.empty_like(x)
        r, f = (0, 10, 10), (0, 3, 3)
        for pm in (0, 1):
            key = 'nlm_pm%d' % pm
            if not want(key):
                continue
            steps = (3 if pm == 0 else 1) if light else (5 if pm == 0 else 2)
            fn = lambda: kernels.pixelwise_nlmeans_3d(x.permute(2, 3, 1, 0), y.permute(2, 3, 1, 0),    # noqa: E731
                                                      (10, 10, 0), (3, 3, 0), 0.5, 0.5, -1, patch_mode=pm)
            dt, km, _ = timed(fn, steps, 1, barrier)
            if quick:
                continue
            res = checks.nlmeans_crops(x, y, r, f, 0.5, 0.5, -1, pm, [(2040, 3000), (0, 0)], size=(6, 48))
            dom = max(km, key=km.get)
            nq = 21 * 21 - 1
            flop = x.numel() * nq * (49 * 3 + 8) if pm else x.numel() * nq * 2
            e = dict(sample=res, TFLOPs_naive_formula=flop / (dt / steps) / 1e12)
            if pm == 1 and not light:
                crop = np.ascontiguousarray(x[:, :1, :640, :640].permute(2, 3, 1, 0).cpu().numpy())
                o = np.empty_like(crop)
                t0 = time.perf_counter()
                O.pixelwise_nlmeans_3d(crop, o, (10, 10, 0), (3, 3, 0), 0.5, 0.5, -1, njobs=cores, patch_mode=1)
                dtc = time.perf_counter() - t0
                e['cpu_baseline'] = {'value': 640 * 640 / dtc / 1e6, 'unit': 'M px.t/s', 'cores': cores,
                                     'kind': 'port', 'sample': '640 x 640 crop of one date, %.2f s' % dtc}
            entry(key, 'NLMeansFilter 7x7 patch / 21x21 search on 12t x 4096 x 4096 f32, patch distances %s'
                  % ('as compiled (reference: window mean)' if pm == 0 else 'signed (true patch distances)'),
                  dt, steps, x.numel(), km,
                  valu_roofline(roof(key, 'nlmeans', dom, km, 8 * x.numel(),
                                     note='HBM traffic is 8 B per px.t: never the bound'),
                                x.numel() * nq * (1 if pm == 0 else 15),
                                '440 dependent float32 additions per output in the reference\'s visiting order '
                                '(packed over two outputs)' if pm == 0 else
                                '440 search offsets x 15 float32 operations per output in the algorithm the kernel '
                                'implements (difference and square 2, 7-wide row sum shared across lanes 3, patch sum '
                                'over 7 row sums 6, exponent and exponential 2, weighted sum and weight total 2); the '
                                'naive formula, 49 x 3 per offset, is TFLOPs_naive_formula.  The row sums take DPP '
                                'operands, which have no packed form: the vector ALUs are busy 97 % of the time '
                                '(profiles/r02_nlmeans_patch2_pmc_after.txt) at this fraction of the packed rate'),
                  res['bad'] == 0, unit_note='Mpx_per_s counts px.t', **e)
        del x, y
        _free()

    # -- the tutorial pipeline, config 5's single-GPU share
    if want('pipeline'):
        ap = A()
        ap.__dict__.update(a.__dict__)
        ap.k, ap.ny, ap.nx, ap.alpha, ap.scaling, ap.patch_mode = 24, 2048, 16384, TUT['alpha'], 'weak', 0
        w = Pipeline(ap, 0, 1, dev)
        # two warm-up steps: the filter's output alternates between two 12.9 GB buffers, and the first
        # allocation of each costs tens of milliseconds of hipMalloc
        dt, km, ch = timed(w.step, 3, 2, barrier)
        if not quick:
            res = w.check(ch, light=light)
            nq_t = 3 * 7 * 7 - 1                              # neighbours of the tutorial's window
            entry('pipeline', w.describe(), dt, 3, w.npix, km,
                  valu_roofline(roof('pipeline', 'nlmeans', w.dom, km, w.alg_bytes),
                                w.npix * w.k * 4 * nq_t,
                                '146 dependent float32 additions per output (three dates x 7 x 7 window, the '
                                'reference\'s visiting order), packed over two outputs'),
                  res['bad'] == 0 and res['change_bad'] == 0, sample=res)
        del w, ch
        _free()
    return out


SECONDARY = (('nlm_cc_pm0', 'nlm_pm0'), ('nlm_cc_pm1', 'nlm_pm1'), ('c3_share_a0.99', 'c3_a0.99'),
             ('pipeline_share', 'pipeline'),
             # the headline stack at the reference's DEFAULT threshold (nd/change.py:32, alpha = 0.01: 99 % of the
             # pixels change), planar and in the reference's own (y, x, time) layout with C12 complex64 --
             # the literal drop-in call
             ('c2_a0.01', 'omnibus_a0.01'), ('c2_yxt_a0.01', 'pm_a0.01'))


def secondary(main, barrier, dev):
    """The other BASELINE configs in the driver-run line, compactly: config 3 (non-local
    means 7 x 7 / 21 x 21 on 12t x 4096 x 4096, the reference-compatible and the signed patch distances), one
    GPU's share of config 4 (full-pol C3 48t x 1024 x 8192 at alpha = 0.99) and of config 5 (the tutorial
    pipeline on 24t x 2048 x 16384 x 4); and the headline stack at the reference's default alpha = 0.01,
    planar and in the reference's (y, x, time) layout.  -> ({key: [ms_per_step, frac, 'hbm' | 'valu', matches_oracle_on_sample]},
    the long entries for bench_detail.json).  frac is of the WHOLE step: algorithmic bytes per step against the
    HBM peak, or the algorithm's dependent float32 additions per step against the packed-add rate."""
    block, long_form = {}, []
    t0 = time.perf_counter()
    es = {e['key']: e for e in extras(main, barrier, dev, only=tuple(ek for _, ek in SECONDARY), light=True)}
    wall = time.perf_counter() - t0
    for key, ekey in SECONDARY:
        e = es[ekey]
        r = e['roofline']
        step_s = e['ms'] * 1e-3
        if r['bound'] == 'valu':
            frac = r['algorithmic_additions_per_launch'] / step_s / 1e12 / VALU_PK_F32_TADD
        else:
            frac = e.get('alg_bytes', r['algorithmic_bytes_per_launch']) / step_s / 1e9 / HBM_PEAK_GBS
        block[key] = [float('%.4g' % e['ms']), float('%.3g' % frac), r['bound'], bool(e['matches_oracle_on_sample'])]
        long_form.append(e)
    long_form.append({'wall_s_of_the_block': wall})
    return block, long_form


LINE_LIMIT = 4096           # bytes; the driver keeps only the tail of stdout and parses its last line
SECONDARY_LIMIT = 500       # bytes of the `secondary` block
LINE_LIMIT_WITH_SECONDARY = 1900    # the driver keeps a 2 000-character tail


def _r(x, nd=4):
    """numbers of the line: 6 significant digits are plenty and keep it short"""
    if isinstance(x, float):
        return float('%.6g' % x)
    if isinstance(x, dict):
        return {k: _r(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v) for v in x]
    return x


def headline(m):
    """The ONE line: the headline object only, from the measurements `m` (a dict main() fills).  Pure
    host code (tests/test_host_logic.py builds it from stub numbers).  Everything explanatory -- notes,
    device clocks, transfer rates, per-kernel second-loop figures, the secondary workloads -- goes to
    the sidecar files (bench_detail.json, bench_extras.json), never into the line."""
    roof = m['roofline']
    line = {
        'metric': m['metric'], 'value': m['value'], 'unit': 'Mpixels/s', 'n_gpus': m['n_gpus'],
        'steps': m['steps'], 'warmup': m['warmup'], 'ms_per_step': m['ms_per_step'],
        'step_ms': m.get('step_ms'),
        'higher_is_better': True, 'scaling': m['scaling'], 'vs_baseline': None,
        'dtype': 'f32', 'data': m['data'],
        'config': m['config'],
        'roofline': {k: roof[k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic',
                                          'traffic_source', 'kernel_ms', 'algorithmic_bytes_per_launch',
                                          'frac_read_write', 'hbm') if k in roof},
        'kernels_ms': m['kernels_ms'],
    }
    if m.get('cpu_baseline') is not None:
        c = m['cpu_baseline']
        line['cpu_baseline'] = {k: c[k] for k in ('value', 'unit', 'cores', 'kind', 'sample', 'one_thread_value',
                                                  'gpu_matches_cpu_on_sample') if k in c}
    if m.get('comm') is not None:
        line['comm'] = m['comm']
    if m.get('matches_oracle_on_sample') is not None:
        line['matches_oracle_on_sample'] = m['matches_oracle_on_sample']
    if m.get('secondary') is not None:
        line['secondary'] = m['secondary']
    for k in ('detail_file', 'extras_file'):
        if m.get(k):
            line[k] = m[k]
    return _r(line)


def emit(line):
    """strict JSON, one line, under LINE_LIMIT bytes -- or no line at all (a line the driver cannot
    parse leaves the round unmeasured: better to fail here, loudly)"""
    text = json.dumps(line, allow_nan=False, separators=(',', ':'))
    limit = LINE_LIMIT
    if line.get('secondary') is not None:
        limit = LINE_LIMIT_WITH_SECONDARY
        sec = json.dumps(line['secondary'], allow_nan=False, separators=(',', ':'))
        if len(sec.encode()) >= SECONDARY_LIMIT:
            raise RuntimeError('the secondary block is %d bytes (limit %d)' % (len(sec.encode()), SECONDARY_LIMIT))
    if '\n' in text or len(text.encode()) >= limit:
        raise RuntimeError('bench line is %d bytes (limit %d)' % (len(text.encode()), limit))
    return text


def _write_sidecar(name, obj):
    """bench_detail.json / bench_extras.json next to bench.py (and a copy under gpurun_out/ when that
    directory exists, so that a gpurun call brings it back).  -> file name, or None if not writable."""
    text = json.dumps(obj, indent=1, default=str)
    wrote = None
    for d in (ROOT, os.path.join(ROOT, 'gpurun_out')):
        if not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, name), 'w') as fh:
                fh.write(text)
            wrote = wrote or name
        except OSError:
            pass
    return wrote


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks the way the
    driver does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <same arguments>) as a CHILD process, before this process has made any GPU
    call (torch is not even imported here), relay rank 0's line and the launcher's exit code.  Never
    continues single-rank: the reference's multi-worker entry is one call as well
    (nd/algorithm.py:57-68 -> nd/utils.py:343-401)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    # the ranks' stdout goes through a file, not a pipe: a pipe stays open for as long as any
    # descendant of the launcher holds it
    import tempfile
    with tempfile.TemporaryFile('w+') as fo:
        rc = subprocess.call(cmd, env=env, stdout=fo, stdin=subprocess.DEVNULL, cwd=os.getcwd())
        fo.seek(0)
        lines = [ln for ln in fo.read().splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith('{')]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln, file=sys.stderr)              # transports' banners: not this program's output
    if rc != 0:
        raise SystemExit(rc if 0 < rc < 256 else 1)
    if not js:
        print('bench.py: the %d ranks printed no result line' % args.gpus, file=sys.stderr)
        raise SystemExit(1)
    print(js[-1])
    sys.stdout.flush()
    raise SystemExit(0)


def first_contact(w, dist, rdev):
    """N > 1, before anything is timed: the filter's two forms on a shard with neighbours -- halo
    exchange first, then one launch (sequential) against interior rows while the halo travels
    (overlapped) -- must give the same filtered values bit for bit on EVERY rank (the reference's
    split -> map -> merge equals the unsplit result, nd/utils.py:288-340,
    nd/tests/test_filters_common.py:54-60).  The overlapped form had only ever run over gloo when this
    was written; under RCCL the transfers run on the communicator's stream.  -> dict for `comm`;
    on a difference the workload is switched to the sequential form for the timed region."""
    import torch
    if not hasattr(w, 'filter_once'):
        return None
    seq = w.filter_once(overlap=False).clone()
    torch.cuda.synchronize()
    ovl = w.filter_once(overlap=True)
    torch.cuda.synchronize()
    if os.environ.get('ND_AMD_BENCH_FORCE_OVERLAP_DIFF') == str(w.rank):
        # test hook (tests/test_multigpu_gpu.py): pretend this rank's overlapped launch read a halo row
        # before it had landed
        ovl = ovl.clone()
        ovl[0, 0, 0, :8] += 1.0
    ndiff = int((seq.view(torch.int32) != ovl.view(torch.int32)).sum().item())
    del seq, ovl
    worst = torch.tensor([ndiff], dtype=torch.int64, device=rdev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    same = int(worst.item()) == 0
    res = {'overlap_equals_sequential': same, 'timed_form': 'overlapped' if same else 'sequential'}
    if not same:
        res['values_differing_max_over_ranks'] = int(worst.item())
        res['values_differing_this_rank'] = ndiff
        w.overlap = False
    return res


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        launch_ranks(args, sys.argv[1:])           # does not return
    # first contact with N > 1 must not be taken on trust: as many ranks as asked for, or no number
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE is %d: refusing to run (and to report) another '
                         'job size than the one asked for' % (args.gpus, world))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU (nd_amd has no CPU path)')
    # ND_AMD_BENCH_REHEARSE=gloo: the N > 1 code path on a box with fewer GPUs than ranks (ranks
    # share the devices, gloo instead of RCCL, which refuses two ranks on one device).  A rehearsal
    # of the sharding / reduction logic (tests/test_multigpu_gpu.py), never a measurement.
    rehearse = os.environ.get('ND_AMD_BENCH_REHEARSE', '') == 'gloo'
    if rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or 'RANK' in os.environ:      # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RCCL's version banner goes to stdout at NCCL_DEBUG=VERSION/INFO; this program's stdout is
        # the one JSON line
        if os.environ.get('NCCL_DEBUG', '').upper() in ('VERSION', 'INFO', 'TRACE'):
            os.environ['NCCL_DEBUG'] = 'WARN'
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    rdev = torch.device('cpu') if rehearse else dev      # where the two scalar reductions live

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    w = WORKLOADS[args.workload](args, rank, world, dev)
    torch.cuda.synchronize()
    if args.traffic_run is not None:
        # the command the rocprofv3 --pmc passes of tools/collect_traffic.sh are pointed at
        global TRAFFIC_MODE
        TRAFFIC_MODE = True
        if args.traffic_run == 'headline':
            timed(w.step, 3, 0, barrier)
        else:
            extras(w, barrier, dev, only=args.traffic_run)
        print(json.dumps({'traffic_run': args.traffic_run}))
        return
    contact = None
    if dist is not None and world > 1:
        if not rehearse:
            devs = [None] * world
            dist.all_gather_object(devs, local_rank)
            if len(set(devs)) != world:
                raise RuntimeError('ranks share devices: %r' % devs)
        contact = first_contact(w, dist, rdev)
    # The timed region: every step is the whole call; inside it only the dominant kernel carries
    # the library's event pair (each pair costs a few microseconds of stream time, and the
    # roofline needs that kernel's duration measured live here), plus one torch event per step
    # boundary for the spread.  The other kernels' durations come from a second, untimed loop.
    per_step = []
    state_before = device_state(local_rank) if args.extras else None
    dt, avg, out = timed(w.step, args.steps, args.warmup, barrier, only=[w.dom], per_step=per_step, settle=False)
    state_after = device_state(local_rank) if args.extras else None
    _, avg_all, _ = timed(w.step, max(3, min(args.steps, 10)), 1, barrier, settle=False)
    avg_timed = dict(avg)                      # measured inside the timed region
    for name, ms in avg_all.items():
        avg.setdefault(name, ms)

    local_dt = dt
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([w.npix], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_px = float(tot.item())
    else:
        total_px = float(w.npix)
    value = total_px * args.steps / dt / 1e6
    flagged = float((out.sum(dim=2) > 0).float().mean().item())
    comm = comm_detail = None
    if dist is not None:
        # who took part, over what, and what travelled: gathered from every rank so that the line
        # carries its own evidence of the N > 1 run
        mine = {'rank': rank, 'device': torch.cuda.get_device_name(local_rank), 'device_index': local_rank,
                'rows': [w.r0, w.r1], 'step_ms': local_dt / args.steps * 1e3,
                'step_ms_median': sorted(per_step)[len(per_step) // 2] if per_step else None}
        if hasattr(w, 'comm'):
            mine.update(w.comm())
            if hasattr(w, 'status_any'):
                from nd_amd import kernels
                kernels.raise_if_no_solution(w.status_any)
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        bcheck = boundary_check(w, out, dist, rank, world, rdev)
        comm_detail = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                       'data_path_collective': 'none (per-pixel path: every rank runs its own rows)'
                       if w.name != 'pipeline' else
                       'one point-to-point halo exchange per step (batch_isend_irecv with the row neighbours, '
                       'overlapped with the filter on the rows that need no halo)',
                       'ranks': ranks, 'boundary_check': bcheck, 'first_contact': contact}
        if rank == 0:
            # the line's form: one short row per rank
            comm = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                    'collective': 'none' if w.name != 'pipeline' else 'p2p halo exchange per step',
                    'rank_cols': ['rank', 'device_index', 'row0', 'row1', 'step_ms'],
                    'ranks': [[r_['rank'], r_['device_index'], r_['rows'][0], r_['rows'][1], r_['step_ms']]
                              for r_ in ranks],
                    'device': ranks[0]['device'],
                    'boundary_check': {k_: bcheck[k_] for k_ in ('boundaries', 'map_bytes_compared',
                                                                 'map_bytes_differing', 'filtered_values_differing')}}
            if 'halo_bytes_sent_per_step' in ranks[0]:
                comm['halo_bytes_sent_per_step'] = [r_['halo_bytes_sent_per_step'] for r_ in ranks]
                comm['exchange_ms_alone'] = [r_['exchange_ms_alone'] for r_ in ranks]
            if contact is not None:
                comm.update(contact)

    if rank == 0:
        is_default = (w.name == 'omnibus' and (w.k, args.ny, w.nx, args.alpha, args.looks, args.change_frac)
                      == (24, 4096, 4096, 0.99, 9, 0.01))
        traffic, source = (profiled_traffic('headline', 'omnibus_c2_retain')
                           if world == 1 and is_default else (None, None))
        if args.traffic_bytes is not None:
            traffic, source = args.traffic_bytes, '--traffic-bytes'
        # the kernel the roofline is quoted on: the workload's pass A, or (low thresholds, where the
        # search is fused into the one streaming kernel) whichever kernel takes the time
        dom_k = w.dom if w.dom in avg else max(avg, key=avg.get)
        ps = sorted(per_step)
        if w.name == 'omnibus':
            # SURVEY 8(d) / BASELINE.md basis: the planes read once, k * 4 * sizeof(T) bytes per pixel
            # (384 B at k = 24 float32); the same kernel also zero-fills the change map (k bytes per
            # pixel more, 408 B in all): that rate is the secondary figure.
            roof = roofline(dom_k, avg[dom_k], w.read_bytes, traffic=traffic, traffic_source=source,
                            note='algorithmic bytes = the four planes read once (%d B per pixel); '
                                 'achieved_read_write adds the change map the same kernel '
                                 'zero-fills (%d B per pixel in all)'
                                 % (w.read_bytes // w.npix, w.alg_bytes // w.npix))
            roof['achieved_read_write'] = w.alg_bytes / (avg[dom_k] * 1e-3) / 1e9
            roof['frac_read_write'] = roof['achieved_read_write'] / HBM_PEAK_GBS
        else:
            roof = roofline(dom_k, avg[dom_k], w.alg_bytes, traffic=traffic, traffic_source=source,
                            note='algorithmic bytes = planes read once + change map written once'
                            if w.name != 'pipeline' else 'algorithmic bytes = filter input + output')
            if w.name == 'pipeline':
                roof = valu_roofline(roof, w.npix * w.k * 4 * (3 * 7 * 7 - 1),
                                     '146 dependent float32 additions per output, packed over two outputs')
                roof['hbm'] = {'achieved': roof['hbm']['achieved'], 'frac': roof['hbm']['frac']}
        m = {
            'metric': w.metric(), 'value': value, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'step_ms': {'min': ps[0], 'median': ps[len(ps) // 2], 'max': ps[-1]} if ps else None,
            'scaling': args.scaling,
            'data': 'synthetic' if not rehearse else
            'synthetic; REHEARSAL: ranks share devices over gloo, not a measurement',
            'config': dict(w.config(), flagged_pixel_fraction=flagged,
                           rows_per_rank=w.rows, sharding='row blocks (tiles.row_partition), %s'
                           % ('no collective' if w.name != 'pipeline' or world == 1 else
                              'p2p halo exchange per step')),
            'roofline': roof, 'kernels_ms': avg, 'comm': comm,
        }
        detail = {
            'describe': w.describe(),
            'arithmetic': 'float32 planes and running sums; float64 product of determinants, logs and '
                          'chi-square pair (the reference\'s rounding points)',
            'roofline': roof, 'comm': comm_detail,
            'kernels_ms_timed_region': avg_timed,
            'kernels_ms_second_loop': {n_: m_ for n_, m_ in avg_all.items() if n_ not in avg_timed},
            'kernels_ms_note': 'kernels_ms_timed_region: HIP events inside the timed region (the dominant '
                               'kernel only: an event pair costs stream time); kernels_ms_second_loop: the '
                               'same step run again behind it; kernels_ms (the line): both together',
            'step_ms_all': per_step,
        }
        if world == 1 and w.name == 'omnibus':
            if args.cpu_rows > 0:
                cb = cpu_baseline_omnibus(w, out)
                detail['cpu_baseline'] = cb
                m['cpu_baseline'] = dict(cb, sample=cb['sample_short'],
                                         one_thread_value=cb['one_thread']['value'])
                if not cb['gpu_matches_cpu_on_sample']:
                    raise RuntimeError('the GPU change map differs from the oracle on the CPU baseline\'s rows')
        elif world == 1:
            chk = w.check(out)
            detail['check'] = chk
            m['matches_oracle_on_sample'] = bool(chk.get('bad', 1) == 0 and chk.get('change_bad', 0) == 0)
        if world == 1 and is_default and not args.no_secondary and not args.extras:
            # configs 3 - 5 in the driver-run line (behind every timed region of the headline)
            m['secondary'], detail['secondary'] = secondary(w, barrier, dev)
        if args.extras and world == 1 and w.name == 'omnibus':
            # the secondary workloads: their own invocation (`bench.py --extras`), their own file
            detail['device_state'] = {'before_timed_region': state_before, 'after_timed_region': state_after}
            detail['transfer'] = transfer_rates(dev)
            ex = extras(w, barrier, dev)
            m['extras_file'] = _write_sidecar('bench_extras.json', {'headline': headline(m), 'extra': ex})
        detail['line'] = headline(m)
        m['detail_file'] = _write_sidecar('bench_detail.json', detail)
        print(emit(headline(m)))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
