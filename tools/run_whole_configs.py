"""BASELINE configs[3] and configs[4] WHOLE on one MI355X (288 GB), once (not part of
pytest -m gpu: 116 GB and 213 GB of device memory):

    python tools/run_whole_configs.py c3          # OmnibusTest full-pol C3, 48t x 8192 x 8192: ONE call
    python tools/run_whole_configs.py pipeline    # NLMeans -> OmnibusTest, 24t x 16384 x 16384 x 4 variables

Both rasters are synthesised per row block (the seeds of bench.py's per-rank shares, seed + block) into one
resident stack.  c3: tiles.omnibus_c3_rows on the whole stack, one launch of each kernel over 67 M pixels.
pipeline: the input (103 GB), the filtered stack (103 GB) and the map (6.4 GB) all resident; the filter and
the test walk the raster in the eight row blocks of the reference's split (tiles.row_partition =
nd/utils.py:305-310), each block a VIEW of the resident stack with its neighbours' rows as halo
(tiles.shard_of: nothing is copied or exchanged) -- one launch over 25.8 G values was not attempted: the
tiled filter kernels have never run beyond 2^31 elements per launch, and a fault on this pool resets the node.
Checks: oracle.checks.omnibus_sample (sampled pixels, first and last rows whole) /
oracle.checks.nlmeans_crops (the four corners, crops across every block boundary, the interior)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ev_ms(fn, torch):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), out


def run_c3():
    import torch
    from nd_amd import synth, tiles, _lib
    from oracle import checks
    dev = torch.device('cuda:0')
    k, ny, nx, looks, alpha, blocks = 48, 8192, 8192, 9, 0.99, 8
    stack = synth.empty_stack(9, k, ny, nx, dev)
    t0 = time.perf_counter()
    for b, (r0, r1) in enumerate(tiles.row_partition(ny, blocks)):
        part = synth.wishart_c3_stack(k, r1 - r0, nx, looks=looks, seed=4321 + b, device=dev, change_frac=0.01)
        stack[:, :, r0:r1, :] = part
        del part
        print('synthesised rows %d..%d (%.0f s)' % (r0, r1, time.perf_counter() - t0), flush=True)
    torch.cuda.synchronize()
    step = lambda: tiles.omnibus_c3_rows(stack, alpha, looks)      # noqa: E731
    ms0, out = ev_ms(step, torch)                                  # (first call: workspace allocation)
    del out
    _lib.timing_enable(64)
    times = []
    for _ in range(3):
        ms, out = ev_ms(step, torch)
        times.append(ms)
    by = {}
    for n_, ms in _lib.timing_collect():
        by.setdefault(n_, []).append(ms)
    share_ms = None
    shr = stack[:, :, :1024, :]
    for _ in range(2):
        share_ms, o2 = ev_ms(lambda: tiles.omnibus_c3_rows(shr, alpha, looks), torch)
    same_share = bool(torch.equal(o2, out[:1024]))
    res = checks.omnibus_sample(stack, out, alpha, looks, nsample=20000, rows=(0, ny - 1), seed=4, pol=3)
    nbytes = stack.shape[0] * k * ny * nx * 4
    print(json.dumps({'config': 'BASELINE configs[3] whole: OmnibusTest full-pol C3 %dt x %d x %d f32 x 9 planes, one GPU, one call'
                                % (k, ny, nx), 'pixels': ny * nx, 'input_GB': round(nbytes / 1e9, 1),
                      'ms_per_step': [round(t, 3) for t in times], 'first_call_ms': round(ms0, 1),
                      'kernels_ms': {n_: round(sum(v) / len(v), 3) for n_, v in by.items()},
                      'hbm_frac_whole_step': round((nbytes + ny * nx * k) / (min(times) * 1e-3) / 8e12, 3),
                      'one_share_ms (first 1024 rows)': round(share_ms, 3), 'eight_shares_ms': round(8 * share_ms, 3),
                      'share_map_equals_rows_of_whole_map': same_share,
                      'changes': int(out.sum().item()), 'check': res}), flush=True)
    return 0 if res['bad'] == 0 and same_share else 1


def run_pipeline():
    import torch
    from nd_amd import synth, tiles, kernels
    from oracle import checks
    import bench
    T = bench.TUT
    dev = torch.device('cuda:0')
    k, ny, nx, looks, blocks = 24, 16384, 16384, 9, 8
    halo = T['r'][1] + T['f'][1]
    stack = synth.empty_stack(4, k, ny, nx, dev)
    t0 = time.perf_counter()
    parts = tiles.row_partition(ny, blocks)
    for b, (r0, r1) in enumerate(parts):
        synth.wishart_c2_stack(k, r1 - r0, nx, looks=looks, seed=99 + b, change_frac=0.01, out=stack[:, :, r0:r1, :])
        print('synthesised rows %d..%d (%.0f s)' % (r0, r1, time.perf_counter() - t0), flush=True)
    filt = synth.empty_stack(4, k, ny, nx, dev)
    change = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    any_status = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        for b, (r0, r1) in enumerate(parts):
            shard = tiles.shard_of(stack, halo, 2, b, blocks)          # a view: the neighbours' rows are the halo
            f_b = tiles.nlmeans_rows(shard, ny, T['r'], T['f'], T['sigma'], T['h'], n_eff=T['n_eff'], patch_mode=0,
                                     status=status)
            any_status.bitwise_or_(status)
            filt[:, :, r0:r1, :] = f_b
            change[r0:r1] = tiles.omnibus_rows(f_b, T['alpha'], T['n'])
            del f_b
        return change

    ms0, _ = ev_ms(step, torch)
    times = [ev_ms(step, torch)[0] for _ in range(2)]
    kernels.raise_if_no_solution(any_status)
    crops = [(0, 0), (0, nx), (ny, 0), (ny, nx), (ny // 2 + 77, nx // 3)]
    crops += [(r1 - 4, 5000 + 300 * b) for b, (r0, r1) in enumerate(parts[:-1])]       # across every block boundary
    res = checks.nlmeans_crops(stack, filt, T['r'], T['f'], T['sigma'], T['h'], T['n_eff'], 0, crops, size=(8, 128),
                               then_omnibus=(T['alpha'], T['n']), change=change)
    nbytes = 4 * k * ny * nx * 4
    print(json.dumps({'config': 'BASELINE configs[4] whole: NLMeansFilter(r=(1,3,3), f=1, n_eff=50) -> OmnibusTest(n=50, alpha=%g) on '
                                '%dt x %d x %d f32 x 4 variables, one GPU, the raster walked in the 8 row blocks of the '
                                'reference\'s split as views of the resident stack' % (T['alpha'], k, ny, nx),
                      'pixels': ny * nx, 'input_GB': round(nbytes / 1e9, 1), 'filtered_GB': round(nbytes / 1e9, 1),
                      'map_GB': round(ny * nx * k / 1e9, 1), 'device_memory_GB_peak': round(torch.cuda.max_memory_allocated() / 1e9, 1),
                      'ms_per_step (copies of the blocks into the resident outputs included)': [round(t, 2) for t in times],
                      'first_step_ms': round(ms0, 1), 'changes': int(change.sum().item()),
                      'crops': len(crops), 'check': res}), flush=True)
    return 0 if res['bad'] == 0 and res.get('change_bad', 0) == 0 else 1


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    sys.exit(run_c3() if what == 'c3' else run_pipeline())
