"""bench.py's host logic, no GPU: the ONE line (headline object only, strict JSON, under 4 KB -- a
longer line is cut by the driver's stdout tail and leaves the round unmeasured), the
--gpus / WORLD_SIZE agreement, and the self-launch of the N ranks (a child process, relayed exit
code)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stub(world):
    import bench
    roof = bench.roofline('omnibus_c2_global', 1.1703619420528413, 6442450944,
                          traffic=6976265109.333333, traffic_source='profiles/r05_traffic.json@0123abc',
                          note='x' * 300)
    roof['achieved_read_write'] = 5848.707038434223
    roof['frac_read_write'] = 0.7310883798042779
    m = {
        'metric': 'Mpixels/s OmnibusTest dual-pol 24t x 4096 x 4096', 'value': 13251.52934073746,
        'n_gpus': world, 'steps': 20, 'warmup': 5, 'ms_per_step': 1.2660588501603343,
        'step_ms': {'min': 1.2521729469299316, 'median': 1.2608519792556763, 'max': 1.3189729452133179},
        'scaling': 'weak', 'data': 'synthetic',
        'config': {'workload': 'OmnibusTest C2 24t x 4096 x 4096 f32 per GPU (BASELINE configs[1]), resident in HBM',
                   'looks': 9, 'alpha': 0.99, 'change_frac': 0.01, 'flagged_pixel_fraction': 0.01918739080429077,
                   'rows_per_rank': 4096, 'sharding': 'row blocks (tiles.row_partition), no collective'},
        'roofline': roof,
        'kernels_ms': {'omnibus_c2_global': 1.1703619420528413, 'omnibus_c2_dense': 0.006888000015169382,
                       'omnibus_c2_search': 0.05704450011253357, 'omnibus_c2_exact': 0.029668400064110756},
        'detail_file': 'bench_detail.json', 'extras_file': 'bench_extras.json',
    }
    if world == 1:
        m['cpu_baseline'] = {'value': 24.18750614527941, 'unit': 'Mpixels/s', 'cores': 16, 'kind': 'port',
                             'sample': 'oracle/nd_oracle.c (OpenMP) on the first 4096 rows of the same stack, '
                                       '0.69 s; one_thread_value: 1024 rows, 1.95 s',
                             'one_thread_value': 2.1496663155674476, 'gpu_matches_cpu_on_sample': True,
                             'one_thread': {'long': 'y' * 500}}
    else:
        m['comm'] = {'backend': 'nccl', 'world_size': world, 'collective': 'p2p halo exchange per step',
                     'rank_cols': ['rank', 'device_index', 'row0', 'row1', 'step_ms'],
                     'ranks': [[r, r, 2048 * r, 2048 * (r + 1), 14.626123456789] for r in range(world)],
                     'device': 'AMD Instinct MI355X',
                     'boundary_check': {'boundaries': world - 1, 'map_bytes_compared': 3145728 * (world - 1),
                                        'map_bytes_differing': 0, 'filtered_values_differing': 0},
                     'halo_bytes_sent_per_step': [25165824] * world,
                     'exchange_ms_alone': [0.312345678] * world,
                     'overlap_equals_sequential': True, 'timed_form': 'overlapped'}
    return m


def test_secondary_block_fits_the_drivers_tail():
    """The default N = 1 run adds `secondary` (BASELINE configs 3 - 5 at one GPU's share): at most 500 bytes,
    the whole line under 1 900 so that the driver's 2 000-character tail keeps it whole; strict JSON."""
    import bench
    m = _stub(1)
    m['secondary'] = {'nlm_cc_pm0': [2.112345, 0.4412345, 'valu', True], 'nlm_cc_pm1': [41.81234, 0.404123, 'valu', True],
                      'c3_share_a0.99': [2.851234, 0.661234, 'hbm', True], 'pipeline_share': [14.85912, 0.494123, 'valu', True],
                      'c2_a0.01': [1.395912, 0.612345, 'hbm', True], 'c2_yxt_a0.01': [1.612123, 0.530123, 'hbm', True]}
    text = bench.emit(bench.headline(m))
    assert len(text.encode()) < 1900, len(text)
    line = json.loads(text)
    assert set(line['secondary']) == {k for k, _ in bench.SECONDARY}
    assert len(json.dumps(line['secondary'], separators=(',', ':'))) < 500
    for ms, frac, bound, ok in line['secondary'].values():
        assert ms > 0 and 0 < frac < 1 and bound in ('hbm', 'valu') and ok is True
    m['secondary']['x' * 300] = [1.0, 0.5, 'hbm', True]
    m['secondary']['y' * 300] = [1.0, 0.5, 'hbm', True]
    with pytest.raises(RuntimeError, match='secondary'):
        bench.emit(bench.headline(m))


@pytest.mark.parametrize('world', [1, 2, 8])
def test_line_is_the_headline_object_only_and_short(world):
    import bench
    text = bench.emit(bench.headline(_stub(world)))
    assert '\n' not in text and len(text.encode()) < 4096, len(text)
    line = json.loads(text)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'step_ms', 'dtype',
                'config', 'roofline', 'kernels_ms', 'higher_is_better', 'scaling', 'vs_baseline', 'data'):
        assert key in line, key
    assert 'extra' not in line and 'device_state' not in line and 'transfer' not in line
    assert line['n_gpus'] == world and line['vs_baseline'] is None and line['dtype'] == 'f32'
    roof = line['roofline']
    assert set(roof) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} and 'note' not in roof
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-4
    assert 'workload' in line['config'] and 'model' not in line['config']
    if world == 1:
        assert set(line['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
        assert 'one_thread' not in line['cpu_baseline']          # the long form stays in bench_detail.json
    else:
        assert len(line['comm']['ranks']) == world


def test_emit_refuses_what_the_driver_could_not_parse():
    import bench
    m = _stub(1)
    m['value'] = float('nan')
    with pytest.raises(ValueError):
        bench.emit(bench.headline(m))
    m = _stub(1)
    m['config']['workload'] = 'w' * 5000
    with pytest.raises(RuntimeError, match='bytes'):
        bench.emit(bench.headline(m))


def _run(args, env=None, timeout=120):
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout, stdin=subprocess.DEVNULL)


def test_world_size_must_equal_gpus():
    """started under a launcher with another rank count than --gpus: an error, before any GPU call"""
    p = _run(['--gpus', '4'], env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert p.returncode != 0 and '--gpus 4 but WORLD_SIZE is 2' in p.stderr
    p = _run(['--gpus', '1'], env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert p.returncode != 0 and 'WORLD_SIZE is 2' in p.stderr


def test_gpus_n_starts_n_ranks_or_fails():
    """`python bench.py --gpus 2` alone: the parent starts the two ranks as a child process (the
    driver's launch line) and relays their exit code.  Here there is no GPU: both ranks stop at
    'needs a ROCm GPU', the launcher fails, and so must the parent -- it must never go on
    single-rank and print n_gpus: 1."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('the CPU form of this test: on a GPU box tests/test_multigpu_gpu.py covers the success path')
    p = _run(['--gpus', '2', '--steps', '1', '--warmup', '0', '--ny', '32', '--nx', '64', '--k', '4'], timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ''                       # no result line of a job that did not run
    assert 'needs a ROCm GPU' in p.stderr
