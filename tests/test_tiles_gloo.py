"""Multi-rank row sharding on CPU: world_size-2 and -3 gloo groups check the halo exchange and
that exchange -> filter tile+halo -> trim reproduces the unsharded filter (the role of
nd/tests/test_filters_common.py:54-60 and test_tiling.py:117-127 for the GPU tile layer).
The filter here is a plain torch box mean: the point is the tile logic, not the kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _box_rows(t, w):
    """box mean of width w along axis -2 with half-sample reflection (scipy 'reflect')."""
    h = w // 2
    idx = torch.arange(-h, t.shape[-2] + h)
    n = t.shape[-2]
    idx = torch.where(idx < 0, -idx - 1, idx)
    idx = torch.where(idx >= n, 2 * n - 1 - idx, idx)
    p = t.index_select(-2, idx)
    return sum(p[..., i:i + n, :] for i in range(w)) / w


def _worker(rank, world, port, ny, w, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nd_amd import tiles
        g = torch.Generator().manual_seed(0)
        full = torch.randn((2, 3, ny, 7), generator=g, dtype=torch.float64)
        r0, r1 = tiles.my_rows(ny)
        core = full[:, :, r0:r1].contiguous()
        halo = w // 2
        ext, lo, hi = tiles.exchange_halo(core, halo, 2)
        # the extended tile is exactly the corresponding rows of the full raster
        assert torch.equal(ext, full[:, :, r0 - lo:r1 + hi])
        assert lo == (halo if rank > 0 else 0) and hi == (halo if rank < world - 1 else 0)
        got = tiles.filter_rows(lambda t: _box_rows(t, w), core, halo, 2)
        want = _box_rows(full, w)[:, :, r0:r1]
        assert torch.allclose(got, want, rtol=0, atol=1e-14), float((got - want).abs().max())
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,ny,w', [(2, 20, 3), (2, 21, 5), (3, 20, 3)])
def test_row_sharded_filter_equals_unsharded(world, ny, w):
    port = _free_port()
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ny, w, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(ret.keys()) == list(range(world))


def test_row_partition_matches_reference_split():
    """tiles.row_partition == xr_split without buffer (nd/utils.py:305-310)."""
    from nd_amd import _adapter, tiles
    for n, parts in [(20, 2), (21, 4), (4096, 8), (5, 8), (16384, 8)]:
        assert tiles.row_partition(n, parts) == [(min(a, n), b) for a, b in _adapter.split_bounds(n, parts, 0)]
        rows = tiles.row_partition(n, parts)
        assert rows[0][0] == 0 and rows[-1][1] == n
        assert all(rows[i][1] == rows[i + 1][0] for i in range(parts - 1))


def test_too_small_tiles_are_refused():
    port = _free_port()
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_small_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    assert ret[0] == 'ValueError' and ret[1] == 'ValueError'


def _small_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nd_amd import tiles
        core = torch.zeros((1, 1, 2, 4))
        try:
            tiles.exchange_halo(core, 5, 2)
            ret[rank] = 'ok'
        except ValueError:
            ret[rank] = 'ValueError'
    finally:
        dist.destroy_process_group()
