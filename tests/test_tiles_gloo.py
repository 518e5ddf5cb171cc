"""Multi-rank row sharding on CPU: world_size-2, -3 and -4 gloo groups check the halo exchange and
that exchange -> filter tile+halo -> trim reproduces the unsharded filter bit for bit (the role of
nd/tests/test_filters_common.py:54-60 and test_tiling.py:117-127 for the GPU tile layer).
The filters here are the CPU oracle's boxcar (scipy.ndimage.convolve arithmetic) and non-local
means (nd/_filters.pyx arithmetic): the real arithmetic of the path, so a wrong halo width, a wrong
trim or a reflection at an interior edge shows up as a value difference."""
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


def _oracle_boxcar(t, w):
    """BoxcarFilter(dims=('y','x'), w) on a (..., y, x) CPU tensor through the oracle."""
    from oracle import oracle as O
    a = np.ascontiguousarray(t.numpy())
    k = np.ones((1,) * (a.ndim - 2) + (w, w)) / float(w * w)
    return torch.from_numpy(O.convolve(a, k))


def _oracle_nlmeans(t, r, f, n_eff=-1):
    """non-local means over (y, x) of a planar (var, time, y, x) CPU tensor through the oracle
    (joint weights over the variables, every date on its own), patch_mode 1."""
    from oracle import oracle as O
    a = np.ascontiguousarray(t.permute(2, 3, 1, 0).numpy())          # (y, x, time, var)
    out = np.empty_like(a)
    O.pixelwise_nlmeans_3d(a, out, (r, r, 0), (f, f, 0), 0.5, 0.6, n_eff, patch_mode=1)
    return torch.from_numpy(out).permute(3, 2, 0, 1).contiguous()


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _worker(rank, world, port, ny, w, ret):
    _init(rank, world, port)
    try:
        from nd_amd import tiles
        g = torch.Generator().manual_seed(0)
        full = torch.rand((2, 3, ny, 9), generator=g, dtype=torch.float64) + 0.2
        r0, r1 = tiles.my_rows(ny)
        halo = w // 2
        # (a) shard allocated with margins: the exchange receives into them, the block stays put
        sh = tiles.empty_shard((2, 3), ny, 9, halo, 'cpu', torch.float64)
        assert (sh.r0, sh.r1) == (r0, r1)
        sh.ext.fill_(float('nan'))
        sh.core.copy_(full[:, :, r0:r1])
        core_ptr = sh.core.data_ptr()
        tiles.exchange_halo_(sh)
        assert sh.core.data_ptr() == core_ptr
        assert torch.equal(sh.ext, full[:, :, r0 - sh.lo:r1 + sh.hi])
        assert sh.lo == (halo if rank > 0 else 0) and sh.hi == (halo if rank < world - 1 else 0)
        got = tiles.filter_rows(lambda t: _oracle_boxcar(t, w), sh, halo, 2)
        want = _oracle_boxcar(full, w)[:, :, r0:r1]
        assert torch.equal(got, want), float((got - want).abs().max())
        # (b) convenience form on a plain block
        ext, lo, hi = tiles.exchange_halo(full[:, :, r0:r1].contiguous(), halo, 2, ny)
        assert torch.equal(ext, full[:, :, r0 - lo:r1 + hi])
        got = tiles.filter_rows(lambda t: _oracle_boxcar(t, w), full[:, :, r0:r1].contiguous(), halo, 2, ny)
        assert torch.equal(got, want)
        # (c) the scatter form equals what the exchange produced
        sc = tiles.shard_of(full, halo, 2, rank, world)
        assert torch.equal(sc.ext, sh.ext) and (sc.lo, sc.hi) == (sh.lo, sh.hi)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,ny,w', [(2, 20, 3), (2, 21, 5), (3, 20, 3), (4, 23, 5)])
def test_row_sharded_boxcar_equals_unsharded(world, ny, w):
    _spawn(_worker, world, (ny, w))


def _nlm_worker(rank, world, port, ny, r, f, n_eff, ret):
    _init(rank, world, port)
    try:
        from nd_amd import tiles
        g = torch.Generator().manual_seed(1)
        full = (torch.rand((2, 2, ny, 11), generator=g, dtype=torch.float32) + 0.5)
        halo = r + f
        sh = tiles.empty_shard((2, 2), ny, 11, halo, 'cpu', torch.float32)
        sh.core.copy_(full[:, :, sh.r0:sh.r1])
        got = tiles.filter_rows(lambda t: _oracle_nlmeans(t, r, f, n_eff), sh, halo, 2)
        want = _oracle_nlmeans(full, r, f, n_eff)[:, :, sh.r0:sh.r1]
        assert torch.equal(got, want), float((got - want).abs().max())
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,ny,r,f,n_eff', [(2, 18, 2, 1, -1), (3, 24, 3, 1, 5.0)])
def test_row_sharded_nlmeans_equals_unsharded(world, ny, r, f, n_eff):
    _spawn(_nlm_worker, world, (ny, r, f, n_eff))


def _spawn(fn, world, args):
    port = _free_port()
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=fn, args=(r, world, port) + tuple(args) + (ret,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(ret.keys()) == list(range(world))
    return ret


def test_row_partition_matches_reference_split():
    """tiles.row_partition == xr_split without buffer (nd/utils.py:305-310)."""
    from nd_amd import _adapter, tiles
    for n, parts in [(20, 2), (21, 4), (4096, 8), (5, 8), (16384, 8)]:
        assert tiles.row_partition(n, parts) == [(min(a, n), b) for a, b in _adapter.split_bounds(n, parts, 0)]
        rows = tiles.row_partition(n, parts)
        assert rows[0][0] == 0 and rows[-1][1] == n
        assert all(rows[i][1] == rows[i + 1][0] for i in range(parts - 1))


def test_partition_check_is_rank_independent():
    """n = 9 rows over 4 ranks gives blocks 3, 3, 3, 0: every rank must refuse, not only the
    neighbours of the empty block (a rank that raised alone would leave the others waiting in
    their receives)."""
    from nd_amd import tiles
    with pytest.raises(ValueError):
        tiles.check_partition(9, 4, 1)
    with pytest.raises(ValueError):
        tiles.check_partition(20, 2, 11)
    tiles.check_partition(20, 2, 10)
    tiles.check_partition(9, 1, 100)          # a single rank never exchanges
    tiles.check_partition(9, 4, 0)


def _small_worker(rank, world, port, ny, halo, ret):
    _init(rank, world, port)
    try:
        from nd_amd import tiles
        try:
            r0, r1 = tiles.my_rows(ny)
            tiles.exchange_halo(torch.zeros((1, 1, r1 - r0, 4)), halo, 2, ny)
            ret[rank] = 'ok'
        except ValueError:
            ret[rank] = 'ValueError'
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,ny,halo', [(2, 4, 5), (4, 9, 1)])
def test_too_small_tiles_are_refused_by_every_rank(world, ny, halo):
    ret = _spawn(_small_worker, world, (ny, halo))
    assert all(ret[r] == 'ValueError' for r in range(world)), dict(ret)
