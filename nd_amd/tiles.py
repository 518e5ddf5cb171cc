"""
nd_amd/tiles.py -- the multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), the (y, x) raster split into contiguous row blocks, one per rank.

This plays the role of the reference's only multi-worker mechanism, `utils.parallel` =
`xr_split(buffer)` -> process pool -> `xr_merge` (nd/utils.py:288-401, driven by
nd/algorithm.py:57-68), with the same arithmetic: rank i owns rows
[i*cs, min((i+1)*cs, n)), cs = ceil(n / world), and a windowed filter needs `halo = _buffer(dim)`
extra rows from each neighbour (kernel//2 for convolution, r+f for non-local means).

  OmnibusTest       per pixel, no exchange at all: every rank runs its rows.
  windowed filters  ONE neighbour exchange of `halo` rows per direction (point-to-point
                    send/recv pairs batched in one group -- each pair rides its own xGMI link),
                    then the kernel runs on tile+halo; rows at the GLOBAL top/bottom use the
                    kernel's own reflection.  No all-reduce / all-gather is involved.
"""
import math

import torch
import torch.distributed as dist


def row_partition(n, parts):
    """[(r0, r1)] per part, xr_split's chunking without the buffer (nd/utils.py:305-310)."""
    cs = int(math.ceil(n / parts))
    return [(min(i * cs, n), min((i + 1) * cs, n)) for i in range(parts)]


def my_rows(n, group=None):
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    return row_partition(n, world)[rank]


def exchange_halo(core, halo, dim, group=None):
    """Extend this rank's row block by up to `halo` rows of each neighbour.

    core : tensor whose axis `dim` holds this rank's rows (ranks are in row order).
    Returns (ext, lo, hi): `ext` = [rows from rank-1 | core | rows from rank+1] along `dim`,
    `lo`/`hi` = how many rows were added in front / behind (0 at the global edges).
    """
    if halo <= 0 or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return core, 0, 0
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = core.shape[dim]
    # every rank must be able to serve a full halo (the reference's chunks overlap by `buffer`
    # in the same way); sizes are agreed on with one small all-gather of row counts
    counts = [torch.zeros(1, dtype=torch.int64, device=core.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n_local], dtype=torch.int64, device=core.device),
                    group=group)
    counts = [int(c.item()) for c in counts]
    lo = min(halo, counts[rank - 1]) if rank > 0 else 0
    hi = min(halo, counts[rank + 1]) if rank < world - 1 else 0
    send_up = min(halo, n_local) if rank > 0 else 0            # my first rows go to rank-1
    send_dn = min(halo, n_local) if rank < world - 1 else 0    # my last rows go to rank+1
    if (rank > 0 and counts[rank - 1] < halo) or (rank < world - 1 and counts[rank + 1] < halo) \
            or (n_local < halo and world > 1):
        raise ValueError('row blocks (%s rows) are smaller than the halo (%d): use fewer ranks'
                         % (counts, halo))

    def rows(t, a, b):
        idx = [slice(None)] * t.dim()
        idx[dim] = slice(a, b)
        return t[tuple(idx)]

    ops = []
    recv_lo = recv_hi = None
    if rank > 0:
        shape = list(core.shape)
        shape[dim] = lo
        recv_lo = torch.empty(shape, dtype=core.dtype, device=core.device)
        ops.append(dist.P2POp(dist.isend, rows(core, 0, send_up).contiguous(), rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_lo, rank - 1, group))
    if rank < world - 1:
        shape = list(core.shape)
        shape[dim] = hi
        recv_hi = torch.empty(shape, dtype=core.dtype, device=core.device)
        ops.append(dist.P2POp(dist.isend, rows(core, n_local - send_dn, n_local).contiguous(),
                              rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_hi, rank + 1, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    parts = [p for p in (recv_lo, core, recv_hi) if p is not None]
    return torch.cat(parts, dim=dim), lo, hi


def trim(ext, lo, hi, dim):
    idx = [slice(None)] * ext.dim()
    idx[dim] = slice(lo, ext.shape[dim] - hi)
    return ext[tuple(idx)]


def filter_rows(fn, core, halo, dim, group=None):
    """Apply `fn` (tile -> filtered tile, reflecting at its own edges) to a row-sharded raster:
    exchange halos, filter tile+halo, drop the halo rows -- xr_split/xr_merge across GPUs."""
    ext, lo, hi = exchange_halo(core, halo, dim, group)
    return trim(fn(ext), lo, hi, dim)


def boxcar_rows(stack, w, group=None):
    """BoxcarFilter(dims=('y','x'), w) on a row-sharded planar stack (..., y_local, x)."""
    import numpy as np
    from . import kernels
    k = np.ones((1,) * (stack.dim() - 2) + (w, w)) / float(w * w)
    if stack.dim() > 4:
        raise NotImplementedError
    return filter_rows(lambda t: kernels.convolve(t, k), stack, w // 2, stack.dim() - 2, group)


def nlmeans_rows(stack, global_ny, r, f, sigma, h, n_eff=-1, patch_mode=0, group=None):
    """NLMeansFilter on a row-sharded planar stack (var, time, y_local, x), joint weights over the
    variables.  r, f: (time, y, x) radii like NLMeansFilter(dims=('time', 'y', 'x')).  The halo
    rows (r_y + f_y) are exchanged once for all variables and dates; reflection happens at the
    global edges.  With r_time = 0 every date is filtered on its own by the LDS-tiled kernels."""
    from . import kernels
    rt, ry, rx = (int(v) for v in r)
    ft, fy, fx = (int(v) for v in f)
    halo = ry + fy
    r0, _ = my_rows(global_ny, group)
    ext, lo, hi = exchange_halo(stack, halo, 2, group)
    nvar, k, ny_ext, nx = ext.shape
    out = torch.empty_like(ext)
    if rt == 0 and ft == 0:
        # (y, x, time, var) view of planar memory: x contiguous
        kernels.pixelwise_nlmeans_3d(
            ext.permute(2, 3, 1, 0), out.permute(2, 3, 1, 0), (ry, rx, 0), (fy, fx, 0), sigma, h,
            n_eff, patch_mode=patch_mode, global_shape=(global_ny, nx, k),
            tile_offset=(r0 - lo, 0, 0), core=((lo, ny_ext - hi), (0, nx), (0, k)))
    else:
        kernels.pixelwise_nlmeans_3d(
            ext.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), (rt, ry, rx), (ft, fy, fx), sigma, h,
            n_eff, patch_mode=patch_mode, global_shape=(k, global_ny, nx),
            tile_offset=(0, r0 - lo, 0), core=((0, k), (lo, ny_ext - hi), (0, nx)))
    return trim(out, lo, hi, 2)


def omnibus_rows(stack, alpha, n, stats=False):
    """OmnibusTest on this rank's rows of a planar stack (4, time, y_local, x): per pixel, no
    exchange."""
    from . import kernels
    return kernels.change_detection(stack[0], stack[1], stack[2], stack[3], alpha=alpha, n=n,
                                    dims=('time', 'y', 'x'), stats=stats)


def nlmeans_then_omnibus(stack, global_ny, r, f, sigma, h, alpha, n, n_eff=-1, patch_mode=0,
                         group=None):
    """The tutorial pipeline (examples/tutorial_s1.ipynb cells 11 and 15) on a row-sharded stack:
    one halo exchange -> non-local means on tile+halo -> omnibus test on the tile's own rows."""
    filtered = nlmeans_rows(stack, global_ny, r, f, sigma, h, n_eff, patch_mode, group)
    return omnibus_rows(filtered, alpha, n)
