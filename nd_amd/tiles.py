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

Memory: a rank's block lives in a `RowShard`, a buffer that already has room for the neighbours'
rows in front of and behind the block's own rows (288 GB of HBM make the two margins free).  The
exchange receives straight into those margins; the block itself is never copied, and every size
in the exchange follows from `row_partition` arithmetic -- no size collective, no host sync.
"""
import math
import os

import torch
import torch.distributed as dist


# the windowed filters' default form on a shard with neighbours: filter the rows that need no halo
# while the halo rows travel (True), or exchange first and filter in one launch (False)
HALO_OVERLAP = os.environ.get('ND_AMD_HALO_OVERLAP', '1') != '0'


def row_partition(n, parts):
    """[(r0, r1)] per part, xr_split's chunking without the buffer (nd/utils.py:305-310)."""
    cs = int(math.ceil(n / parts)) if parts > 0 else n
    return [(min(i * cs, n), min((i + 1) * cs, n)) for i in range(parts)]


def _world_rank(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _peer(group, group_rank):
    """P2POp wants GLOBAL ranks; `group_rank` is a rank inside `group`."""
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def my_rows(n, group=None):
    world, rank = _world_rank(group)
    return row_partition(n, world)[rank]


def check_partition(global_n, world, halo):
    """Every block must be able to serve a full halo to its neighbours (the reference's chunks
    overlap by `buffer` in the same way).  Pure arithmetic on (global_n, world, halo): every rank
    reaches the same verdict, so either all of them raise here or none does -- before any
    point-to-point operation is posted (a rank raising alone would leave its neighbours waiting)."""
    if world <= 1 or halo <= 0:
        return
    counts = [b - a for a, b in row_partition(global_n, world)]
    if any(c < halo for c in counts):
        raise ValueError('row blocks (%s rows) are smaller than the halo (%d): use fewer ranks'
                         % (counts, halo))


def halo_extent(global_n, halo, rank, world):
    """(r0, r1, lo, hi): this rank's rows and how many halo rows it holds in front / behind
    (none at the raster's own edges, where the kernels reflect)."""
    r0, r1 = row_partition(global_n, world)[rank]
    lo = halo if (rank > 0 and halo > 0) else 0
    hi = halo if (rank < world - 1 and halo > 0) else 0
    return r0, r1, lo, hi


def _rows(t, dim, a, b):
    idx = [slice(None)] * t.dim()
    idx[dim] = slice(a, b)
    return t[tuple(idx)]


class RowShard:
    """Rows [r0, r1) of a raster whose row axis is `dim` of `ext`, stored with `lo` / `hi` margin
    rows for the neighbours' halos: `ext` = rows [r0 - lo, r1 + hi) of the whole raster once
    `exchange_halo_` has run, `core` = the view of the block's own rows."""

    __slots__ = ('ext', 'dim', 'lo', 'hi', 'r0', 'r1', 'global_n', 'halo', 'bufs')

    def __init__(self, ext, dim, lo, hi, r0, r1, global_n, halo):
        self.ext, self.dim, self.lo, self.hi = ext, dim, lo, hi
        self.r0, self.r1, self.global_n, self.halo = r0, r1, global_n, halo
        self.bufs = {}            # packed send / landing buffers of the halo exchange, kept across steps
        if ext.shape[dim] != (r1 - r0) + lo + hi:
            raise ValueError('shard buffer has %d rows, expected %d + %d + %d'
                             % (ext.shape[dim], lo, r1 - r0, hi))

    @property
    def core(self):
        return _rows(self.ext, self.dim, self.lo, self.lo + (self.r1 - self.r0))

    @property
    def n_local(self):
        return self.r1 - self.r0

    def like(self, ext):
        """A shard with the same geometry around another buffer (a filter's output)."""
        return RowShard(ext, self.dim, self.lo, self.hi, self.r0, self.r1, self.global_n, self.halo)


def empty_shard(lead, global_ny, nx, halo, device, dtype=torch.float32, group=None, rank=None,
                world=None):
    """Uninitialised shard of a planar raster (*lead, y, x) for this rank (or an explicit
    rank / world): rows [r0 - lo, r1 + hi), x fastest.  With two leading axes (variable, date) the
    date planes get nd_amd.synth's padding, which the streaming kernels like."""
    from . import synth
    if world is None or rank is None:
        world, rank = _world_rank(group)
    check_partition(global_ny, world, halo)
    r0, r1, lo, hi = halo_extent(global_ny, halo, rank, world)
    lead = tuple(int(v) for v in lead)
    rows = (r1 - r0) + lo + hi
    if len(lead) == 2:
        ext = synth.empty_stack(lead[0], lead[1], rows, nx, device, dtype)
    else:
        ext = torch.empty(lead + (rows, nx), dtype=dtype, device=device)
    return RowShard(ext, len(lead), lo, hi, r0, r1, global_ny, halo)


def shard_of(full, halo, dim, rank, world, device=None):
    """The shard (margins already filled) a rank would hold of the whole raster `full`: the
    scatter form, for data that starts in one place -- xr_split with its buffer."""
    n = full.shape[dim]
    check_partition(n, world, halo)
    r0, r1, lo, hi = halo_extent(n, halo, rank, world)
    ext = _rows(full, dim, r0 - lo, r1 + hi)
    if device is not None:
        ext = ext.to(device)
    return RowShard(ext, dim, lo, hi, r0, r1, n, halo)


class HaloExchange:
    """A halo exchange in flight (exchange_halo_begin): `wait()` makes the current stream wait for
    it and unpacks what did not land in place."""

    def __init__(self, shard, reqs, landing, nbytes):
        self.shard, self.reqs, self.landing, self.nbytes = shard, reqs, landing, nbytes

    def wait(self):
        for req in self.reqs:
            req.wait()
        for view, buf in self.landing:
            view.copy_(buf, non_blocking=True)
        self.reqs, self.landing = [], []
        return self.shard


def exchange_halo_begin(shard, group=None):
    """Post the exchange of `shard`'s halo rows with the neighbouring ranks and return at once:
    one batched group of point-to-point operations (two sends, two receives at most).  Only
    halo-sized pieces are ever packed or unpacked, into buffers the shard keeps from step to step;
    the block itself stays where it is.  With RCCL the transfers run on the communicator's own
    stream: whatever is launched before `wait()` -- the filter on the rows that need no halo --
    overlaps them."""
    world, rank = _world_rank(group)
    check_partition(shard.global_n, world, shard.halo)        # identical verdict on every rank
    if world == 1 or shard.halo <= 0:
        return HaloExchange(shard, [], [], 0)
    ext, dim, lo, hi, n = shard.ext, shard.dim, shard.lo, shard.hi, shard.n_local
    halo = shard.halo
    ops, landing = [], []
    nbytes = 0
    # RCCL moves device memory itself.  A `gloo` group (CPU tests, or several ranks sharing one
    # GPU, which RCCL refuses) only moves host memory: halo pieces of device shards are staged
    # through host buffers there -- halo-sized copies, the block still never moves.
    via_host = ext.is_cuda and dist.get_backend(group) == 'gloo'

    def kept(key, like):
        buf = shard.bufs.get(key)
        if buf is None or buf.shape != like.shape or buf.dtype != like.dtype or buf.device != like.device:
            buf = torch.empty(like.shape, dtype=like.dtype, device=like.device)
            shard.bufs[key] = buf
        return buf

    def send_rows(a, b, peer, key):
        view = _rows(ext, dim, a, b)
        piece = view if view.is_contiguous() else kept(key, view)
        if piece is not view:
            piece.copy_(view, non_blocking=True)
        if via_host:
            piece = piece.cpu()
        ops.append(dist.P2POp(dist.isend, piece, _peer(group, peer), group))
        return piece.numel() * piece.element_size()

    def recv_into(view, peer, key):
        direct = view.is_contiguous() and not via_host
        if direct:
            buf = view
        elif via_host:
            buf = torch.empty(view.shape, dtype=view.dtype, device='cpu')
        else:
            buf = kept(key, view)
        ops.append(dist.P2POp(dist.irecv, buf, _peer(group, peer), group))
        if buf is not view:
            landing.append((view, buf))

    if rank > 0:                                   # my first rows go up, rank-1's last rows come in
        nbytes += send_rows(lo, lo + halo, rank - 1, 'send_up')
        recv_into(_rows(ext, dim, 0, lo), rank - 1, 'recv_up')
    if rank < world - 1:
        nbytes += send_rows(lo + n - halo, lo + n, rank + 1, 'send_down')
        recv_into(_rows(ext, dim, lo + n, lo + n + hi), rank + 1, 'recv_down')
    return HaloExchange(shard, dist.batch_isend_irecv(ops), landing, nbytes)


def exchange_halo_(shard, group=None):
    """Fill the margins of `shard` with the neighbours' edge rows, in place (begin + wait)."""
    return exchange_halo_begin(shard, group).wait()


def exchange_halo(core, halo, dim, global_n=None, group=None):
    """Convenience form for a block that was NOT allocated with margins: builds the shard (one
    copy of the block into it), exchanges, returns (ext, lo, hi).  Pipelines should allocate with
    `empty_shard` / `shard_of` instead and call `exchange_halo_`.  `global_n` = rows of the whole
    raster (None: a single process, or the block is the whole raster)."""
    world, rank = _world_rank(group)
    if halo <= 0 or world == 1:
        return core, 0, 0
    if global_n is None:
        raise ValueError('exchange_halo needs the global row count to size the neighbours\' blocks')
    check_partition(global_n, world, halo)
    r0, r1, lo, hi = halo_extent(global_n, halo, rank, world)
    if core.shape[dim] != r1 - r0:
        raise ValueError('this rank holds %d rows, the partition of %d rows over %d ranks gives it %d'
                         % (core.shape[dim], global_n, world, r1 - r0))
    shape = list(core.shape)
    shape[dim] = (r1 - r0) + lo + hi
    ext = torch.empty(shape, dtype=core.dtype, device=core.device)
    shard = RowShard(ext, dim, lo, hi, r0, r1, global_n, halo)
    shard.core.copy_(core)
    exchange_halo_(shard, group)
    return ext, lo, hi


def trim(ext, lo, hi, dim):
    return _rows(ext, dim, lo, ext.shape[dim] - hi)


def filter_rows(fn, core, halo, dim, global_n=None, group=None):
    """Apply `fn` (tile -> filtered tile, reflecting at its own edges) to a row-sharded raster:
    exchange halos, filter tile+halo, drop the halo rows -- xr_split/xr_merge across GPUs.
    `core` may be a RowShard (no copy) or a plain block (copied into a shard once)."""
    if isinstance(core, RowShard):
        exchange_halo_(core, group)
        return trim(fn(core.ext), core.lo, core.hi, core.dim)
    ext, lo, hi = exchange_halo(core, halo, dim, global_n, group)
    return trim(fn(ext), lo, hi, dim)


def _boxcar_kernel(ndim, w):
    import numpy as np
    return np.ones((1,) * (ndim - 2) + (w, w)) / float(w * w)


def boxcar_rows(stack, w, global_ny=None, group=None):
    """BoxcarFilter(dims=('y','x'), w) on a row-sharded planar stack (..., y_local, x): a
    RowShard, or a plain block together with `global_ny`."""
    from . import kernels
    t = stack.ext if isinstance(stack, RowShard) else stack
    if t.dim() > 4:
        raise NotImplementedError
    k = _boxcar_kernel(t.dim(), w)
    return filter_rows(lambda e: kernels.convolve(e, k), stack, w // 2, t.dim() - 2, global_ny, group)


def nlmeans_rows(stack, global_ny, r, f, sigma, h, n_eff=-1, patch_mode=0, group=None, status=None,
                 overlap=None):
    """NLMeansFilter on a row-sharded planar stack (var, time, y_local, x), joint weights over the
    variables.  r, f: (time, y, x) radii like NLMeansFilter(dims=('time', 'y', 'x')).  The halo
    rows (r_y + f_y) are exchanged once for all variables and dates; reflection happens at the
    global edges.  With r_time = 0 every date is filtered on its own by the LDS-tiled kernels.
    `stack`: a RowShard whose halo is r_y + f_y (nothing is copied), or a plain block.

    A RowShard with neighbours is filtered in three launches: the rows that need no halo while the
    exchange is in flight, then the two edge bands (a pixel's value does not depend on how the
    raster is cut, so the result is the single launch's, bit for bit).
    status: int32 device tensor of one element that collects the find_weight flag of n_eff >= 0
    without a host synchronisation (kernels.raise_if_no_solution reads it later).
    overlap: False = exchange first, then ONE launch over all rows; True = the three-launch form
    above; None = HALO_OVERLAP (module switch, ND_AMD_HALO_OVERLAP=0 in the environment turns it
    off).  bench.py compares the two forms bit for bit on every rank before it times anything
    (first contact with RCCL) and falls back to the sequential form on a difference."""
    from . import kernels, synth
    rt, ry, rx = (int(v) for v in r)
    ft, fy, fx = (int(v) for v in f)
    halo = ry + fy
    pending = None
    if isinstance(stack, RowShard):
        if stack.global_n != global_ny or (stack.halo != halo and _world_rank(group)[0] > 1):
            raise ValueError('the shard was allocated for %d rows / halo %d, the filter needs %d / %d'
                             % (stack.global_n, stack.halo, global_ny, halo))
        pending = exchange_halo_begin(stack, group)
        ext, lo, hi, r0 = stack.ext, stack.lo, stack.hi, stack.r0
    else:
        r0, _ = my_rows(global_ny, group)
        ext, lo, hi = exchange_halo(stack, halo, 2, global_ny, group)
    nvar, k, ny_ext, nx = ext.shape
    out = synth.empty_stack(nvar, k, ny_ext, nx, ext.device, ext.dtype)
    scratch = None
    if status is not None:
        status.zero_()
        scratch = torch.zeros_like(status)

    def run(row_lo, row_hi):
        """filter the rows [row_lo, row_hi) of the extended tile"""
        if row_hi <= row_lo:
            return
        if rt == 0 and ft == 0:
            # (y, x, time, var) view of planar memory: x contiguous
            kernels.pixelwise_nlmeans_3d(
                ext.permute(2, 3, 1, 0), out.permute(2, 3, 1, 0), (ry, rx, 0), (fy, fx, 0), sigma, h,
                n_eff, patch_mode=patch_mode, global_shape=(global_ny, nx, k),
                tile_offset=(r0 - lo, 0, 0), core=((row_lo, row_hi), (0, nx), (0, k)), status=scratch)
        else:
            kernels.pixelwise_nlmeans_3d(
                ext.permute(1, 2, 3, 0), out.permute(1, 2, 3, 0), (rt, ry, rx), (ft, fy, fx), sigma, h,
                n_eff, patch_mode=patch_mode, global_shape=(k, global_ny, nx),
                tile_offset=(0, r0 - lo, 0), core=((0, k), (row_lo, row_hi), (0, nx)), status=scratch)
        if status is not None:
            status.bitwise_or_(scratch)

    first, last = lo, ny_ext - hi                  # the block's own rows inside `ext`
    # The overlapped form has run on gloo groups and on ranks sharing one GPU only (no multi-GPU box in
    # development): `overlap=False` / HALO_OVERLAP is the way back should its stream semantics differ
    # under RCCL.
    if overlap is None:
        overlap = HALO_OVERLAP
    if pending is not None and pending.reqs and overlap:
        in_lo = first + (halo if lo else 0)        # rows whose windows stay inside the block
        in_hi = last - (halo if hi else 0)
        if in_hi > in_lo:
            run(in_lo, in_hi)                      # ... while the halo rows travel
            pending.wait()
            run(first, in_lo)
            run(in_hi, last)
        else:
            pending.wait()
            run(first, last)
    else:
        if pending is not None:
            pending.wait()
        run(first, last)
    return trim(out, lo, hi, 2)


def omnibus_rows(stack, alpha, n, stats=False):
    """OmnibusTest on this rank's rows of a planar stack (4, time, y_local, x): per pixel, no
    exchange."""
    from . import kernels
    if isinstance(stack, RowShard):
        stack = stack.core
    return kernels.change_detection(stack[0], stack[1], stack[2], stack[3], alpha=alpha, n=n,
                                    dims=('time', 'y', 'x'), stats=stats)


def omnibus_c3_rows(stack, alpha, n, stats=False):
    """Full-pol omnibus on this rank's rows of a planar stack (9, time, y_local, x)."""
    from . import kernels
    if isinstance(stack, RowShard):
        stack = stack.core
    return kernels.change_detection_c3([stack[c] for c in range(9)], alpha=alpha, n=n,
                                       dims=('time', 'y', 'x'), stats=stats)


def nlmeans_then_omnibus(stack, global_ny, r, f, sigma, h, alpha, n, n_eff=-1, patch_mode=0,
                         group=None):
    """The tutorial pipeline (examples/tutorial_s1.ipynb cells 11 and 15) on a row-sharded stack:
    one halo exchange -> non-local means on tile+halo -> omnibus test on the tile's own rows."""
    filtered = nlmeans_rows(stack, global_ny, r, f, sigma, h, n_eff, patch_mode, group)
    return omnibus_rows(filtered, alpha, n)
