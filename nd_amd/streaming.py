"""
nd_amd/streaming.py -- OmnibusTest over a stack that lives in HOST memory (or is larger than the
device): the raster is cut into row tiles, and the upload of tile i+1 (pinned staging buffer,
copy stream) overlaps the kernels of tile i and the download of tile i-1's change map.

This is the in-memory counterpart of the reference's on-disk tiling (`nd/tiling.py:18-179`: tile,
map over tiles, merge) for the one algorithm that needs no halo.  The omnibus test is per pixel, so
the tiles are independent and the result is bit-identical to the untiled one.  Throughput is bounded
by the host link (PCIe Gen5 x16, 63 GB/s spec -> 164 Mpx/s at 384 B per pixel; measured 125 Mpx/s),
two orders of magnitude below the device-resident rate that bench.py reports.

`nlmeans_omnibus_streamed` is the same pipeline for a windowed stage in front of the test: tiles
carry `buffer` halo rows exactly like `tiling.tile(..., buffer=...)` (nd/tiling.py:18-120).
"""
import os
import warnings
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import kernels, synth


def _pinned_like(shape, dtype):
    return torch.empty(shape, dtype=dtype).pin_memory()


_POOL = None


def _pool():
    # pageable <-> pinned staging copies are plain memcpys that release the GIL: a few threads
    # lift them from ~5 GB/s to the host's memory bandwidth
    global _POOL
    if _POOL is None:
        n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 4)
        _POOL = ThreadPoolExecutor(max_workers=max(2, min(16, n)))
    return _POOL


def _parallel_copy(pairs):
    """[(dst tensor view, src tensor view)] copied concurrently."""
    list(_pool().map(lambda ds: ds[0].copy_(ds[1]), pairs))


def _check_planes(planes):
    with warnings.catch_warnings():
        # read-only inputs (numpy.memmap opened with mode 'r') are only ever read here
        warnings.filterwarnings('ignore', message='The given NumPy array is not writable')
        planes = [torch.from_numpy(np.ascontiguousarray(p)) if isinstance(p, np.ndarray) else p
                  for p in planes]
    if len(planes) != 4:
        raise ValueError('four covariance planes expected')
    p0 = planes[0]
    for p in planes:
        if p.is_cuda or p.shape != p0.shape or p.dtype != p0.dtype or p.dim() != 3:
            raise ValueError('planes must be four host arrays (time, y, x) of one dtype and shape')
    if p0.dtype not in (torch.float32, torch.float64):
        raise TypeError('float32 or float64 expected')
    return planes


def _stream_rows(planes, rows_per_tile, halo, process, device=None, out=None):
    """Row-tile pipeline shared by the streamed entry points.  For every tile of `rows_per_tile`
    rows the rows [r0 - halo, r1 + halo) (clipped to the raster) of the four planes are uploaded
    into a device stack (4, time, rows, x); `process(stack, e0, r0, r1)` -- e0 = first uploaded row
    -- must return the uint8 change map (r1 - r0, x, time) of the tile's own rows as a device
    tensor.  Two staging slots; per tile: H2D on the copy stream, `process` on the compute stream,
    D2H of the change map on the copy stream."""
    planes = _check_planes(planes)
    p0 = planes[0]
    k, ny, nx = p0.shape
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    rows = max(1, min(int(rows_per_tile), ny))
    halo = int(halo)
    ext_rows = min(ny, rows + 2 * halo)
    if out is None:
        out = np.empty((ny, nx, k), np.uint8)
    out_t = torch.from_numpy(out)
    with torch.cuda.device(dev):
        copy_s, comp_s = torch.cuda.Stream(), torch.cuda.Stream()
        slots = []
        for _ in range(2):
            slots.append({
                'dev_in': synth.empty_stack(4, k, ext_rows, nx, dev, p0.dtype),
                'host_out': _pinned_like((rows, nx, k), torch.uint8),
                'uploaded': torch.cuda.Event(), 'computed': torch.cuda.Event(),
                'downloaded': torch.cuda.Event(), 'result': None, 'span': None,
            })
        tiles = [(r0, min(r0 + rows, ny)) for r0 in range(0, ny, rows)]

        def drain(slot):
            # the change map of the tile that used this slot is on the host once `downloaded` fired
            if slot['span'] is not None:
                slot['downloaded'].synchronize()
                r0, r1 = slot['span']
                nr = r1 - r0
                step = max(1, nr // 8)
                _parallel_copy([(out_t[r0 + a:min(r0 + a + step, r1)],
                                 slot['host_out'][a:min(a + step, nr)]) for a in range(0, nr, step)])
                slot['span'] = None

        for i, (r0, r1) in enumerate(tiles):
            s = slots[i % 2]
            drain(s)                                    # slot free again (its D2H finished)
            e0, e1 = max(r0 - halo, 0), min(r1 + halo, ny)
            ne, nr = e1 - e0, r1 - r0
            # one copy per (variable, date): each source block is contiguous in host memory, and
            # the runtime moves pageable memory at close to link speed by itself (measured 56 GB/s
            # on the MI355X host) -- several times faster than staging through a pinned buffer
            # with host threads
            with torch.cuda.stream(copy_s):
                for v in range(4):
                    for t in range(k):
                        s['dev_in'][v, t, :ne].copy_(planes[v][t, e0:e1], non_blocking=True)
                s['uploaded'].record(copy_s)
            with torch.cuda.stream(comp_s):
                comp_s.wait_event(s['uploaded'])
                s['result'] = process(s['dev_in'][:, :, :ne], e0, r0, r1)
                s['computed'].record(comp_s)
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(s['computed'])
                s['host_out'][:nr].copy_(s['result'], non_blocking=True)
                s['result'].record_stream(copy_s)
                s['downloaded'].record(copy_s)
            s['span'] = (r0, r1)
        for s in slots:
            drain(s)
    return out


def omnibus_streamed(planes, alpha, n=1, rows_per_tile=1024, device=None, out=None):
    """planes: four host arrays (numpy, numpy.memmap or CPU tensors) of shape (time, y, x), float32
    or float64, [C11, C12re, C12im, C22].  Returns a uint8 numpy array (y, x, time) (written into
    `out` if given).  The omnibus test is per pixel: tiles need no halo and the result equals the
    untiled one bit for bit."""
    def process(stack, e0, r0, r1):
        return kernels.change_detection(stack[0], stack[1], stack[2], stack[3], alpha=alpha, n=n)
    return _stream_rows(planes, rows_per_tile, 0, process, device, out)


def nlmeans_omnibus_streamed(planes, r, f, sigma, h, alpha, n=1, n_eff=-1, patch_mode=0,
                             rows_per_tile=1024, device=None, out=None):
    """The tutorial pipeline (non-local means over (time, y, x) with joint weights over the four
    covariance terms, then the omnibus test) over a host-resident stack: the reference's
    `tiling.map_over_tiles` with `buffer = r_y + f_y` (nd/tiling.py:243-330), tiles cut along y.
    Every tile is uploaded with its halo rows, filtered with the reflection applied at the edges
    of the WHOLE raster (global_shape / tile_offset of the C ABI), and tested on its own rows, so
    the change map equals the one computed on the whole raster at once.
    r, f: (time, y, x) radii as in NLMeansFilter(dims=('time', 'y', 'x'))."""
    rt, ry, rx = (int(v) for v in r)
    ft, fy, fx = (int(v) for v in f)
    ny = planes[0].shape[1]

    def process(stack, e0, r0, r1):
        nvar, k, ne, nx = stack.shape
        filtered = torch.empty_like(stack)
        lo, hi = r0 - e0, r1 - e0
        if rt == 0 and ft == 0:
            kernels.pixelwise_nlmeans_3d(
                stack.permute(2, 3, 1, 0), filtered.permute(2, 3, 1, 0), (ry, rx, 0), (fy, fx, 0),
                sigma, h, n_eff, patch_mode=patch_mode, global_shape=(ny, nx, k),
                tile_offset=(e0, 0, 0), core=((lo, hi), (0, nx), (0, k)))
        else:
            kernels.pixelwise_nlmeans_3d(
                stack.permute(1, 2, 3, 0), filtered.permute(1, 2, 3, 0), (rt, ry, rx), (ft, fy, fx),
                sigma, h, n_eff, patch_mode=patch_mode, global_shape=(k, ny, nx),
                tile_offset=(0, e0, 0), core=((0, k), (lo, hi), (0, nx)))
        own = filtered[:, :, lo:hi]
        return kernels.change_detection(own[0], own[1], own[2], own[3], alpha=alpha, n=n)

    return _stream_rows(planes, rows_per_tile, ry + fy, process, device, out)
