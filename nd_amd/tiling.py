"""
nd_amd/tiling.py -- on-disk tiling of datasets larger than memory: split into (buffered) tiles,
map an algorithm over the tiles, merge -- the role of nd/tiling.py (`tile` :18-120,
`map_over_tiles` :123-198, `debuffer` / `auto_merge` :243-422) without the NetCDF stack (h5netcdf,
dask and xarray are absent from the build and GPU images).

Tile format: ENVI, numpy only.  A tile is a directory

    {prefix}.{dim}_{start}_{stop}.{dim}_{start}_{stop}.envi/
        <variable>.img   raw C-order dump of the variable (the last two axes are lines x samples,
        <variable>.hdr   everything in front of them are bands: ENVI "bsq"); complex variables use
                         ENVI data types 6 / 9
        tile.json        dims of every variable, coordinates, attributes, and -- what the reference
                         leaves as a TODO (nd/tiling.py:4) -- the tile's position in the whole
                         raster and its buffer, so that merging never has to guess the overlap

Interruption safety follows nd/tiling.py:95-100: a tile is written as `<name>.part` and renamed
when complete; existing tiles are skipped, so a run that died half way resumes where it stopped.
`map_over_tiles` applies the same rule to its outputs (the reference's version re-computes and
writes `*_new` files instead; skipping is what a restart wants).

Variables come back as read-only numpy memory maps: an algorithm's `.apply` uploads them to the
GPU tile by tile (nd_amd.filters / nd_amd.change stage host arrays themselves).
"""
import glob
import itertools
import json
import os
import shutil
from collections import OrderedDict

import numpy as np

from . import xr_lite

__all__ = ['tile', 'map_over_tiles', 'auto_merge', 'debuffer', 'open_tile', 'write_tile']

_ENVI_TYPES = {'uint8': 1, 'int16': 2, 'int32': 3, 'float32': 4, 'float64': 5, 'complex64': 6,
               'complex128': 9, 'uint16': 12, 'uint32': 13, 'int64': 14, 'uint64': 15, 'bool': 1}
_ENVI_DTYPES = {v: k for k, v in _ENVI_TYPES.items() if k != 'bool'}
EXT = '.envi'


# ------------------------------------------------------------------------------------------
# ENVI files
# ------------------------------------------------------------------------------------------
def _write_envi(path_noext, arr):
    arr = np.asarray(arr)
    if arr.dtype.name not in _ENVI_TYPES:
        raise TypeError('ENVI has no data type for %s' % arr.dtype)
    shape = arr.shape if arr.ndim >= 2 else (1,) * (2 - arr.ndim) + arr.shape
    bands = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    with open(path_noext + '.img', 'wb') as fh:
        # row blocks: never materialise a second copy of a large (possibly memory-mapped) array
        flat = arr.reshape(shape)
        if flat.ndim == 2:
            np.ascontiguousarray(flat).tofile(fh)
        else:
            lead = flat.reshape((-1,) + shape[-2:])
            for b in range(lead.shape[0]):
                np.ascontiguousarray(lead[b]).tofile(fh)
    with open(path_noext + '.hdr', 'w') as fh:
        fh.write('ENVI\ndescription = {nd_amd tile variable}\n')
        fh.write('samples = %d\nlines = %d\nbands = %d\n' % (shape[-1], shape[-2], bands))
        fh.write('header offset = 0\nfile type = ENVI Standard\n')
        fh.write('data type = %d\ninterleave = bsq\nbyte order = 0\n' % _ENVI_TYPES[arr.dtype.name])


def _read_envi_header(hdr_path):
    meta = {}
    with open(hdr_path) as fh:
        for line in fh:
            if '=' in line:
                k, v = line.split('=', 1)
                meta[k.strip().lower()] = v.strip()
    return meta


def _open_envi(path_noext, shape=None, dtype=None):
    h = _read_envi_header(path_noext + '.hdr')
    if h.get('interleave', 'bsq').lower() != 'bsq' or int(h.get('byte order', 0)) != 0:
        raise ValueError('only little-endian bsq ENVI files are read here')
    dt = np.dtype(dtype or _ENVI_DTYPES[int(h['data type'])])
    hdr_shape = (int(h['bands']), int(h['lines']), int(h['samples']))
    shape = tuple(shape) if shape is not None else hdr_shape
    if int(np.prod(shape)) != int(np.prod(hdr_shape)):
        raise ValueError('%s.img holds %s elements, tile.json says %s' % (path_noext, hdr_shape, shape))
    if int(np.prod(shape)) == 0:
        return np.empty(shape, dt)
    return np.memmap(path_noext + '.img', dtype=dt, mode='r', shape=shape,
                     offset=int(h.get('header offset', 0)))


# ------------------------------------------------------------------------------------------
# one tile
# ------------------------------------------------------------------------------------------
def _jsonable(v):
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    return v


def write_tile(ds, tile_path, placement=None):
    """Write the Dataset `ds` as the tile directory `tile_path` (via `<tile_path>.part`).
    placement: {'slices': {dim: [start, stop]}, 'core': {dim: [start, stop]}, 'shape': {dim: n}} --
    where the tile sits in the whole raster, with and without its buffer."""
    if isinstance(ds, xr_lite.DataArray):
        ds = ds.to_dataset(name=ds.name or 'data')
    tmp = tile_path + '.part'
    if os.path.isdir(tmp):
        shutil.rmtree(tmp)                       # leftover of an interrupted run
    os.makedirs(tmp)
    meta = {'variables': OrderedDict(), 'attrs': _jsonable(dict(ds.attrs)), 'coords': {},
            'placement': _jsonable(placement) if placement else None}
    for name, da in ds.data_vars.items():
        vals = da.values
        if hasattr(vals, 'is_cuda'):             # torch tensor
            vals = vals.detach().cpu().numpy()
        vals = np.asarray(vals)
        stored = vals.astype(np.uint8) if vals.dtype == np.bool_ else vals
        _write_envi(os.path.join(tmp, name), stored)
        meta['variables'][name] = {'dims': list(da.dims), 'shape': list(vals.shape),
                                   'dtype': vals.dtype.name, 'attrs': _jsonable(dict(da.attrs))}
    for cname, cvals in ds.coords.items():
        c = np.asarray(cvals)
        if c.dtype.kind in 'M':                  # datetimes: ISO strings
            meta['coords'][cname] = {'datetime64': [str(x) for x in c.astype('datetime64[ns]')]}
        elif c.dtype.kind in 'iufb' or c.dtype.kind in 'US':
            meta['coords'][cname] = {'values': c.tolist()}
    with open(os.path.join(tmp, 'tile.json'), 'w') as fh:
        json.dump(meta, fh)
    if os.path.isdir(tile_path):
        shutil.rmtree(tile_path)
    os.rename(tmp, tile_path)
    return tile_path


def open_tile(tile_path):
    """The tile as an xr_lite.Dataset of read-only memory maps; `.attrs['_placement']` carries the
    tile's position (None for data that was not produced by `tile`)."""
    with open(os.path.join(tile_path, 'tile.json')) as fh:
        meta = json.load(fh, object_pairs_hook=OrderedDict)
    coords = OrderedDict()
    for cname, c in meta.get('coords', {}).items():
        if 'datetime64' in c:
            coords[cname] = np.array(c['datetime64'], dtype='datetime64[ns]')
        else:
            coords[cname] = np.asarray(c['values'])
    ds = xr_lite.Dataset(coords=coords, attrs=dict(meta.get('attrs', {})))
    for name, info in meta['variables'].items():
        dt = np.dtype(info['dtype'])
        arr = _open_envi(os.path.join(tile_path, name), info['shape'],
                         np.uint8 if dt == np.bool_ else dt)
        if dt == np.bool_:
            arr = arr.view(np.bool_) if isinstance(arr, np.memmap) else arr.astype(np.bool_)
        ds[name] = (tuple(info['dims']), arr, dict(info.get('attrs', {})))
    ds.attrs['_placement'] = meta.get('placement')
    return ds


# ------------------------------------------------------------------------------------------
# tile / map / merge
# ------------------------------------------------------------------------------------------
def _dim_slices(n, chunk, buf):
    """[(start, stop, core_start, core_stop)] along one dimension: chunks of `chunk` samples, each
    extended by `buf` on both sides and clipped to [0, n) (nd/tiling.py:53-74)."""
    out = []
    start = 0
    while start < n:
        l = min(chunk, n - start)
        out.append((max(0, start - buf), min(n, start + l + buf), start, start + l))
        start += l
    return out or [(0, 0, 0, 0)]


def tile(ds, path, prefix='part', chunks=None, buffer=0):
    """Split `ds` into tiles and write them below `path` (nd/tiling.py:18-120).
    chunks : {dim: samples per tile} for every dimension to cut (required here: there are no dask
             chunks to fall back on)
    buffer : overlap stored around each tile, one integer or {dim: n}
    Existing tiles are skipped, unfinished ones (`*.part`) rewritten.  Returns the tile paths."""
    if os.path.isfile(path):
        raise ValueError('`path` cannot be a file!')
    os.makedirs(path, exist_ok=True)
    if isinstance(ds, str):
        ds = open_tile(ds)
    if not chunks:
        raise ValueError('`chunks` is required: {dim: samples per tile}')
    sizes = ds.sizes
    per_dim = OrderedDict()
    for dim, c in chunks.items():
        if dim not in sizes:
            raise ValueError("The dataset has no dimension '%s'." % dim)
        buf = buffer if isinstance(buffer, int) else int(buffer.get(dim, 0))
        per_dim[dim] = _dim_slices(sizes[dim], int(c), int(buf))
    paths = []
    for combo in itertools.product(*per_dim.values()):
        sl = OrderedDict((d, (s[0], s[1])) for d, s in zip(per_dim, combo))
        core = OrderedDict((d, (s[2], s[3])) for d, s in zip(per_dim, combo))
        name = '{}.{}{}'.format(prefix, '.'.join('{}_{}_{}'.format(d, a, b) for d, (a, b) in sl.items()), EXT)
        tpath = os.path.join(path, name)
        paths.append(tpath)
        if os.path.isdir(tpath):
            continue                              # skip existing tiles (nd/tiling.py:95-96)
        subset = ds.isel(**{d: slice(a, b) for d, (a, b) in sl.items()})
        write_tile(subset, tpath, placement={'slices': sl, 'core': core,
                                             'shape': {d: sizes[d] for d in per_dim}})
    return paths


def _as_paths(files):
    if isinstance(files, str):
        files = sorted(glob.glob(files))
    return [f for f in files if not f.endswith('.part')]


def map_over_tiles(files, fn, args=(), kwargs=None, path=None, suffix='', merge=True,
                   overwrite=False):
    """Apply `fn(tile_dataset, *args, **kwargs) -> Dataset | DataArray` to every tile and write the
    results as tiles `{stem}{suffix}.envi` into `path` (default: next to the inputs), one at a
    time (nd/tiling.py:123-198).  A result that already exists is kept unless `overwrite`; a result
    is only visible under its final name once it is complete.  merge=True returns the merged
    result (buffers removed), else the list of result paths."""
    kwargs = kwargs or {}
    files = _as_paths(files)
    if not files:
        raise ValueError('No files found!')
    if path is not None:
        os.makedirs(path, exist_ok=True)

    def result_of(f):
        root, name = os.path.split(f.rstrip('/'))
        stem = name[:-len(EXT)] if name.endswith(EXT) else name
        return os.path.join(root if path is None else path, stem + suffix + EXT)

    # Results written next to the inputs match the same glob on the next call (resuming an
    # interrupted run, which the skip-existing rule exists for): a file that is itself the result of
    # another input is a result, not an input.  (The reference never skips, so it never meets this.)
    for f in files:
        if os.path.abspath(result_of(f)) == os.path.abspath(f.rstrip('/')):
            raise ValueError('the result would overwrite its input: give `path` or `suffix`')
    produced = {os.path.abspath(result_of(f)) for f in files}
    dropped = [f for f in files if os.path.abspath(f.rstrip('/')) in produced]
    if dropped:
        import warnings
        warnings.warn('map_over_tiles: %d file(s) match the input pattern but are results of other inputs '
                      '(name + suffix) and are not processed: %s' % (len(dropped), ', '.join(sorted(dropped)[:8])))
    files = [f for f in files if os.path.abspath(f.rstrip('/')) not in produced]
    if not files:
        raise ValueError('No input tiles left: every file is the result of another one')
    results = []
    for f in files:
        out_file = result_of(f)
        if os.path.abspath(out_file) == os.path.abspath(f):
            raise ValueError('the result would overwrite its input: give `path` or `suffix`')
        results.append(out_file)
        if os.path.isdir(out_file) and not overwrite:
            continue
        data = open_tile(f)
        placement = data.attrs.pop('_placement', None)
        result = fn(data, *args, **kwargs)
        if isinstance(result, xr_lite.DataArray):
            result = result.to_dataset(name=result.name or 'data')
        result.attrs.pop('_placement', None)
        write_tile(result, out_file, placement=placement)
        del data, result
    return auto_merge(results) if merge else results


def debuffer(datasets):
    """Cut every tile back to its own samples (nd/tiling.py:243-297), using the placement the tiles
    carry.  Returns [(core dataset, {dim: (start, stop)})]."""
    out = []
    for ds in datasets:
        pl = ds.attrs.get('_placement')
        if not pl:
            raise ValueError('tile without placement information: was it written by nd_amd.tiling.tile?')
        sel = {}
        core = OrderedDict()
        for d, (a, b) in pl['slices'].items():
            ca, cb = pl['core'][d]
            if d in ds.dims:
                sel[d] = slice(ca - a, cb - a)
            core[d] = (ca, cb)
        out.append((ds.isel(**sel), core, pl['shape']))
    return out


def auto_merge(datasets, buffer=True, out=None):
    """Merge tiles (paths, a glob expression, or opened tiles) back into one Dataset
    (nd/tiling.py:336-422): buffers are dropped and every tile's own samples are pasted at their
    place in the whole raster.  out: optional {variable: array} of preallocated targets (e.g.
    numpy.memmap for results larger than memory)."""
    if isinstance(datasets, str):
        datasets = sorted(glob.glob(datasets))
    datasets = [d for d in datasets if not (isinstance(d, str) and d.endswith('.part'))]
    if len(datasets) == 0:
        raise ValueError('No files found!')
    if isinstance(datasets[0], str):
        datasets = [open_tile(p) for p in datasets]
    if not buffer:
        for ds in datasets:
            pl = ds.attrs.get('_placement')
            if pl:
                pl['core'] = dict(pl['slices'])
    parts = debuffer(datasets)
    first, _, gshape = parts[0]
    merged = xr_lite.Dataset(attrs={k: v for k, v in first.attrs.items() if k != '_placement'})
    for name, da in first.data_vars.items():
        shape = tuple(int(gshape[d]) if d in gshape else n for d, n in zip(da.dims, da.shape))
        if out is not None and name in out:
            target = out[name]
            if tuple(target.shape) != shape:
                raise ValueError('out[%r] has shape %s, expected %s' % (name, target.shape, shape))
        else:
            target = np.empty(shape, dtype=np.asarray(da.values[..., :0]).dtype)
        for ds, core, _ in parts:
            idx = tuple(slice(*core[d]) if d in core else slice(None) for d in da.dims)
            target[idx] = ds[name].values
        merged[name] = (tuple(da.dims), target, dict(da.attrs))
    # coordinates of the cut dimensions: stitched from the tiles' own pieces
    for cname in first.coords:
        if cname in gshape:
            full = np.empty(int(gshape[cname]), dtype=np.asarray(first.coords[cname]).dtype)
            for ds, core, _ in parts:
                if cname in ds.coords:
                    full[slice(*core[cname])] = np.asarray(ds.coords[cname])
            merged.coords[cname] = full
        else:
            merged.coords[cname] = first.coords[cname]
    return merged
