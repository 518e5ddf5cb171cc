"""
nd_amd/_adapter.py -- the few dataset helpers of nd/utils.py the hot path relies on, written so
they serve both real xarray objects (when xarray is importable) and nd_amd.xr_lite.

  get_vars_for_dims   nd/utils.py:450-469
  expand_variables    nd/utils.py:472-499
  is_complex          nd/utils.py:502-524
  xr_split / xr_merge nd/utils.py:288-340   (the halo arithmetic the GPU tile layer reuses)
"""
import numpy as np

from . import xr_lite

try:
    import xarray as xr          # optional
except Exception:                # xarray is absent from the build and GPU images
    xr = None

try:
    import torch
except Exception:                # pragma: no cover
    torch = None


def namespace(obj):
    """xarray or xr_lite, whichever `obj` belongs to."""
    if xr is not None and isinstance(obj, (xr.Dataset, xr.DataArray)):
        return xr
    if isinstance(obj, (xr_lite.Dataset, xr_lite.DataArray)):
        return xr_lite
    raise ValueError('Not an xarray (or nd_amd.xr_lite) Dataset or DataArray: {}'.format(repr(obj)))


def is_dataarray(obj):
    return isinstance(obj, namespace(obj).DataArray)


def iscomplexobj(a):
    if torch is not None and isinstance(a, torch.Tensor):
        return a.is_complex()
    return np.iscomplexobj(a)


def get_vars_for_dims(ds, dims, invert=False):
    """Names of the variables that carry every dimension in `dims` (or, inverted, the others)."""
    wanted = set(dims)
    return [name for name in ds.data_vars if wanted.issubset(ds[name].dims) is not invert]


def expand_variables(da, dim='variable'):
    """Undo `Dataset.to_array()`: one variable per label of `dim`; the array's attributes move to
    the dataset."""
    if namespace(da) is xr_lite:
        return xr_lite.expand_variables(da, dim)
    pieces = {}
    for label in da[dim].values:
        piece = da.sel(**{dim: label}).drop_vars(dim)
        piece.attrs = {}
        pieces[str(label)] = piece
    out = xr.Dataset(pieces)
    out.attrs = dict(da.attrs)
    return out


def empty_like(da):
    """A DataArray with the dims / coords / attrs / name of `da` and uninitialised values of the same
    type, dtype and device: the output buffer of a filter, which overwrites every element."""
    vals = da.values
    blank = torch.empty_like(vals) if (torch is not None and isinstance(vals, torch.Tensor)) \
        else np.empty_like(vals)
    if namespace(da) is xr_lite:
        return xr_lite.DataArray(blank, da.dims, da.coords, da.attrs, da.name)
    return da.copy(deep=False, data=blank)


def is_complex(ds):
    ns = namespace(ds)
    if isinstance(ds, ns.DataArray):
        return iscomplexobj(ds.values)
    return bool(np.any([iscomplexobj(v.values) for v in ds.data_vars.values()]))


def _map_values(obj, fn):
    """A shallow copy of an xr_lite Dataset / DataArray with `fn` applied to every variable's
    values (xarray objects hold numpy arrays only and are returned unchanged)."""
    if namespace(obj) is not xr_lite:
        return obj
    if isinstance(obj, xr_lite.DataArray):
        return xr_lite.DataArray(fn(obj.values), obj.dims, obj.coords, obj.attrs, obj.name)
    out = obj.copy(deep=False)
    for name in list(obj.data_vars):
        da = obj[name]
        out[name] = (tuple(da.dims), fn(da.values), da.attrs)
    return out


def home_device(obj):
    """The device of the first device-resident variable of `obj`, None when all data is on the host."""
    if torch is None or namespace(obj) is not xr_lite:
        return None
    vals = [obj.values] if isinstance(obj, xr_lite.DataArray) else [v.values for v in obj.data_vars.values()]
    for v in vals:
        if isinstance(v, torch.Tensor) and v.is_cuda:
            return v.device
    return None


def to_device(obj, device):
    """Device-resident variables of `obj` moved to `device` (peer-to-peer copy when they live on
    another GPU); host arrays stay where they are (the algorithms upload them themselves, to the
    current device)."""
    def move(v):
        if torch is not None and isinstance(v, torch.Tensor) and v.is_cuda and v.device != device:
            return v.to(device, non_blocking=True)
        return v
    return _map_values(obj, move)


def split_bounds(n, chunks, buffer=0):
    """Index arithmetic of xr_split (nd/utils.py:305-310): chunk i covers
    [max(i*cs - buffer, 0), min((i+1)*cs + buffer, n)) with cs = ceil(n / chunks)."""
    chunksize = int(np.ceil(n / chunks))
    out = []
    for i in range(chunks):
        low = max(i * chunksize - buffer, 0)
        high = min((i + 1) * chunksize + buffer, n)
        out.append((low, high))
    return out


def safe_chunks(n, chunks, buffer=0):
    """The largest chunk count <= `chunks` for which xr_split / xr_merge lose nothing: no empty
    chunk, and a last chunk at least `buffer` long (its neighbour's extension is clipped at the end
    of the data, and xr_merge trims a full `buffer` from it).  The split arithmetic itself is the
    reference's; it has no such guard (nd/utils.py:305-340)."""
    n, chunks, buffer = int(n), max(1, int(chunks)), int(buffer)
    if n <= 0:                            # an empty dimension: one (empty) chunk, as the reference's split yields
        return 1
    while chunks > 1:
        cs = int(np.ceil(n / chunks))
        used = int(np.ceil(n / cs))
        if used < chunks:                 # trailing chunks would be empty
            chunks = used
            continue
        if n - (chunks - 1) * cs >= max(buffer, 1) and cs >= max(buffer, 1):
            break
        chunks -= 1
    return chunks


def xr_split(ds, dim, chunks, buffer=0):
    n = ds.sizes[dim]
    for low, high in split_bounds(n, chunks, buffer):
        yield ds.isel(**{dim: slice(low, high)})


def xr_merge(ds_list, dim, buffer=0):
    """Concatenate chunks produced by xr_split, dropping the `buffer` samples each chunk shares
    with its neighbours (the outer ends of the first and last chunk were never extended)."""
    ns = namespace(ds_list[0])
    halo = int(buffer)
    last = len(ds_list) - 1
    if halo > 0 and last > 0:
        trimmed = []
        for i, part in enumerate(ds_list):
            start = halo if i > 0 else None
            stop = -halo if i < last else None
            trimmed.append(part.isel(**{dim: slice(start, stop)}))
        ds_list = trimmed
    return ns.concat(ds_list, dim=dim)
