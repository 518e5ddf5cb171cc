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
    return [v for v in ds.data_vars
            if set(ds[v].dims).issuperset(set(dims)) != invert]


def expand_variables(da, dim='variable'):
    if namespace(da) is xr_lite:
        return xr_lite.expand_variables(da, dim)
    _vars = []
    attrs = da.attrs
    da.attrs = {}
    for v in da[dim]:
        _var = da.sel(**{dim: v})
        _var.name = str(_var[dim].values)
        del _var[dim]
        _vars.append(_var)
    result = xr.merge(_vars)
    result.attrs = attrs
    return result


def is_complex(ds):
    ns = namespace(ds)
    if isinstance(ds, ns.DataArray):
        return iscomplexobj(ds.values)
    return bool(np.any([iscomplexobj(v.values) for v in ds.data_vars.values()]))


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


def xr_split(ds, dim, chunks, buffer=0):
    n = ds.sizes[dim]
    for low, high in split_bounds(n, chunks, buffer):
        yield ds.isel(**{dim: slice(low, high)})


def xr_merge(ds_list, dim, buffer=0):
    ns = namespace(ds_list[0])
    if buffer > 0 and len(ds_list) > 1:
        idx_first = slice(None, -int(buffer))
        idx_middle = slice(int(buffer), -int(buffer))
        idx_end = slice(int(buffer), None)
        parts = [ds_list[0].isel(**{dim: idx_first})] + \
                [ds.isel(**{dim: idx_middle}) for ds in ds_list[1:-1]] + \
                [ds_list[-1].isel(**{dim: idx_end})]
    else:
        parts = ds_list
    return ns.concat(parts, dim=dim)
