"""
nd_amd/xr_lite.py -- a small labelled-array container with the part of the
xarray API that the reference's Algorithm classes touch (`dims`, `data_vars`,
`values`, `copy(deep)`, `isel`, `transpose`, `to_array`, `concat`, `equals`).

xarray is not installed in the build or GPU images, so the `.apply(ds)` surface
is developed and tested against this container; real `xarray.Dataset` /
`DataArray` objects go through the same code paths when xarray is importable
(nd_amd/_adapter.py picks the namespace from the type of the input).

Values may be numpy arrays or torch tensors (device-resident pipelines).
"""
from collections import OrderedDict

import numpy as np

try:                                    # torch is plumbing here, never required for host logic
    import torch
except Exception:                       # pragma: no cover
    torch = None


def _is_torch(a):
    return torch is not None and isinstance(a, torch.Tensor)


def _copy(a):
    return a.clone() if _is_torch(a) else np.array(a, copy=True)


def _transpose(a, axes):
    return a.permute(*axes) if _is_torch(a) else np.transpose(a, axes)


def _equal(a, b):
    if _is_torch(a) or _is_torch(b):
        a = a if _is_torch(a) else torch.as_tensor(a)
        b = b if _is_torch(b) else torch.as_tensor(b)
        if a.shape != b.shape:
            return False
        return bool(torch.all((a.to(b.device) == b) | (torch.isnan(a.to(b.device)) & torch.isnan(b)))
                    if a.is_floating_point() and b.is_floating_point()
                    else torch.equal(a.to(b.device), b))
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    try:
        return bool(np.array_equal(a, b, equal_nan=True))
    except TypeError:
        return bool(np.array_equal(a, b))


class DataArray:
    def __init__(self, data, dims=None, coords=None, attrs=None, name=None):
        if not _is_torch(data):
            data = np.asarray(data)
        if dims is None:
            dims = tuple('dim_%d' % i for i in range(data.ndim))
        dims = tuple(dims)
        if len(dims) != data.ndim:
            raise ValueError('dims %r do not match data of %d dimensions' % (dims, data.ndim))
        self._data = data
        self.dims = dims
        self.coords = OrderedDict(coords or {})
        self.attrs = OrderedDict(attrs or {})
        self.name = name

    # -- xarray-like properties --------------------------------------------
    @property
    def values(self):
        return self._data

    @values.setter
    def values(self, v):
        self._data = v

    @property
    def data(self):
        return self._data

    @property
    def shape(self):
        return tuple(self._data.shape)

    @property
    def ndim(self):
        return len(self.dims)

    @property
    def dtype(self):
        return self._data.dtype

    @property
    def sizes(self):
        return OrderedDict(zip(self.dims, self.shape))

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def real(self):
        return DataArray(self._data.real, self.dims, self.coords, self.attrs, self.name)

    @property
    def imag(self):
        return DataArray(self._data.imag, self.dims, self.coords, self.attrs, self.name)

    # -- methods -------------------------------------------------------------
    def copy(self, deep=True):
        return DataArray(_copy(self._data) if deep else self._data, self.dims,
                         self.coords, self.attrs, self.name)

    def transpose(self, *dims):
        if not dims:
            dims = self.dims[::-1]
        if set(dims) != set(self.dims) or len(dims) != len(self.dims):
            raise ValueError('%r is not a permutation of %r' % (dims, self.dims))
        axes = [self.dims.index(d) for d in dims]
        return DataArray(_transpose(self._data, axes), dims, self.coords, self.attrs, self.name)

    def isel(self, **indexers):
        idx = []
        dims = []
        coords = OrderedDict(self.coords)
        for d in self.dims:
            sel = indexers.get(d, slice(None))
            idx.append(sel)
            if isinstance(sel, slice):
                dims.append(d)
                if d in coords and np.ndim(coords[d]) == 1:
                    coords[d] = coords[d][sel]
            else:
                coords.pop(d, None)
        unknown = set(indexers) - set(self.dims)
        if unknown:
            raise ValueError('dimensions %r do not exist' % sorted(unknown))
        return DataArray(self._data[tuple(idx)], dims, coords, self.attrs, self.name)

    def equals(self, other):
        return (isinstance(other, DataArray) and self.dims == other.dims
                and _equal(self._data, other._data))

    def to_dataset(self, name=None):
        name = name or self.name
        if name is None:
            raise ValueError('unable to convert unnamed DataArray to a Dataset')
        return Dataset({name: self}, coords=self.coords, attrs=self.attrs)

    def all(self):
        return bool(self._data.all())

    def sum(self, dim=None):
        if dim is None:
            return self._data.sum()
        ax = self.dims.index(dim)
        dims = tuple(d for d in self.dims if d != dim)
        return DataArray(self._data.sum(ax), dims, self.coords, self.attrs, self.name)

    def __eq__(self, other):
        o = other._data if isinstance(other, DataArray) else other
        return DataArray(self._data == o, self.dims, self.coords, self.attrs, self.name)

    __hash__ = None

    def _binop(self, other, op):
        o = other._data if isinstance(other, DataArray) else other
        return DataArray(op(self._data, o), self.dims, self.coords, self.attrs, self.name)

    def __add__(self, other):
        return self._binop(other, lambda a, b: a + b)

    def __sub__(self, other):
        return self._binop(other, lambda a, b: a - b)

    def __mul__(self, other):
        return self._binop(other, lambda a, b: a * b)

    def __repr__(self):
        return '<nd_amd.xr_lite.DataArray %r %s %s>' % (
            self.name, ', '.join('%s: %d' % kv for kv in self.sizes.items()), self.dtype)


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.data_vars = OrderedDict()
        self.coords = OrderedDict(coords or {})
        self.attrs = OrderedDict(attrs or {})
        for k, v in (data_vars or {}).items():
            self[k] = v

    # -- mapping ---------------------------------------------------------------
    def __setitem__(self, name, value):
        if isinstance(value, DataArray):
            da = DataArray(value.values, value.dims, self.coords, value.attrs, name)
        elif isinstance(value, tuple) and len(value) >= 2:
            da = DataArray(value[1], value[0], self.coords,
                           value[2] if len(value) > 2 else None, name)
        else:
            raise TypeError('assign a DataArray or a (dims, data) tuple')
        for d, n in da.sizes.items():
            if d in self.dims and self.dims[d] != n:
                raise ValueError('conflicting size for dimension %r: %d vs %d'
                                 % (d, n, self.dims[d]))
        self.data_vars[name] = da

    def __getitem__(self, key):
        if isinstance(key, (list, tuple)):
            return Dataset(OrderedDict((k, self.data_vars[k]) for k in key), self.coords, self.attrs)
        if key in self.data_vars:
            return self.data_vars[key]
        if key in self.coords:
            return DataArray(np.asarray(self.coords[key]), (key,), name=key)
        if key in self.dims:
            return DataArray(np.arange(self.dims[key]), (key,), name=key)
        raise KeyError(key)

    def __delitem__(self, key):
        del self.data_vars[key]

    def __contains__(self, key):
        return key in self.data_vars

    def __iter__(self):
        return iter(self.data_vars)

    def __getattr__(self, name):
        dv = self.__dict__.get('data_vars', {})
        if name in dv:
            return dv[name]
        raise AttributeError(name)

    @property
    def dims(self):
        out = OrderedDict()
        for da in self.data_vars.values():
            for d, n in da.sizes.items():
                out.setdefault(d, n)
        return OrderedDict(sorted(out.items()))      # xarray reports Dataset.dims sorted

    @property
    def sizes(self):
        return self.dims

    @property
    def chunks(self):
        return {}

    def persist(self):
        return self

    def copy(self, deep=False):
        return Dataset(OrderedDict((k, v.copy(deep)) for k, v in self.data_vars.items()),
                       self.coords, self.attrs)

    def isel(self, **indexers):
        coords = OrderedDict(self.coords)
        for d, sel in indexers.items():
            if d not in self.dims:
                raise ValueError('dimension %r does not exist' % d)
            if d in coords and np.ndim(coords[d]) == 1:
                coords[d] = coords[d][sel] if isinstance(sel, slice) else None
                if coords[d] is None:
                    del coords[d]
        out = Dataset(coords=coords, attrs=self.attrs)
        for k, v in self.data_vars.items():
            out.data_vars[k] = v.isel(**{d: s for d, s in indexers.items() if d in v.dims})
            out.data_vars[k].coords = coords
        return out

    def to_array(self, dim='variable'):
        names = list(self.data_vars)
        arrs = [self.data_vars[n] for n in names]
        d0 = arrs[0].dims
        arrs = [a if a.dims == d0 else a.transpose(*d0) for a in arrs]
        if _is_torch(arrs[0].values):
            data = torch.stack([a.values for a in arrs], dim=0)
        else:
            data = np.stack([a.values for a in arrs], axis=0)
        coords = OrderedDict(self.coords)
        coords[dim] = np.array(names)
        return DataArray(data, (dim,) + tuple(d0), coords, self.attrs)

    def equals(self, other):
        return (isinstance(other, Dataset) and list(self.data_vars) == list(other.data_vars)
                and all(self.data_vars[k].equals(other.data_vars[k]) for k in self.data_vars))

    def __repr__(self):
        return '<nd_amd.xr_lite.Dataset (%s) vars=%s>' % (
            ', '.join('%s: %d' % kv for kv in self.dims.items()), list(self.data_vars))


def concat(objs, dim):
    objs = list(objs)
    first = objs[0]
    if isinstance(first, DataArray):
        ax = first.dims.index(dim)
        vals = [o.values for o in objs]
        data = torch.cat(vals, dim=ax) if _is_torch(vals[0]) else np.concatenate(vals, axis=ax)
        coords = OrderedDict(first.coords)
        if dim in coords and all(dim in o.coords for o in objs):
            coords[dim] = np.concatenate([np.asarray(o.coords[dim]) for o in objs])
        return DataArray(data, first.dims, coords, first.attrs, first.name)
    coords = OrderedDict(first.coords)
    if dim in coords and all(dim in o.coords for o in objs):
        coords[dim] = np.concatenate([np.asarray(o.coords[dim]) for o in objs])
    out = Dataset(coords=coords, attrs=first.attrs)
    for k, v in first.data_vars.items():
        if dim in v.dims:
            out[k] = concat([o.data_vars[k] for o in objs], dim)
        else:
            out[k] = v
    return out


def expand_variables(da, dim='variable'):
    """Inverse of Dataset.to_array() (nd/utils.py:472-499)."""
    ax = da.dims.index(dim)
    names = [str(n) for n in np.asarray(da.coords[dim])]
    dims = tuple(d for d in da.dims if d != dim)
    coords = OrderedDict((k, v) for k, v in da.coords.items() if k != dim)
    out = Dataset(coords=coords, attrs=da.attrs)
    for i, n in enumerate(names):
        idx = [slice(None)] * da.ndim
        idx[ax] = i
        out[n] = DataArray(da.values[tuple(idx)], dims)
    return out
