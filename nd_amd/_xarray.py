"""
nd_amd/_xarray.py -- optional xarray accessors mirroring nd/_xarray.py:135-161 for the algorithms
this package provides: `ds.nd_amd.change_omnibus(...)`, `ds.nd_amd.nlmeans(...)`,
`ds.nd_amd.boxcar(...)`, `ds.nd_amd.convolve(...)`, `ds.nd_amd.gaussian(...)`.

Registered only when xarray is importable (it is not installed in the build or GPU images); the
accessor name differs from the reference's (`nd`, `filter`) so both can be loaded side by side.
"""
try:
    import xarray as xr
except Exception:            # pragma: no cover - xarray absent here
    xr = None


# accessor method -> (module, function form of the algorithm)
_METHODS = {
    'change_omnibus': ('change', 'omnibus'),
    'nlmeans': ('filters', 'nlmeans'),
    'boxcar': ('filters', 'boxcar'),
    'convolve': ('filters', 'convolution'),
    'gaussian': ('filters', 'gaussian'),
}


def _forward(target):
    def method(accessor, *args, **kwargs):
        return target(accessor._obj, *args, **kwargs)
    method.__name__ = target.__name__
    method.__doc__ = target.__doc__
    return method


def register():
    """Attach the `nd_amd` accessor to xarray Datasets and DataArrays; False without xarray."""
    if xr is None:
        return False
    from . import change, filters
    modules = {'change': change, 'filters': filters}
    namespace = {'__init__': lambda accessor, obj: setattr(accessor, '_obj', obj)}
    for name, (module, function) in _METHODS.items():
        namespace[name] = _forward(getattr(modules[module], function))
    accessor = type('NdAmdAccessor', (), namespace)
    xr.register_dataset_accessor('nd_amd')(accessor)
    xr.register_dataarray_accessor('nd_amd')(accessor)
    return True
