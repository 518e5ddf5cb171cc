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


def register():
    if xr is None:
        return False
    from . import change, filters

    class _Accessor:
        def __init__(self, obj):
            self._obj = obj

        def change_omnibus(self, *args, **kwargs):
            return change.omnibus(self._obj, *args, **kwargs)

        def nlmeans(self, *args, **kwargs):
            return filters.nlmeans(self._obj, *args, **kwargs)

        def boxcar(self, *args, **kwargs):
            return filters.boxcar(self._obj, *args, **kwargs)

        def convolve(self, *args, **kwargs):
            return filters.convolution(self._obj, *args, **kwargs)

        def gaussian(self, *args, **kwargs):
            return filters.gaussian(self._obj, *args, **kwargs)

    xr.register_dataset_accessor('nd_amd')(_Accessor)
    xr.register_dataarray_accessor('nd_amd')(_Accessor)
    return True
