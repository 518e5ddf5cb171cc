"""
nd_amd/algorithm.py -- the plugin API of the reference, re-stated (nd/algorithm.py).

  Algorithm       abstract `apply(ds)`, overridable `_buffer(dim)` and `_parallel_dimension(ds)`
                  (nd/algorithm.py:15-35)
  parallelize     decorator adding the keyword-only `njobs` (nd/algorithm.py:38-105).  The
                  reference forks `njobs` processes over halo-buffered chunks of one dimension
                  and concatenates; here the same chunks (same arithmetic: _adapter.split_bounds)
                  are processed one after another on the GPU -- a GPU launch already covers the
                  whole raster, the chunking is kept so that `njobs` means what it meant.
  wrap_algorithm  class -> function `(ds, *init_args, **init_kwargs)` (nd/algorithm.py:108-198)
"""
import inspect
import multiprocessing as mp
from abc import ABC, abstractmethod
from functools import partial

from . import _adapter


class Algorithm(ABC):

    @abstractmethod
    def apply(self, ds):
        """Must be implemented by derived classes (decorate with @parallelize where useful)."""
        return

    def _buffer(self, dim):
        """Halo needed on each side when the data is split along `dim`."""
        return 0

    def _parallel_dimension(self, ds):
        """The dimension along which to split."""
        return 'y'


def _sorted_parameters(parameters):
    # variadic parameters last, parameters without default first (as the reference orders them)
    ordered = sorted(parameters, key=lambda p: (p.kind, p.default is not inspect.Parameter.empty))
    out = []
    for p in ordered:
        if p not in out:
            out.append(p)
    return out


def parallel(fn, dim=None, chunks=None, buffer=0):
    """Chunked application of `fn` along `dim` with halo `buffer` (nd/utils.py:343-401)."""
    if dim is None:
        dim = 'y'
    if chunks is None:
        chunks = mp.cpu_count()

    def wrapper(ds, *args, **kwargs):
        if dim not in ds.dims:
            raise ValueError("The dataset has no dimension '{}'.".format(dim))
        parts = list(_adapter.xr_split(ds, dim=dim, chunks=chunks, buffer=buffer))
        output = [fn(part, *args, **kwargs) for part in parts]
        return _adapter.xr_merge(output, dim=dim, buffer=buffer)

    return wrapper


def parallelize(func):
    """Give `func(self, ds, ...)` the keyword-only argument `njobs` (nd/algorithm.py:38-105):
    1 = call `func` on the whole dataset, -1 = one chunk per CPU core, n = n chunks along
    `self._parallel_dimension(ds)`, each extended by `self._buffer(dim)` and trimmed on merge."""

    def run(self, ds, *args, njobs=1, **kwargs):
        call = partial(func, self)
        chunks = mp.cpu_count() if njobs == -1 else int(njobs)
        if chunks > 1:
            dim = self._parallel_dimension(ds)
            halo = int(self._buffer(dim))
            # never more chunks than the dimension can carry with its halo
            chunks = max(1, min(chunks, ds.sizes[dim] // (2 * halo + 1)))
            if chunks > 1:
                return parallel(call, dim=dim, chunks=chunks, buffer=halo)(ds, *args, **kwargs)
        return call(ds, *args, **kwargs)

    own = inspect.signature(func)
    njobs_param = inspect.signature(run).parameters['njobs']
    run.__signature__ = own.replace(
        parameters=_sorted_parameters(tuple(own.parameters.values()) + (njobs_param,)))
    run.__doc__ = (func.__doc__ or '').rstrip() + (
        '\n        njobs : int, optional\n'
        '            Number of chunks to process separately (halo-buffered, merged afterwards).\n'
        '            -1 uses the number of available cores; 1 disables chunking (default).\n')
    run.__name__ = getattr(func, '__name__', 'apply')
    return run


def extract_arguments(fn, args, kwargs):
    """Split (*args, **kwargs) into the parameters of `fn` and the leftovers
    ('args' / 'kwargs' entries), like nd/utils.py:727-749."""
    def _(*args, **kwargs):
        pass
    sig = inspect.signature(fn)
    params = list(sig.parameters.values())
    if params and params[0].name == 'self':
        params = params[1:]
    names = {p.name for p in params}
    extra = [p for p in inspect.signature(_).parameters.values() if p.name not in names]
    new_sig = sig.replace(parameters=_sorted_parameters(params + extra))
    bound = new_sig.bind(*args, **kwargs)
    bound.apply_defaults()
    return bound.arguments


def wrap_algorithm(algo, name=None):
    """Function form of an Algorithm class (nd/algorithm.py:108-198): `f(ds, <init arguments>)` builds
    `algo(<init arguments>)` and returns its `.apply(ds)`; arguments of `apply` itself (such as
    `njobs`) are recognised by name."""
    if not (inspect.isclass(algo) and issubclass(algo, Algorithm)):
        raise ValueError('Class must be an instance of `nd.Algorithm`.')

    def function(*args, **kwargs):
        for_apply = dict(extract_arguments(algo.apply, args, kwargs))
        instance = algo(*for_apply.pop('args', ()), **for_apply.pop('kwargs', {}))
        return instance.apply(**for_apply)

    function.__module__ = algo.__module__
    if name is not None:
        function.__name__ = function.__qualname__ = name
    init_sig = inspect.signature(algo.__init__)
    merged = [p for sig in (inspect.signature(algo.apply), init_sig)
              for p in list(sig.parameters.values())[1:]]                 # drop `self`
    function.__signature__ = init_sig.replace(parameters=_sorted_parameters(merged))
    function.__doc__ = 'Wrapper for :class:`{}.{}`.\n\n{}'.format(algo.__module__, algo.__name__,
                                                                  algo.__doc__ or '')
    return function
