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
    def wrapper(self, ds, *args, njobs=1, **kwargs):
        method = partial(func, self)
        if njobs == -1:
            njobs = mp.cpu_count()
        if njobs == 1:
            return method(ds, *args, **kwargs)
        dim = self._parallel_dimension(ds)
        buffer = self._buffer(dim)
        # never more chunks than the dimension can carry with its halo
        n = ds.sizes[dim]
        chunks = max(1, min(int(njobs), n // max(1, 2 * int(buffer) + 1)))
        if chunks == 1:
            return method(ds, *args, **kwargs)
        return parallel(method, dim=dim, chunks=chunks, buffer=buffer)(ds, *args, **kwargs)

    sig_func = inspect.signature(func)
    sig_wrapper = inspect.signature(wrapper)
    parameters = tuple(sig_func.parameters.values()) + (sig_wrapper.parameters['njobs'],)
    wrapper.__signature__ = sig_func.replace(parameters=_sorted_parameters(parameters))
    doc = func.__doc__ or ''
    wrapper.__doc__ = doc.rstrip() + (
        '\n        njobs : int, optional\n'
        '            Number of chunks to process separately (halo-buffered, merged afterwards).\n'
        '            -1 uses the number of available cores; 1 disables chunking (default).\n')
    wrapper.__name__ = getattr(func, '__name__', 'apply')
    return wrapper


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
    if not (inspect.isclass(algo) and issubclass(algo, Algorithm)):
        raise ValueError('Class must be an instance of `nd.Algorithm`.')

    def _wrapper(*args, **kwargs):
        apply_kwargs = dict(extract_arguments(algo.apply, args, kwargs))
        init_args = apply_kwargs.pop('args', ())
        init_kwargs = apply_kwargs.pop('kwargs', {})
        return algo(*init_args, **init_kwargs).apply(**apply_kwargs)

    _wrapper.__module__ = algo.__module__
    if name is not None:
        _wrapper.__name__ = name
        _wrapper.__qualname__ = name
    sig_init = inspect.signature(algo.__init__)
    sig_apply = inspect.signature(algo.apply)
    parameters = tuple(sig_apply.parameters.values())[1:] + \
        tuple(sig_init.parameters.values())[1:]
    _wrapper.__signature__ = sig_init.replace(parameters=_sorted_parameters(parameters))
    link = ':class:`{}.{}`'.format(algo.__module__, algo.__name__)
    _wrapper.__doc__ = 'Wrapper for {}.\n\n{}'.format(link, algo.__doc__ or '')
    return _wrapper
