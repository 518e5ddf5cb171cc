"""
nd_amd/algorithm.py -- the plugin API of the reference, re-stated (nd/algorithm.py).

  Algorithm       abstract `apply(ds)`, overridable `_buffer(dim)` and `_parallel_dimension(ds)`
                  (nd/algorithm.py:15-35)
  parallelize     decorator adding the keyword-only `njobs` (nd/algorithm.py:38-105) and
                  `devices`.  The reference forks `njobs` processes over halo-buffered chunks of
                  one dimension and concatenates.  Here the same chunks (same arithmetic:
                  _adapter.split_bounds) go to the GPUs: with several devices (`devices=[0, 1, ..]`,
                  or `njobs > 1` on a machine that has more than one) chunk i runs on device
                  i mod n, one host thread and one HIP stream per device, halos included in the
                  chunk exactly like xr_split's buffer; with one device the chunks run one after
                  another on it (a GPU launch already covers the whole raster, the chunking is
                  kept so that `njobs` means what it meant).
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


def resolve_devices(devices=None, njobs=1):
    """The list of torch devices a chunked run spreads over, or None for "the current device only".
    `devices`: explicit list of device indices / torch.device objects.  Without it, `njobs` workers
    map onto the visible GPUs when there is more than one (njobs = -1: all of them) -- the GPU
    counterpart of the reference's process pool (nd/algorithm.py:57-68)."""
    import torch
    if devices is not None:
        devs = [torch.device('cuda', d) if isinstance(d, int) else torch.device(d) for d in devices]
        if not devs:
            raise ValueError('`devices` must name at least one device')
        for d in devs:
            if d.type != 'cuda':
                raise ValueError('nd_amd computes on ROCm devices only, got %s' % d)
        return devs
    if njobs in (0, 1):
        return None
    # One process per GPU (torch.distributed, torchrun): the other visible GPUs belong to the other
    # ranks.  Spreading over them is opt-in through `devices`.
    import os
    import torch.distributed as dist
    if (dist.is_available() and dist.is_initialized()) or 'LOCAL_RANK' in os.environ or 'RANK' in os.environ:
        return None
    n = torch.cuda.device_count()
    if n <= 1:
        return None
    want = n if njobs == -1 else min(int(njobs), n)
    return [torch.device('cuda', i) for i in range(want)] if want > 1 else None


def parallel(fn, dim=None, chunks=None, buffer=0, devices=None):
    """Chunked application of `fn` along `dim` with halo `buffer` (nd/utils.py:343-401).
    devices: list of torch devices; chunk i is processed on devices[i % len(devices)] by one host
    thread per device (device tensors are moved there peer-to-peer, numpy chunks are uploaded
    there), and the pieces are merged on the first device (or on the host for numpy data)."""
    if dim is None:
        dim = 'y'
    if chunks is None:
        chunks = mp.cpu_count()

    def wrapper(ds, *args, **kwargs):
        if dim not in ds.dims:
            raise ValueError("The dataset has no dimension '{}'.".format(dim))
        nchunks = _adapter.safe_chunks(ds.sizes[dim], chunks, buffer)
        parts = list(_adapter.xr_split(ds, dim=dim, chunks=nchunks, buffer=buffer))

        def merged(output):
            res = _adapter.xr_merge(output, dim=dim, buffer=buffer)
            # safe_chunks exists so that the reference's split / merge arithmetic loses no samples when
            # halo-buffered chunks are trimmed: that promise is checked where it applies (buffer > 0).
            # A function that changes the length along `dim` by itself is the caller's business, as in
            # nd.utils.parallel, which has no such restriction.
            if buffer > 0 and dim in res.dims and res.sizes[dim] != ds.sizes[dim]:
                raise RuntimeError('chunked run returned %d samples along %r, the input has %d'
                                   % (res.sizes[dim], dim, ds.sizes[dim]))
            return res

        if not devices:
            return merged([fn(part, *args, **kwargs) for part in parts])
        import torch
        from concurrent.futures import ThreadPoolExecutor
        home = _adapter.home_device(ds)

        def lane(j):
            # everything chunk-related of one device happens in this thread, on that device's
            # current stream; torch's current device is per thread
            dev = devices[j]
            done = []
            with torch.cuda.device(dev):
                for i in range(j, len(parts), len(devices)):
                    res = fn(_adapter.to_device(parts[i], dev), *args, **kwargs)
                    done.append((i, res if home is None else _adapter.to_device(res, home)))
                torch.cuda.current_stream(dev).synchronize()
            return done

        with ThreadPoolExecutor(max_workers=len(devices)) as pool:
            results = sorted((r for lst in pool.map(lane, range(len(devices))) for r in lst),
                             key=lambda ir: ir[0])
        return merged([r for _, r in results])

    return wrapper


def parallelize(func):
    """Give `func(self, ds, ...)` the keyword-only argument `njobs` (nd/algorithm.py:38-105):
    1 = call `func` on the whole dataset, -1 = one chunk per CPU core, n = n chunks along
    `self._parallel_dimension(ds)`, each extended by `self._buffer(dim)` and trimmed on merge."""

    def run(self, ds, *args, njobs=1, devices=None, **kwargs):
        call = partial(func, self)
        devs = resolve_devices(devices, njobs)
        chunks = mp.cpu_count() if (njobs == -1 and not devs) else int(njobs)
        if devs:
            chunks = max(chunks, len(devs))
            if chunks % len(devs):
                chunks += len(devs) - chunks % len(devs)      # equal load per device
        if chunks > 1:
            dim = self._parallel_dimension(ds)
            halo = int(self._buffer(dim))
            # never more chunks than the dimension can carry with its halo: no empty chunk, no
            # last chunk shorter than the halo (xr_merge would trim rows that were never there)
            chunks = max(1, min(chunks, ds.sizes[dim] // (2 * halo + 1)))
            chunks = _adapter.safe_chunks(ds.sizes[dim], chunks, halo)
            if chunks > 1:
                return parallel(call, dim=dim, chunks=chunks, buffer=halo,
                                devices=devs)(ds, *args, **kwargs)
        if devs:      # a single chunk after all: run it on the first listed device
            import torch
            with torch.cuda.device(devs[0]):
                return call(ds, *args, **kwargs)
        return call(ds, *args, **kwargs)

    own = inspect.signature(func)
    extra = tuple(inspect.signature(run).parameters[n] for n in ('njobs', 'devices'))
    run.__signature__ = own.replace(
        parameters=_sorted_parameters(tuple(own.parameters.values()) + extra))
    run.__doc__ = (func.__doc__ or '').rstrip() + (
        '\n        njobs : int, optional\n'
        '            Number of chunks to process separately (halo-buffered, merged afterwards).\n'
        '            -1 uses the number of available cores; 1 disables chunking (default).\n'
        '        devices : list, optional\n'
        '            ROCm devices to spread the chunks over (default: the current device; with\n'
        '            njobs > 1 and several visible GPUs, the first njobs of them).\n')
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
