"""
nd_amd/change.py -- OmnibusTest on the GPU behind the reference's interface (nd/change.py).

`OmnibusTest(ml=None, n=1, alpha=0.01, njobs=1).apply(ds)` takes a dataset in covariance-matrix
format (variables C11, C22 real and C12 complex -- or already split into C12__re / C12__im) and
returns a boolean DataArray `change` with dims ('y', 'x', 'time'), exactly as
nd.change.OmnibusTest does (nd/change.py:32-116); the per-pixel work that the reference hands to
`nd._change.change_detection` (nd/change.py:69) runs in the HIP kernels of nd_amd/csrc/omnibus.hip.
"""
import os

import numpy as np
import torch

from . import _adapter, _device, _lib, kernels, synth, xr_lite
from .algorithm import Algorithm, wrap_algorithm
from .io import disassemble_complex

__all__ = ['ChangeDetection', 'OmnibusTest', 'omnibus', 'omnibus_statistics', 'change_count',
           'first_change']

_VARS = ['C11', 'C12__re', 'C12__im', 'C22']      # column order of nd/change.py:66
# full-pol extension (no reference counterpart): plane order of nd_amd_omnibus_c3
_VARS3 = ['C11', 'C22', 'C33', 'C12__re', 'C12__im', 'C13__re', 'C13__im', 'C23__re', 'C23__im']
_COMPLEX = {'C12', 'C13', 'C23'}


class ChangeDetection(Algorithm):
    """Base of the change detectors (nd/change.py:21-29): only carries `njobs`."""

    njobs = 1

    def __init__(self, njobs=1):
        self.njobs = njobs


def _covariance_planes(ds_m, device, names=_VARS):
    """The covariance terms `names` as one planar device stack (len(names), time, y, x), x fastest
    -- the layout the streaming kernel wants (coalesced along x, one plane per date)."""
    arrs = []
    for v in names:
        if v not in ds_m.data_vars:
            raise KeyError("OmnibusTest needs the variables C11, C12 (or C12__re/C12__im) and "
                           "C22 (plus C33, C13, C23 for full-pol data); '%s' is missing" % v)
        da = ds_m[v]
        for d in ('y', 'x', 'time'):
            if d not in da.dims:
                raise ValueError("variable %s lacks dimension '%s'" % (v, d))
        if len(da.dims) != 3:
            raise ValueError('variable %s must have exactly the dimensions y, x, time' % v)
        arrs.append((da.transpose('time', 'y', 'x').values,
                     da.transpose('y', 'x', 'time').values))
    dtype = np.result_type(*[_device.np_dtype(a) for a, _ in arrs])
    if dtype not in (np.float32, np.float64):
        dtype = np.dtype(np.float64)           # integer / half input: the reference would refuse
    tdtype = torch.float32 if dtype == np.float32 else torch.float64
    k, ny, nx = arrs[0][0].shape
    stack = synth.empty_stack(len(names), k, ny, nx, device, tdtype)
    placed = set()
    for i, v in enumerate(names):
        # the two halves of an interleaved complex term: one pass over its memory
        if v.endswith('__re') and v[:-4] + '__im' in names:
            j = names.index(v[:-4] + '__im')
            re, im = arrs[i][1], arrs[j][1]
            if (_device.is_tensor(re) and re.dtype == tdtype
                    and kernels.relayout_planar_complex(re, im, stack[i], stack[j])):
                placed.update((i, j))
    for i, (tyx, yxt) in enumerate(arrs):
        if i in placed:
            continue
        # device data in the reference's (y, x, time) layout goes through the transpose kernel
        if not (_device.is_tensor(yxt) and yxt.dtype == tdtype and kernels.relayout_planar(yxt, stack[i])):
            stack[i].copy_(_device.to_device(tyx, device))
    return stack


def _planes_in_place(ds_m, device, names):
    """Time-first device datasets (the CF / NetCDF order (time, y, x)) already ARE planar: real
    variables are used where they lie (the C ABI takes one pointer per plane set and one set of
    strides), only what does not share their strides -- the two halves of an interleaved complex
    term -- is packed.  Returns the list of planes, or None when the dataset is laid out
    differently (the transpose kernels of _covariance_planes then build the stack)."""
    arrs = []
    for v in names:
        if v not in ds_m.data_vars:
            return None
        da = ds_m[v]
        if set(da.dims) != {'time', 'y', 'x'} or len(da.dims) != 3:
            return None
        a = da.transpose('time', 'y', 'x').values
        if not (_device.is_tensor(a) and a.is_cuda and a.device == torch.device(device)
                and a.dtype in (torch.float32, torch.float64)):
            return None
        arrs.append(a)
    if len({a.dtype for a in arrs}) != 1 or len({tuple(a.shape) for a in arrs}) != 1:
        return None
    ref = next((a for a in arrs if a.is_contiguous()), None)
    if ref is None or ref.numel() == 0 or ref.data_ptr() % 16:
        return None
    out = [a if a.stride() == ref.stride() and a.data_ptr() % 16 == 0 else None for a in arrs]
    for i, v in enumerate(names):
        # the two halves of one interleaved complex term: a single pass over its memory
        if out[i] is None and v.endswith('__re') and v[:-4] + '__im' in names:
            j = names.index(v[:-4] + '__im')
            if out[j] is None:
                pair = kernels.split_complex(arrs[i], arrs[j])
                if pair is not None:
                    out[i], out[j] = pair
    return [a if a is not None else arrs[i].contiguous() for i, a in enumerate(out)]


def _multilook_planes(stack, ml):
    """BoxcarFilter(w=ml) with dims ('y', 'x') (nd/change.py:61-63) applied to the planar stack
    (4, time, y, x): every (variable, date) plane is filtered on its own, exactly as the filter
    does variable by variable on the dataset -- same window, same weights ones / ml**2, scipy's
    'reflect' border -- but without leaving the layout the omnibus kernel reads."""
    kernel = (np.ones((ml, ml), dtype=np.float64) / ml ** 2).reshape(1, 1, ml, ml)
    out = synth.empty_stack(stack.shape[0], stack.shape[1], stack.shape[2], stack.shape[3],
                            stack.device, stack.dtype)
    return kernels.convolve(stack, kernel, out=out)


def _on_device(ds, device, wanted):
    """The covariance variables of a host dataset as device tensors in their own layout (complex
    C12 included): one plain upload each -- re-ordering 6 GB on the host would take seconds, on the
    device it is a 3 ms transpose (nd_amd_relayout_planar)."""
    out = xr_lite.Dataset()
    for name in ds.data_vars:
        if name in wanted:
            da = ds[name]
            out[name] = (tuple(da.dims), _device.to_device(da.values, device))
    return out


def _omnibus_change_detection(ds, alpha=0.01, ml=None, n=1, njobs=1, device=None, stats=False,
                              pol='dual'):
    if pol not in ('dual', 'full'):
        raise ValueError("pol must be 'dual' (C11, C12, C22: the reference's test) or 'full'")
    ns = _adapter.namespace(ds)
    ds.persist() if hasattr(ds, 'persist') else None
    full_pol = pol == 'full'               # 3 x 3 covariance: the extension kernel, opt-in
    # like nd/change.py:66 the dual-pol test picks C11 / C12 / C22 and ignores anything else the
    # dataset carries (a C33 next to them changes nothing)
    wanted = (set(_VARS3) | _COMPLEX) if full_pol else (set(_VARS) | {'C12'})
    present = [v for v in list(ds.data_vars) if v in wanted]
    host = not any(_device.is_tensor(ds[v].values) for v in present)
    dev = _device.device_of(*[ds[v].values for v in present], device=device)
    with torch.cuda.device(dev):
        ds_m = disassemble_complex(_on_device(ds, dev, wanted) if host else ds)
        res = None
        if not full_pol and ml is None and all(v in ds_m.data_vars for v in _VARS):
            # the reference's own layout on the device (numpy inputs were uploaded as they are):
            # the pixel-major kernel reads it directly, no transpose at all
            vals = [ds_m[v] for v in _VARS]
            if all(set(da.dims) == {'y', 'x', 'time'} and len(da.dims) == 3 for da in vals):
                res = kernels.change_detection_pixel_major(
                    *[da.transpose('y', 'x', 'time').values for da in vals], alpha=alpha, n=int(n),
                    stats=stats)
        if full_pol and ml is None and all(v in ds_m.data_vars for v in _VARS3):
            # the same for the full-pol test: nine (y, x, time) variables read where they lie
            vals = [ds_m[v] for v in _VARS3]
            if all(set(da.dims) == {'y', 'x', 'time'} and len(da.dims) == 3 for da in vals):
                res = kernels.change_detection_c3_pixel_major(
                    [da.transpose('y', 'x', 'time').values for da in vals], alpha=alpha, n=int(n), stats=stats)
        if res is None:
            names = _VARS3 if full_pol else _VARS
            # (time-first device datasets are planar where they lie -- also in front of the fused
            #  multilooking, which only reads them)
            # ND_AMD_ML_FUSED: 0 never, 2 at every threshold; default: in the sparse regime.  Below it the
            # fused kernel carries the dense_chain search behind its window sums in one 12-wave block per
            # CU (6.2 ms on 24 x 4096^2 at alpha = 0.01) and loses to the boxcar kernel + streaming search
            # (4.1 ms); the maps are the same (tests/test_omnibus_ml_gpu.py runs both at every threshold).
            ml_env = os.environ.get('ND_AMD_ML_FUSED', '1')
            fused_ml = ml is not None and ml_env != '0' and (float(alpha) >= 0.93 or ml_env == '2')
            stack = _planes_in_place(ds_m, dev, names) if (ml is None or fused_ml) else None
            if stack is None:
                stack = _covariance_planes(ds_m, dev, names)
            if fused_ml and not full_pol:
                # multilooking fused into the test: the planes are read once (nd/change.py:61-69)
                res = kernels.change_detection_multilooked(stack[0], stack[1], stack[2], stack[3],
                                                           alpha=alpha, ml=int(ml), stats=stats)
            if res is None:
                if ml is not None:  # spatial multilooking first; the looks multiply accordingly
                    if not torch.is_tensor(stack):
                        stack = _covariance_planes(ds_m, dev, names)
                    stack, n = _multilook_planes(stack, int(ml)), ml * ml
                if full_pol:
                    res = kernels.change_detection_c3(list(stack), alpha=alpha, n=int(n),
                                                      dims=('time', 'y', 'x'), stats=stats)
                else:
                    res = kernels.change_detection(stack[0], stack[1], stack[2], stack[3], alpha=alpha,
                                                   n=int(n), dims=('time', 'y', 'x'), stats=stats)
    change = res[0] if stats else res
    change = change.view(torch.bool)            # 0 / 1 bytes: reinterpreted, not copied
    dims = ['y', 'x', 'time']
    data = _device.to_host(change) if host else change
    change_arr = ns.DataArray(data, dims=dims, coords=ds.coords, attrs=ds.attrs, name='change')
    if not stats:
        return change_arr
    z, P = (_device.to_host(t) if host else t for t in res[1:])
    return (change_arr,
            ns.DataArray(z, dims=['y', 'x'], attrs=ds.attrs, name='z'),
            ns.DataArray(P, dims=['y', 'x'], attrs=ds.attrs, name='P'))


class OmnibusTest(ChangeDetection):
    """Conradsen et al.'s omnibus test for change in a time series of complex-Wishart covariance
    matrices, followed by the sequential search for the dates of change -- the detector of
    nd/change.py:81-116, computed on the GPU.

    ml      optional boxcar window: the covariance terms are first averaged over ml x ml pixels and
            the number of looks becomes ml**2 (nd/change.py:61-63)
    n       number of looks of the data when `ml` is not given (default 1)
    alpha   threshold on the test's probability: a change is declared where P > alpha
            (nd/_change.pyx:239-249; default 0.01, the reference's tests use 0.9)
    device  optional torch device for host inputs (default: the current ROCm device)
    njobs   the reference hands this to its OpenMP loop over image rows (nd/_change.pyx:280); here
            the rows are cut into `njobs` blocks (plus ml // 2 halo rows when multilooking) that
            are spread over the visible GPUs -- one block per GPU launch; with a single GPU the
            whole raster is one launch whatever `njobs` says
    devices explicit list of ROCm devices to spread the row blocks over
    pol     'dual' (default): the reference's 2 x 2 test on C11, C12, C22, whatever else the
            dataset holds.  'full': the same test with p = 3 on C11, C22, C33, C12, C13, C23 -- an
            extension without a reference counterpart, hence opt-in.

    `apply(ds)` returns the boolean DataArray 'change' with dimensions ('y', 'x', 'time')."""

    def __init__(self, ml=None, n=1, alpha=0.01, *args, **kwargs):
        _lib.lib()          # ImportError when libnd_amd.so is missing, like nd/change.py:106-108
        self.device = kwargs.pop('device', None)
        self.devices = kwargs.pop('devices', None)
        self.pol = kwargs.pop('pol', 'dual')
        ChangeDetection.__init__(self, *args, **kwargs)
        self.ml, self.n, self.alpha = ml, n, alpha

    def _buffer(self, dim):
        return int(self.ml) // 2 if (self.ml is not None and dim in ('y', 'x')) else 0

    def apply(self, ds):
        from .algorithm import parallel, resolve_devices
        run = lambda part: _omnibus_change_detection(                      # noqa: E731
            part, alpha=self.alpha, ml=self.ml, n=self.n, njobs=self.njobs, device=self.device,
            pol=self.pol)
        devs = resolve_devices(self.devices, self.njobs) if self.device is None else None
        if devs and 'y' in ds.dims:
            halo = self._buffer('y')
            chunks = max(len(devs), int(self.njobs) if self.njobs and self.njobs > 1 else 1)
            chunks -= chunks % len(devs)
            chunks = max(1, min(chunks, ds.sizes['y'] // (2 * halo + 1)))
            chunks = _adapter.safe_chunks(ds.sizes['y'], chunks, halo)
            if chunks > 1:
                return parallel(run, dim='y', chunks=chunks, buffer=halo, devices=devs)(ds)
            with torch.cuda.device(devs[0]):
                return run(ds)
        return run(ds)


omnibus = wrap_algorithm(OmnibusTest, 'omnibus')


def omnibus_statistics(ds, ml=None, n=1, alpha=0.01, device=None, pol='dual'):
    """change map plus the rasters the reference only computes per pixel: the test statistic
    z = -2 rho ln Q and the probability P of the global test over the whole series
    (nd/_change.pyx:46-77, 133-151).  Returns (change, z, P)."""
    return _omnibus_change_detection(ds, alpha=alpha, ml=ml, n=n, device=device, stats=True, pol=pol)


def change_count(change):
    """Number of detected changes per pixel, `change.sum('time')` of the tutorial
    (examples/tutorial_s1.ipynb cell 20); keeps the container type and device of `change`."""
    ns = _adapter.namespace(change)
    ax = change.dims.index('time')
    vals = change.values
    data = vals.sum(dim=ax) if _device.is_tensor(vals) else vals.sum(axis=ax)
    return ns.DataArray(data, dims=[d for d in change.dims if d != 'time'], attrs=change.attrs,
                        name='change_count')


def first_change(change):
    """Index of the first detected change per pixel along 'time', -1 where there is none."""
    ns = _adapter.namespace(change)
    ax = change.dims.index('time')
    vals = change.values
    if _device.is_tensor(vals):
        any_ = vals.any(dim=ax)
        idx = vals.to(torch.uint8).argmax(dim=ax)
        data = torch.where(any_, idx, torch.full_like(idx, -1))
    else:
        any_ = vals.any(axis=ax)
        data = np.where(any_, vals.argmax(axis=ax), -1)
    return ns.DataArray(data, dims=[d for d in change.dims if d != 'time'], attrs=change.attrs,
                        name='first_change')
