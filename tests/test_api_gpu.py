"""The reference's own hot-path tests, re-stated against nd_amd's Algorithm classes
(nd/tests/test_change_omnibus.py, test_change_common.py, test_convolution_filter.py,
test_nlmeans_filter.py, test_filters_common.py).  Datasets are nd_amd.xr_lite containers
(xarray is not installed); the arithmetic runs on the GPU through the C ABI."""
import inspect
from collections import OrderedDict

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _ds(**kw):
    return synth.lite_test_dataset(**kw)


def _allclose(a, b, rtol=1e-5, atol=1e-8):
    assert list(a.data_vars) == list(b.data_vars)
    for v in a.data_vars:
        assert a[v].dims == b[v].dims
        np.testing.assert_allclose(a[v].values, b[v].values, rtol=rtol, atol=atol)


# ---------------------------------------------------------------- change detection
def test_change(device):
    """nd/tests/test_change_omnibus.py:6-19"""
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    dims = OrderedDict([('y', 5), ('x', 5), ('time', 10)])
    ds1 = _ds(dims=dims, mean=[1, 0, 0, 1], sigma=0.1).isel(time=slice(None, 5))
    ds2 = _ds(dims=dims, mean=[10, 0, 0, 10], sigma=0.1).isel(time=slice(5, None))
    ds = xr_lite.concat([ds1, ds2], dim='time')
    changes = OmnibusTest(n=9, alpha=0.9).apply(ds)
    assert changes.dims == ('y', 'x', 'time')
    assert changes.values.dtype == np.bool_
    assert changes.isel(time=5).all()
    assert (changes.sum(dim='time') == 1).all()


def test_change_input_output(device):
    """nd/tests/test_change_common.py:21-32"""
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    ds = _ds(dims=OrderedDict([('y', 20), ('x', 30), ('time', 10)]))
    result = OmnibusTest().apply(ds)
    assert isinstance(result, xr_lite.DataArray)
    assert result.name == 'change'
    assert result.shape == (20, 30, 10)
    assert not result.values.any()              # N(0,1) data: negative determinants -> NaN -> no change


def test_change_function_and_complex_input(oracle, device):
    """omnibus() wrapper == class; a complex C12 variable is split like nd/change.py:59."""
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest, omnibus
    planes = synth.omnibus_stack(seed=5, k=9, ny=12, nx=16, dtype=np.float32, change_frac=0.2)
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    ds = xr_lite.Dataset()
    ds['C11'] = (('y', 'x', 'time'), yxt[0])
    ds['C12'] = (('y', 'x', 'time'), (yxt[1] + 1j * yxt[2]).astype(np.complex64))
    ds['C22'] = (('y', 'x', 'time'), yxt[3])
    a = OmnibusTest(n=9, alpha=0.95).apply(ds)
    b = omnibus(ds, n=9, alpha=0.95)
    assert a.equals(b)
    want = oracle.change_detection_planes(yxt, 0.95, 9)
    np.testing.assert_array_equal(a.values, want.astype(bool))
    assert want.sum() > 0


def test_change_multilook(oracle, device):
    """ml=w: boxcar multilooking first, then n = w**2 looks (nd/change.py:62-64)."""
    import scipy.ndimage as ndi
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    planes = synth.omnibus_stack(seed=6, k=8, ny=24, nx=20, looks=1, dtype=np.float32,
                                 change_frac=0.3, factor=6.0)
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    ds = xr_lite.Dataset()
    for v, a in zip(('C11', 'C12__re', 'C12__im', 'C22'), yxt):
        ds[v] = (('y', 'x', 'time'), a)
    got = OmnibusTest(ml=3, alpha=0.9).apply(ds)
    k = (np.ones((3, 3)) / 9).reshape(3, 3, 1)
    ml = [ndi.convolve(a, k) for a in yxt]
    want = oracle.change_detection_planes(ml, 0.9, 9)
    np.testing.assert_array_equal(got.values, want.astype(bool))


def test_statistics_rasters(oracle, device):
    from nd_amd import xr_lite
    from nd_amd.change import omnibus_statistics
    planes = synth.omnibus_stack(seed=8, k=10, ny=9, nx=14, dtype=np.float64, change_frac=0.2)
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    ds = xr_lite.Dataset()
    for v, a in zip(('C11', 'C12__re', 'C12__im', 'C22'), yxt):
        ds[v] = (('y', 'x', 'time'), a)
    ch, z, P = omnibus_statistics(ds, n=9, alpha=0.9)
    ch0, z0, P0 = oracle.change_detection_planes(yxt, 0.9, 9, stats=True)
    np.testing.assert_array_equal(ch.values, ch0.astype(bool))
    np.testing.assert_allclose(z.values, z0, rtol=1e-5)
    np.testing.assert_allclose(P.values, P0, rtol=1e-5, atol=1e-300)


# ---------------------------------------------------------------- filters, common
def _filter_classes():
    from nd_amd.filters import BoxcarFilter, ConvolutionFilter, GaussianFilter, NLMeansFilter
    return [ConvolutionFilter, BoxcarFilter, NLMeansFilter, GaussianFilter]


@pytest.mark.parametrize('i', range(4))
def test_filter_input_output(device, i):
    """nd/tests/test_filters_common.py:20-33"""
    from nd_amd import xr_lite
    f = _filter_classes()[i]
    ds = _ds(dims=OrderedDict([('y', 20), ('x', 30), ('time', 10)]))
    result = f(dims=('y', 'x')).apply(ds)
    assert isinstance(result, xr_lite.Dataset)
    for v in ds.data_vars:
        assert ds[v].dims == result[v].dims
        assert ds[v].shape == result[v].shape


def test_filter_signature():
    """nd/tests/test_filters_common.py:36-41"""
    for f in _filter_classes():
        assert list(inspect.signature(f._filter).parameters) == ['self', 'arr', 'axes', 'output']


@pytest.mark.parametrize('i', range(4))
def test_filter_mutable_dimension(device, i):
    """nd/tests/test_filters_common.py:44-51"""
    f = _filter_classes()[i]
    ds = _ds(dims=OrderedDict([('y', 20), ('x', 30), ('time', 10)]))
    _allclose(f(dims=('y', 'x')).apply(ds), f(dims=('x', 'y')).apply(ds))


@pytest.mark.parametrize('i', range(4))
@pytest.mark.parametrize('dims', [('x', 'y'), ('x', 'y', 'time')])
def test_parallelized_filter(device, i, dims):
    """nd/tests/test_filters_common.py:54-60: njobs=2 (halo-buffered chunks) == serial"""
    f = _filter_classes()[i]
    ds = _ds(dims=OrderedDict([('y', 20), ('x', 30), ('time', 10)]))
    _allclose(f(dims=dims).apply(ds), f(dims=dims).apply(ds, njobs=2))


def test_inplace_not_implemented(device):
    from nd_amd.filters import BoxcarFilter
    with pytest.raises(NotImplementedError):
        BoxcarFilter().apply(_ds(), inplace=True)


# ---------------------------------------------------------------- convolution
def test_convolve_dataset_identity(device):
    from nd_amd.filters import ConvolutionFilter
    ds = _ds()
    ident = np.zeros((3, 3)); ident[1, 1] = 1
    assert ConvolutionFilter(('y', 'x'), ident).apply(ds).equals(ds)


def test_convolve_dataset(device):
    """nd/tests/test_convolution_filter.py:39-47: bit-equal to scipy.ndimage.convolve"""
    import scipy.ndimage as ndi
    from nd_amd.filters import ConvolutionFilter, _expand_kernel
    ds = _ds()
    np.random.seed(42)
    kernel = np.random.rand(5, 5)
    dims = ('y', 'x')
    nd_kernel = _expand_kernel(kernel, dims, ds.C11.dims)
    np.testing.assert_array_equal(ConvolutionFilter(dims, kernel).apply(ds).C11.values,
                                  ndi.convolve(ds.C11.values, nd_kernel))


def test_convolve_complex(device):
    """nd/tests/test_convolution_filter.py:50-57"""
    from nd_amd.filters import ConvolutionFilter
    from nd_amd.io import assemble_complex
    ds_complex = assemble_complex(_ds())
    assert 'C12' in ds_complex and np.iscomplexobj(ds_complex['C12'].values)
    ident = np.zeros((3, 3)); ident[1, 1] = 1
    out = ConvolutionFilter(('y', 'x'), ident).apply(ds_complex)
    assert out.equals(ds_complex)


def test_boxcar(device):
    """nd/tests/test_convolution_filter.py:60-66"""
    from nd_amd.filters import BoxcarFilter, ConvolutionFilter, boxcar
    ds = _ds()
    w = 5
    kernel = np.ones((w, w)) / w**2
    a = BoxcarFilter(('y', 'x'), w).apply(ds)
    assert a.equals(ConvolutionFilter(('y', 'x'), kernel).apply(ds))
    assert a.equals(boxcar(ds, dims=('y', 'x'), w=w))


def test_expand_kernel():
    from nd_amd.filters import _expand_kernel
    k = _expand_kernel(np.ones((2, 3)), ('x', 'y'), ('x', 'a', 'y', 's'))
    assert k.shape == (2, 1, 3, 1)


# ---------------------------------------------------------------- nlmeans
def test_nlmeans_mean_preserved(device):
    """nd/tests/test_nlmeans_filter.py:9-14"""
    from nd_amd.filters import NLMeansFilter
    ds = _ds()
    out = NLMeansFilter(dims=('y', 'x'), r=0, f=1, sigma=2, h=2).apply(ds)
    for v in ds.data_vars:
        assert abs(ds[v].values.mean() - out[v].values.mean()) < 1e-3


def test_nlmeans_zero_radius_and_empty_dim(device):
    """nd/tests/test_nlmeans_filter.py:17-25"""
    from nd_amd.filters import NLMeansFilter
    ds = _ds()
    assert ds.equals(NLMeansFilter(dims=('y', 'x'), r=0, f=1, sigma=1, h=1).apply(ds))
    assert ds.equals(NLMeansFilter(dims=(), r=1, f=1, sigma=1, h=1).apply(ds))


def test_nlmeans_reduce_std_and_ignore_time(device):
    """nd/tests/test_nlmeans_filter.py:28-43"""
    from nd_amd.filters import NLMeansFilter
    ds = _ds()
    out = NLMeansFilter(dims=('y', 'x', 'time'), r=(1, 1, 0), sigma=2, h=2).apply(ds)
    for v in ds.data_vars:
        assert out[v].values.std() < ds[v].values.std()
    t0 = ds.isel(time=0)
    t0_nlm = NLMeansFilter(dims=('y', 'x'), r=1, sigma=2, h=2).apply(t0)
    for v in ds.data_vars:
        assert np.abs(out[v].isel(time=0).values - t0_nlm[v].values).max() < 1e-8


def test_nlmeans_matches_reference_golden(device):
    """Dataset path == the real reference kernel on the same stacked array."""
    import os
    from nd_amd import xr_lite
    from nd_amd.filters import NLMeansFilter
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'nlmeans_ref.npz'))
    a, want = g['ds_r110_f000__in'], g['ds_r110_f000__out']       # (y, x, time, var)
    names = ['C11', 'C12__im', 'C12__re', 'C22']
    ds = xr_lite.Dataset()
    for i, n in enumerate(names):
        ds[n] = (('y', 'x', 'time'), np.ascontiguousarray(a[..., i]))
    out = NLMeansFilter(dims=('y', 'x', 'time'), r=(1, 1, 0), f=0, sigma=0.5, h=0.7).apply(ds)
    for i, n in enumerate(names):
        np.testing.assert_allclose(out[n].values, want[..., i], rtol=1e-5, atol=0)


def test_device_resident_pipeline(oracle, device):
    """torch tensors in -> torch tensors out, no host copies: boxcar -> omnibus."""
    import torch
    import scipy.ndimage as ndi
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    from nd_amd.filters import BoxcarFilter
    planes = synth.omnibus_stack(seed=9, k=6, ny=16, nx=16, looks=1, dtype=np.float32,
                                 change_frac=0.3, factor=8.0)
    ds = xr_lite.Dataset()
    for v, p in zip(('C11', 'C12__re', 'C12__im', 'C22'), planes):
        ds[v] = (('time', 'y', 'x'), torch.from_numpy(p).to(device))
    sm = BoxcarFilter(dims=('y', 'x'), w=3).apply(ds)
    assert sm['C11'].values.is_cuda
    ch = OmnibusTest(n=9, alpha=0.9).apply(sm)
    assert ch.values.is_cuda and ch.dims == ('y', 'x', 'time')
    k = (np.ones((3, 3)) / 9).reshape(1, 3, 3)
    ml = [np.ascontiguousarray(np.moveaxis(ndi.convolve(p, k), 0, -1)) for p in planes]
    want = oracle.change_detection_planes(ml, 0.9, 9)
    np.testing.assert_array_equal(ch.values.cpu().numpy(), want.astype(bool))


def test_row_tile_pipeline_matches_unsharded(oracle, device):
    """tiles.nlmeans_then_omnibus on manual row tiles (what each rank of a multi-GPU run does,
    minus the exchange, which tests/test_tiles_gloo.py covers) == the unsharded pipeline."""
    import torch
    from nd_amd import kernels, tiles
    planes = synth.omnibus_stack(seed=12, k=8, ny=60, nx=72, looks=4, dtype=np.float32,
                                 change_frac=0.2, factor=5.0)
    stack = torch.from_numpy(np.stack(planes)).to(device)              # (4, k, y, x)
    r, f = (0, 3, 3), (0, 1, 1)
    full_f = tiles.nlmeans_rows(stack, 60, r, f, 0.5, 2.0, patch_mode=1)
    full_ch = tiles.omnibus_rows(full_f, 0.9, 16)
    # unsharded reference through the oracle: nlmeans (patch_mode 1) per date, then omnibus
    a = np.ascontiguousarray(np.stack(planes).transpose(2, 3, 1, 0))   # (y, x, t, var)
    want_f = np.empty_like(a)
    oracle.pixelwise_nlmeans_3d(a, want_f, (3, 3, 0), (1, 1, 0), 0.5, 2.0, -1, njobs=8, patch_mode=1)
    # signed variables cross zero: tolerance relative to the data scale
    np.testing.assert_allclose(full_f.permute(2, 3, 1, 0).cpu().numpy(), want_f, rtol=1e-5,
                               atol=1e-5 * float(np.abs(a).max()))
    # manual sharding with halos and global coordinates
    halo = 4
    out = torch.zeros_like(full_ch)
    for lo, hi in tiles.row_partition(60, 3):
        tlo, thi = max(lo - halo, 0), min(hi + halo, 60)
        ext = stack[:, :, tlo:thi].contiguous()
        fo = torch.empty_like(ext)
        kernels.pixelwise_nlmeans_3d(ext.permute(2, 3, 1, 0), fo.permute(2, 3, 1, 0), (3, 3, 0),
                                     (1, 1, 0), 0.5, 2.0, -1, patch_mode=1,
                                     global_shape=(60, 72, 8), tile_offset=(tlo, 0, 0),
                                     core=((lo - tlo, hi - tlo), (0, 72), (0, 8)))
        core = fo[:, :, lo - tlo:hi - tlo].contiguous()
        out[lo:hi] = tiles.omnibus_rows(core, 0.9, 16)
    assert torch.equal(out, full_ch)
    # 3-D search window (tutorial form) runs through the generic kernel
    f3 = tiles.nlmeans_rows(stack, 60, (1, 2, 2), (0, 0, 0), 0.5, 2.0)
    assert f3.shape == stack.shape and bool(torch.isfinite(f3).all())


def test_complex_torch_tensors_stay_on_device(oracle, device):
    """C12 as a complex64 ROCm tensor: split into C12__re / C12__im on the device
    (nd/change.py:59 / nd/io.py:26-69 for device-resident data)."""
    import torch
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    planes = synth.omnibus_stack(seed=15, k=7, ny=10, nx=12, dtype=np.float32, change_frac=0.3)
    ds = xr_lite.Dataset()
    ds['C11'] = (('time', 'y', 'x'), torch.from_numpy(planes[0]).to(device))
    ds['C12'] = (('time', 'y', 'x'), torch.complex(torch.from_numpy(planes[1]), torch.from_numpy(planes[2])).to(device))
    ds['C22'] = (('time', 'y', 'x'), torch.from_numpy(planes[3]).to(device))
    ch = OmnibusTest(n=9, alpha=0.9).apply(ds)
    assert ch.values.is_cuda
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    np.testing.assert_array_equal(ch.values.cpu().numpy(), oracle.change_detection_planes(yxt, 0.9, 9).astype(bool))


def test_time_first_device_variables_are_used_in_place(oracle, device):
    """(time, y, x) device datasets -- the NetCDF / CF order -- are planar already: real variables
    are read where they lie, the halves of a complex C12 are packed; odd widths, separate
    C12__re / C12__im variables, a sliced (non-contiguous) variable that sends the dataset down
    the general path, float64, and the full-pol test."""
    import torch
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    for dtype, shape in ((np.float32, (9, 11, 37)), (np.float64, (6, 5, 130))):
        k, ny, nx = shape
        planes = synth.omnibus_stack(seed=31 + nx, k=k, ny=ny, nx=nx, dtype=dtype, change_frac=0.3)
        yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
        want = oracle.change_detection_planes(yxt, 0.9, 9).astype(bool)
        t = [torch.from_numpy(p).to(device) for p in planes]
        dims = ('time', 'y', 'x')
        for variant in ('complex', 'split', 'sliced'):
            ds = xr_lite.Dataset()
            if variant == 'sliced':
                big = torch.zeros((k, ny, nx + 3), dtype=t[0].dtype, device=device)
                big[:, :, 1:nx + 1] = t[0]
                ds['C11'] = (dims, big[:, :, 1:nx + 1])
            else:
                ds['C11'] = (dims, t[0])
            if variant == 'complex':
                ds['C12'] = (dims, torch.complex(t[1], t[2]))
            else:
                ds['C12__re'] = (dims, t[1])
                ds['C12__im'] = (dims, t[2])
            ds['C22'] = (dims, t[3])
            ch = OmnibusTest(n=9, alpha=0.9).apply(ds)
            assert ch.values.is_cuda and ch.dims == ('y', 'x', 'time')
            np.testing.assert_array_equal(ch.values.cpu().numpy(), want, err_msg=variant)
    p3 = synth.omnibus_stack_c3(seed=5, k=6, ny=7, nx=41, dtype=np.float32, change_frac=0.3)
    want3 = oracle.change_detection_pol([np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in p3], 3, 0.9, 9)
    ds = xr_lite.Dataset()
    t3 = [torch.from_numpy(p).to(device) for p in p3]
    for i, name in enumerate(('C11', 'C22', 'C33')):
        ds[name] = (('time', 'y', 'x'), t3[i])
    for j, name in enumerate(('C12', 'C13', 'C23')):
        ds[name] = (('time', 'y', 'x'), torch.complex(t3[3 + 2 * j], t3[4 + 2 * j]))
    ch3 = OmnibusTest(n=9, alpha=0.9, pol='full').apply(ds)
    np.testing.assert_array_equal(ch3.values.cpu().numpy(), want3.astype(bool))


def test_nlmeans_dataset_paths_use_fast_layout(oracle, device):
    """Datasets with the reference's (y, x, time) variables, large enough for the re-layout path:
    dims ('y','x') and the tutorial's dims ('time','y','x') against the oracle on the stacked array."""
    from nd_amd import xr_lite
    from nd_amd.filters import NLMeansFilter
    rng = np.random.default_rng(3)
    names = ['C11', 'C12__im', 'C12__re', 'C22']
    ds = xr_lite.Dataset()
    data = {n: rng.gamma(4.0, 0.25, (40, 48, 6)).astype(np.float32) for n in names}
    for n in names:
        ds[n] = (('y', 'x', 'time'), data[n])
    stacked = np.stack([data[n] for n in names], axis=-1)                  # (y, x, t, var)
    # dims ('y', 'x'): reference-compatible and true-patch modes
    for pd, pm in (('reference', 0), ('signed', 1)):
        out = NLMeansFilter(dims=('y', 'x'), r=3, f=1, sigma=0.5, h=0.5, patch_distances=pd).apply(ds)
        want = np.empty_like(stacked)
        oracle.pixelwise_nlmeans_3d(stacked, want, (3, 3, 0), (1, 1, 0), 0.5, 0.5, -1, njobs=8, patch_mode=pm)
        for i, n in enumerate(names):
            np.testing.assert_allclose(out[n].values, want[..., i], rtol=1e-5)
    # tutorial: dims ('time', 'y', 'x'), r = (1, 3, 3), n_eff
    out = NLMeansFilter(dims=('time', 'y', 'x'), r=(1, 3, 3), f=1, sigma=0.5, h=0.5, n_eff=20).apply(ds)
    st = np.ascontiguousarray(stacked.transpose(2, 0, 1, 3))               # (t, y, x, var)
    want = np.empty_like(st)
    oracle.pixelwise_nlmeans_3d(st, want, (1, 3, 3), (1, 1, 1), 0.5, 0.5, 20.0, njobs=8, patch_mode=0)
    for i, n in enumerate(names):
        assert out[n].dims == ('y', 'x', 'time')
        np.testing.assert_array_equal(out[n].values, want[..., i].transpose(1, 2, 0))


def test_host_streamed_omnibus_equals_untiled(oracle, device):
    """nd_amd.streaming: row tiles uploaded / computed / downloaded on overlapping streams give
    the untiled change map (ragged last tile, float32 and float64)."""
    from nd_amd import streaming
    for dtype in (np.float32, np.float64):
        planes = synth.omnibus_stack(seed=23, k=10, ny=150, nx=130, dtype=dtype, change_frac=0.2)
        got = streaming.omnibus_streamed(planes, alpha=0.9, n=9, rows_per_tile=64)
        yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
        np.testing.assert_array_equal(got, oracle.change_detection_planes(yxt, 0.9, 9, njobs=8))


def test_host_streamed_pipeline_equals_whole_raster(oracle, device, tmp_path):
    """nlmeans -> omnibus over a host stack cut into row tiles with halo (the map_over_tiles
    analogue): equal to filtering and testing the whole raster at once, here checked against the
    CPU oracle; inputs come from numpy.memmap files like a raster too large for memory would."""
    from nd_amd import streaming
    planes = synth.omnibus_stack(seed=29, k=6, ny=90, nx=140, dtype=np.float32, change_frac=0.2)
    mm = []
    for i, p in enumerate(planes):
        path = tmp_path / ('plane%d.npy' % i)
        np.save(path, p)
        mm.append(np.load(path, mmap_mode='r'))
    for r, f, pm, ne in (((1, 3, 3), (1, 1, 1), 0, 20.0), ((0, 3, 2), (0, 1, 1), 1, -1)):
        got = streaming.nlmeans_omnibus_streamed(mm, r, f, 0.5, 0.5, alpha=0.9, n=9, n_eff=ne,
                                                 patch_mode=pm, rows_per_tile=32)
        st = np.ascontiguousarray(np.stack(planes, axis=-1))                 # (t, y, x, var)
        filt = np.empty_like(st)
        oracle.pixelwise_nlmeans_3d(st, filt, r, f, 0.5, 0.5, ne, njobs=8, patch_mode=pm)
        if pm == 0:
            yxt = [np.ascontiguousarray(np.moveaxis(filt[..., v], 0, -1)) for v in range(4)]
            np.testing.assert_array_equal(got, oracle.change_detection_planes(yxt, 0.9, 9, njobs=8))
        # whichever mode: the tiled result equals the device's own whole-raster result
        whole = streaming.nlmeans_omnibus_streamed(mm, r, f, 0.5, 0.5, alpha=0.9, n=9, n_eff=ne,
                                                   patch_mode=pm, rows_per_tile=90)
        np.testing.assert_array_equal(got, whole)


def test_change_multilook_device_dataset_and_complex(oracle, device):
    """ml on a device-resident dataset with a complex C12 (the form nd/change.py:59-63 sees): the
    planar multilook path equals BoxcarFilter + OmnibusTest applied one after the other."""
    import torch
    from nd_amd import xr_lite
    from nd_amd.change import OmnibusTest
    from nd_amd.filters import BoxcarFilter
    planes = synth.omnibus_stack(seed=16, k=7, ny=40, nx=150, looks=1, dtype=np.float32,
                                 change_frac=0.3, factor=6.0)
    yxt = [torch.from_numpy(np.ascontiguousarray(np.moveaxis(p, 0, -1))).to(device) for p in planes]
    ds = xr_lite.Dataset()
    ds['C11'] = (('y', 'x', 'time'), yxt[0])
    ds['C12'] = (('y', 'x', 'time'), torch.complex(yxt[1], yxt[2]))
    ds['C22'] = (('y', 'x', 'time'), yxt[3])
    got = OmnibusTest(ml=5, alpha=0.9).apply(ds)
    assert torch.is_tensor(got.values) and got.dims == ('y', 'x', 'time')
    two_step = OmnibusTest(n=25, alpha=0.9).apply(BoxcarFilter(w=5).apply(ds))
    assert torch.equal(got.values, two_step.values)
    import scipy.ndimage as ndi
    k = (np.ones((5, 5)) / 25).reshape(5, 5, 1)
    ml = [ndi.convolve(np.moveaxis(p, 0, -1), k) for p in planes]
    want = oracle.change_detection_planes([np.ascontiguousarray(a) for a in ml], 0.9, 25)
    np.testing.assert_array_equal(got.values.cpu().numpy(), want.astype(bool))


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_relayout_planar_kernel(device, dtype):
    """(y, x, time) -> planar (time, y, x): real arrays, the two halves of a complex array, odd
    sizes, long series, padded destination planes; anything else is declined."""
    import torch
    from nd_amd import kernels, synth as dsynth
    rng = np.random.default_rng(61)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    for ny, nx, k in [(1, 1, 1), (3, 5, 2), (17, 33, 24), (64, 64, 7), (9, 300, 100), (2, 70, 700), (32, 32, 24)]:
        a = rng.normal(size=(ny, nx, k)).astype(dtype)
        src = torch.from_numpy(a).to(device)
        dst = dsynth.empty_stack(1, k, ny, nx, device, tdt)[0]
        assert kernels.relayout_planar(src, dst)
        np.testing.assert_array_equal(dst.cpu().numpy(), a.transpose(2, 0, 1))
        c = torch.complex(src, torch.from_numpy(rng.normal(size=(ny, nx, k)).astype(dtype)).to(device))
        for part in (c.real, c.imag):
            out = torch.full((k, ny, nx), 7.0, dtype=tdt, device=device)
            assert kernels.relayout_planar(part, out)
            assert torch.equal(out, part.permute(2, 0, 1))
    # not the reference layout: declined, destination untouched
    src = torch.zeros((4, 6, 5), dtype=tdt, device=device)
    dst = torch.ones((5, 4, 6), dtype=tdt, device=device)
    assert not kernels.relayout_planar(src.permute(1, 0, 2), dst.permute(0, 2, 1))
    assert not kernels.relayout_planar(src[:, ::2], torch.ones((5, 4, 3), dtype=tdt, device=device))
    assert bool((dst == 1).all())


def test_relayout_round_trip_and_filter_on_reference_layout(device):
    """planar -> (y, x, time) inverse kernel; BoxcarFilter on device variables in the reference's
    layout (real and complex) goes through both transposes and still equals scipy bit for bit."""
    import scipy.ndimage as ndi
    import torch
    from nd_amd import kernels, xr_lite
    from nd_amd.filters import BoxcarFilter
    rng = np.random.default_rng(62)
    for ny, nx, k in [(1, 1, 1), (5, 7, 3), (33, 65, 24), (10, 300, 90)]:
        a = torch.from_numpy(rng.normal(size=(k, ny, nx)).astype(np.float32)).to(device)
        out = torch.full((ny, nx, k), 3.0, device=device)
        assert kernels.relayout_pixel_major(a, out)
        assert torch.equal(out, a.permute(1, 2, 0))
        c = torch.zeros((ny, nx, k), dtype=torch.complex64, device=device)
        assert kernels.relayout_pixel_major(a, c.imag) and kernels.relayout_pixel_major(a * 2, c.real)
        assert torch.equal(c.imag, a.permute(1, 2, 0)) and torch.equal(c.real, (a * 2).permute(1, 2, 0))
    re = rng.normal(size=(70, 260, 5)).astype(np.float32)
    im = rng.normal(size=(70, 260, 5)).astype(np.float32)
    ds = xr_lite.Dataset()
    ds['A'] = (('y', 'x', 'time'), torch.from_numpy(re).to(device))
    ds['B'] = (('y', 'x', 'time'), torch.complex(torch.from_numpy(re), torch.from_numpy(im)).to(device))
    res = BoxcarFilter(w=3).apply(ds)
    kern = (np.ones((3, 3)) / 9).reshape(3, 3, 1)
    np.testing.assert_array_equal(res['A'].values.cpu().numpy(), ndi.convolve(re, kern))
    np.testing.assert_array_equal(res['B'].values.real.cpu().numpy(), ndi.convolve(re, kern))
    np.testing.assert_array_equal(res['B'].values.imag.cpu().numpy(), ndi.convolve(im, kern))


def test_host_dataset_is_staged_once_and_left_untouched(oracle, device):
    """numpy datasets go to the device variable by variable in their own layout; the caller's
    dataset (complex variable, attributes, values) is the same afterwards, the result is numpy and
    equals the oracle."""
    from nd_amd import xr_lite
    from nd_amd.filters import NLMeansFilter, BoxcarFilter
    rng = np.random.default_rng(71)
    a = rng.gamma(4.0, 0.25, (30, 40, 4)).astype(np.float32)
    c = (rng.normal(size=(30, 40, 4)) + 1j * rng.normal(size=(30, 40, 4))).astype(np.complex64)
    ds = xr_lite.Dataset()
    ds['A'] = (('y', 'x', 'time'), a.copy(), {'unit': 'dB'})
    ds['C'] = (('y', 'x', 'time'), c.copy())
    ds['scalar_per_date'] = (('time',), np.arange(4.0))
    out = NLMeansFilter(dims=('y', 'x'), r=2, f=1, sigma=0.5, h=0.5).apply(ds)
    assert sorted(ds.data_vars) == ['A', 'C', 'scalar_per_date'] and np.iscomplexobj(ds['C'].values)
    np.testing.assert_array_equal(ds['A'].values, a)
    np.testing.assert_array_equal(ds['C'].values, c)
    assert sorted(out.data_vars) == ['A', 'C__im', 'C__re', 'scalar_per_date']
    assert all(isinstance(out[n].values, np.ndarray) for n in out.data_vars)
    names = ['A', 'C__re', 'C__im']
    order = [n for n in out.data_vars if n in names]
    stacked = np.ascontiguousarray(np.stack([{'A': a, 'C__re': c.real, 'C__im': c.imag}[n] for n in order], axis=-1))
    want = np.empty_like(stacked)
    oracle.pixelwise_nlmeans_3d(stacked, want, (2, 2, 0), (1, 1, 0), 0.5, 0.5, -1, njobs=4, patch_mode=0)
    for i, n in enumerate(order):
        np.testing.assert_array_equal(out[n].values, want[..., i])
    np.testing.assert_array_equal(out['scalar_per_date'].values, np.arange(4.0))
    box = BoxcarFilter(w=3).apply(ds)
    assert np.iscomplexobj(box['C'].values) and box['A'].attrs.get('unit') == 'dB'


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_relayout_planar_complex_both_halves(device, dtype):
    import torch
    from nd_amd import kernels, synth as dsynth
    rng = np.random.default_rng(63)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    for ny, nx, k in [(1, 1, 1), (7, 9, 3), (40, 70, 24), (3, 200, 130)]:
        c = torch.complex(torch.from_numpy(rng.normal(size=(ny, nx, k)).astype(dtype)),
                          torch.from_numpy(rng.normal(size=(ny, nx, k)).astype(dtype))).to(device)
        st = dsynth.empty_stack(2, k, ny, nx, device, tdt)
        assert kernels.relayout_planar_complex(c.real, c.imag, st[0], st[1])
        assert torch.equal(st[0], c.real.permute(2, 0, 1)) and torch.equal(st[1], c.imag.permute(2, 0, 1))
    # two unrelated real tensors are not the halves of one complex tensor
    a = torch.zeros((4, 5, 6), dtype=tdt, device=device)
    b = torch.zeros((4, 5, 6), dtype=tdt, device=device)
    o = torch.ones((2, 6, 4, 5), dtype=tdt, device=device)
    assert not kernels.relayout_planar_complex(a, b, o[0], o[1])
    assert bool((o == 1).all())


def test_relayout_declines_series_beyond_the_staging_buffer(device):
    """A series too long for the transpose kernels' LDS image is declined (torch copies then)."""
    import torch
    from nd_amd import kernels
    k = 13000
    src = torch.zeros((1, 2, k), device=device)
    dst = torch.ones((k, 1, 2), device=device)
    assert not kernels.relayout_planar(src, dst)
    assert not kernels.relayout_pixel_major(dst, src)
    c = torch.zeros((1, 2, 7000), dtype=torch.complex64, device=device)
    o = torch.ones((2, 7000, 1, 2), device=device)
    assert not kernels.relayout_planar_complex(c.real, c.imag, o[0], o[1])
    assert bool((dst == 1).all()) and bool((o == 1).all())


def test_complex_variable_time_first_split_filter_merge(device):
    """A contiguous complex64 / complex128 device variable in (time, y, x) order: ConvolutionFilter /
    BoxcarFilter split it in one pass, filter two packed arrays and merge them (nd/filters.py:261-265
    filters real and imaginary parts separately) -- bit-equal to scipy on each part."""
    import scipy.ndimage as ndi
    import torch
    from nd_amd import xr_lite
    from nd_amd.filters import BoxcarFilter, ConvolutionFilter
    rng = np.random.default_rng(71)
    for cdtype in (np.complex64, np.complex128):
        z = (rng.normal(size=(3, 90, 257)) + 1j * rng.normal(size=(3, 90, 257))).astype(cdtype)
        ds = xr_lite.Dataset()
        ds['C12'] = (('time', 'y', 'x'), torch.from_numpy(z).to(device))
        ds['C11'] = (('time', 'y', 'x'), torch.from_numpy(np.ascontiguousarray(z.real)).to(device))
        kern = rng.normal(size=(5, 5))
        for flt, k2 in ((BoxcarFilter(w=3), np.ones((3, 3)) / 9.0), (ConvolutionFilter(kernel=kern), kern)):
            out = flt.apply(ds)
            got = out['C12'].values.cpu().numpy()
            k3 = k2.reshape((1,) + k2.shape)
            want = ndi.convolve(z.real, k3) + 1j * ndi.convolve(z.imag, k3)
            np.testing.assert_array_equal(got, want.astype(cdtype))
            np.testing.assert_array_equal(out['C11'].values.cpu().numpy(), ndi.convolve(z.real, k3))
            # the input is left as it was
            np.testing.assert_array_equal(ds['C12'].values.cpu().numpy(), z)
