"""CPU tests of the host side: the C ABI loads and exports what include/nd_amd.h declares, argument
validation that needs no GPU, the footprint builder, the xr_lite container, the Algorithm API
(nd/tests/test_algorithm.py re-stated) and the chunk/halo arithmetic."""
import ctypes
import inspect
import os
import re
from collections import OrderedDict

import numpy as np
import pytest

from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI
def _declared_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'nd_amd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(nd_amd_[a-z0-9_]+)\s*\(', hdr)))


def test_library_exports_every_declared_symbol():
    from nd_amd import _lib, build
    lib = build.build()
    L = ctypes.CDLL(lib)
    declared = _declared_symbols()
    assert declared, 'no prototypes found in include/nd_amd.h'
    for s in declared:
        assert hasattr(L, s), 'libnd_amd.so does not export %s' % s
    assert sorted(_lib.SYMBOLS) == declared
    assert _lib.lib().nd_amd_abi_version() == 1


def test_argument_validation_without_gpu():
    from nd_amd import _lib
    L = _lib.lib()
    # bad dtype / negative shape are rejected before any HIP call
    rc = L.nd_amd_omnibus_c2(None, None, None, None, 7, 4, 4, 4, 4, 1, 16, 1, 0.5, None, None, None,
                             None, 0, None)
    assert rc == _lib.EINVAL
    assert b'dtype' in L.nd_amd_last_error()
    rc = L.nd_amd_omnibus_c2(None, None, None, None, 0, -1, 4, 4, 4, 1, 16, 1, 0.5, None, None, None,
                             None, 0, None)
    assert rc == _lib.EINVAL
    # empty raster: nothing to do
    rc = L.nd_amd_omnibus_c2(None, None, None, None, 0, 0, 4, 4, 4, 1, 16, 1, 0.5, None, None, None,
                             None, 0, None)
    assert rc == _lib.OK
    with pytest.raises(_lib.NdAmdError):
        _lib.check(_lib.EINVAL)
    mn = ctypes.c_size_t(0)
    rec = L.nd_amd_omnibus_c2_workspace_bytes(0, 4096, 4096, 24, ctypes.byref(mn))
    assert rec >= mn.value > 4096 * 4096 * 4


def test_ops_refuse_cpu_tensors():
    import torch
    from nd_amd import kernels
    t = torch.zeros((3, 4, 5))
    with pytest.raises(ValueError, match='ROCm device'):
        kernels.change_detection(t, t, t, t, alpha=0.5)
    with pytest.raises(ValueError, match='ROCm device'):
        kernels.convolve(t, np.ones((1, 3, 3)))


# ------------------------------------------------------------------ footprint
def _correlate_numpy(a, offs, w, mode='reflect'):
    """Straightforward evaluation of the footprint with numpy.pad (reflect = scipy 'reflect')."""
    pad = int(np.abs(offs).max()) if len(offs) else 0
    ap = np.pad(a, pad, mode='symmetric')
    out = np.zeros(a.shape, np.float64)
    for o, wt in zip(offs, w):
        sl = tuple(slice(pad + o[d], pad + o[d] + a.shape[d]) for d in range(a.ndim))
        out = out + wt * ap[sl]
    return out


@pytest.mark.parametrize('kshape', [(3, 3), (5, 4), (2, 2), (1, 7), (6, 1)])
def test_footprint_matches_scipy(kshape):
    import scipy.ndimage as ndi
    from nd_amd import kernels
    rng = np.random.default_rng(5)
    a = rng.normal(size=(11, 13))
    k = rng.normal(size=kshape)
    k.flat[1 % k.size] = 0.0
    offs, w = kernels.footprint(k)
    assert len(w) == np.count_nonzero(k)
    np.testing.assert_allclose(_correlate_numpy(a, offs, w), ndi.convolve(a, k), rtol=1e-12, atol=1e-12)
    offs_c, w_c = kernels.footprint(k, convolution=False)
    np.testing.assert_allclose(_correlate_numpy(a, offs_c, w_c), ndi.correlate(a, k), rtol=1e-12, atol=1e-12)
    with pytest.raises(ValueError, match='invalid origin'):
        kernels.footprint(k, origin=9)


# ------------------------------------------------------------------ xr_lite + io
def test_xr_lite_basics():
    from nd_amd import xr_lite
    ds = synth.lite_test_dataset()
    assert list(ds.data_vars) == ['C11', 'C12__im', 'C12__re', 'C22']
    assert ds.dims == OrderedDict([('time', 10), ('x', 20), ('y', 20)])
    sub = ds.isel(time=slice(2, 5))
    assert sub['C11'].shape == (20, 20, 3) and len(sub.coords['time']) == 3
    da = ds[['C11', 'C22']].to_array().transpose('y', 'x', 'time', 'variable')
    assert da.shape == (20, 20, 10, 2)
    back = xr_lite.expand_variables(da)
    assert back['C22'].equals(ds['C22'])
    cat = xr_lite.concat([ds.isel(y=slice(0, 7)), ds.isel(y=slice(7, None))], dim='y')
    assert cat.equals(ds)
    deep = ds.copy(deep=True)
    deep['C11'].values[0, 0, 0] += 1
    assert not deep.equals(ds)


def test_complex_convention():
    """nd/tests/test_convert.py: C12 <-> C12__re / C12__im"""
    from nd_amd.io import assemble_complex, disassemble_complex
    ds = synth.lite_test_dataset()
    c = assemble_complex(ds)
    assert set(c.data_vars) == {'C11', 'C22', 'C12'}
    np.testing.assert_array_equal(c['C12'].values, ds['C12__re'].values + 1j * ds['C12__im'].values)
    d = disassemble_complex(c)
    assert set(d.data_vars) == {'C11', 'C22', 'C12__re', 'C12__im'}
    np.testing.assert_array_equal(d['C12__im'].values, ds['C12__im'].values)
    assert 'C12' in c                       # not modified in place
    disassemble_complex(c, inplace=True)
    assert 'C12' not in c and 'C12__re' in c


# ------------------------------------------------------------------ Algorithm API
def test_wrap_algorithm_and_parallelize():
    """nd/tests/test_algorithm.py:35-88"""
    from nd_amd import xr_lite
    from nd_amd.algorithm import Algorithm, parallelize, wrap_algorithm

    class DummyAlgorithm(Algorithm):
        """test docstring"""

        def __init__(self, value, *args, **kwargs):
            self.value = value

        def apply(self, ds):
            """Apply dummy algorithm."""
            return ds + self.value

    class ParallelDummyAlgorithm(Algorithm):
        """test docstring"""

        def __init__(self, value, *args, **kwargs):
            self.value = value

        @parallelize
        def apply(self, ds):
            """Apply dummy algorithm."""
            return ds + self.value

    da = xr_lite.DataArray(np.random.default_rng(0).normal(size=(20, 20, 10)), ('y', 'x', 'time'), name='v')
    wrapper = wrap_algorithm(DummyAlgorithm, 'wrapper_name')
    assert DummyAlgorithm(0.1).apply(da).equals(wrapper(da, 0.1))
    assert wrapper.__name__ == 'wrapper_name'
    assert wrapper.__doc__ == ('Wrapper for :class:`%s.DummyAlgorithm`.\n\n' % DummyAlgorithm.__module__
                               + DummyAlgorithm.__doc__)
    assert list(inspect.signature(wrapper).parameters) == ['ds', 'value', 'args', 'kwargs']

    class MissingApply(Algorithm):
        def __init__(self):
            pass
    with pytest.raises(TypeError, match='abstract'):
        MissingApply()
    with pytest.raises(ValueError):
        wrap_algorithm(int)

    algo = ParallelDummyAlgorithm(3)
    ref = algo.apply(da)
    for njobs in (-1, 1, 2):
        assert ref.equals(algo.apply(da, njobs=njobs))
    assert 'njobs' in inspect.signature(ParallelDummyAlgorithm.apply).parameters


def test_split_merge_roundtrip():
    """xr_split / xr_merge halo arithmetic (nd/utils.py:288-340, nd/tests/test_utils.py:139-147)."""
    from nd_amd import _adapter
    ds = synth.lite_test_dataset()
    for chunks, buffer in [(2, 0), (3, 2), (4, 1), (7, 2)]:
        parts = list(_adapter.xr_split(ds, 'y', chunks, buffer))
        assert len(parts) == chunks
        merged = _adapter.xr_merge(parts, 'y', buffer)
        assert merged.equals(ds)
    assert _adapter.split_bounds(20, 3, 2) == [(0, 9), (5, 16), (12, 20)]


def test_filter_host_attributes():
    from nd_amd.filters import BoxcarFilter, ConvolutionFilter, GaussianFilter, NLMeansFilter
    b = BoxcarFilter(dims=('y', 'x'), w=5)
    assert b.kernel.shape == (5, 5) and b.kernel.dtype == np.float64 and b.kernel[0, 0] == 1 / 25
    assert b._buffer('y') == 2 and b._buffer('time') == 0
    ds = synth.lite_test_dataset(dims=OrderedDict([('y', 20), ('x', 30), ('time', 10)]))
    assert b._parallel_dimension(ds) == 'time'
    assert ConvolutionFilter(dims=('y', 'x', 'time'))._parallel_dimension(ds) == 'x'
    n = NLMeansFilter(dims=('time', 'y', 'x'), r=(0, 3, 3), f=1)
    assert list(n.r) == [0, 3, 3] and list(n.f) == [0, 1, 1]
    assert n._buffer('y') == 4 and n._buffer('time') == 0
    assert GaussianFilter(sigma=1.5)._buffer('x') == 6
    with pytest.raises(ValueError):
        NLMeansFilter(patch_distances='other')


def test_change_summaries():
    from nd_amd import xr_lite
    from nd_amd.change import change_count, first_change
    ch = np.zeros((2, 3, 5), bool)
    ch[0, 0, 2] = ch[0, 0, 4] = ch[1, 2, 1] = True
    da = xr_lite.DataArray(ch, ('y', 'x', 'time'), name='change')
    np.testing.assert_array_equal(change_count(da).values, ch.sum(axis=2))
    fc = first_change(da).values
    assert fc[0, 0] == 2 and fc[1, 2] == 1 and fc[0, 1] == -1
    assert change_count(da).dims == ('y', 'x')


def test_layout_recognition_helpers():
    """Pure stride logic of the transpose / pixel-major wrappers (no GPU needed)."""
    import torch
    from nd_amd import kernels
    t = torch.zeros((3, 5, 7))
    assert kernels._pixel_major_stride(t) == 1
    c = torch.zeros((3, 5, 7), dtype=torch.complex64)
    assert kernels._pixel_major_stride(c.real) == 2 and kernels._pixel_major_stride(c.imag) == 2
    assert kernels._pixel_major_stride(t.permute(1, 0, 2)) is None
    assert kernels._pixel_major_stride(t[:, ::2]) is None
    assert kernels._pixel_major_stride(torch.zeros((1, 1, 24))) == 1           # length-1 axes
    assert kernels._pixel_major_stride(torch.zeros((1, 4, 1))) == 1
    p = torch.zeros((7, 3, 5))
    assert kernels._planar_ok(p, 3, 5)
    assert kernels._planar_ok(torch.zeros((7, 20))[:, :15].view(7, 3, 5), 3, 5)    # padded planes
    assert not kernels._planar_ok(p.permute(0, 2, 1), 5, 3)
    # CPU tensors are declined by the wrappers themselves
    assert kernels.relayout_planar(t, p) is False
    assert kernels.change_detection_pixel_major(t, t, t, t, alpha=0.9) is None


def test_overlap_test_of_the_filter_wrappers():
    """kernels._may_overlap: conservative address-range test behind `out` aliasing `inp`."""
    import torch
    from nd_amd import kernels
    a = torch.zeros((4, 10, 12))
    assert kernels._may_overlap(a, a)
    assert kernels._may_overlap(a[1:], a[:2])
    assert not kernels._may_overlap(a[:2], a[2:])
    assert not kernels._may_overlap(a, torch.zeros_like(a))
    assert kernels._may_overlap(a[:, :, :6], a[:, :, 6:])        # interleaved columns: ranges intersect
    assert not kernels._may_overlap(a[:0], a)
    z = torch.zeros((5, 7), dtype=torch.complex64)
    assert kernels._may_overlap(z.real, z.imag)
    # the packed-split helpers decline anything that is not a CUDA tensor pair
    assert kernels.split_complex(z.real, z.imag) is None
    assert kernels.merge_complex(torch.zeros((5, 7)), torch.zeros((5, 7)), z) is False


def test_chunk_count_never_drops_rows():
    """xr_split / xr_merge lose rows when the last chunk is empty or shorter than the halo (n = 16,
    chunks = 5, halo = 1 merged to 15 rows: round-2 advisor finding); safe_chunks lowers the count
    until nothing is lost, and `parallel` checks the merged size."""
    from nd_amd import _adapter, xr_lite
    from nd_amd.algorithm import parallel
    assert _adapter.safe_chunks(16, 5, 1) == 4
    assert _adapter.safe_chunks(100, 16, 0) == 15          # ceil(100 / 16) = 7 -> 15 chunks, none empty
    assert _adapter.safe_chunks(10, 4, 3) == 2
    assert _adapter.safe_chunks(5, 1, 2) == 1
    for n in range(1, 60):
        for chunks in (1, 2, 3, 5, 8, 16):
            for halo in (0, 1, 2, 4):
                c = _adapter.safe_chunks(n, chunks, halo)
                assert 1 <= c <= chunks
                ds = xr_lite.Dataset(coords={'y': np.arange(n)})
                ds['a'] = (('y', 'x'), np.arange(n * 3, dtype=np.float32).reshape(n, 3))
                out = parallel(lambda part: part, dim='y', chunks=chunks, buffer=halo)(ds)
                np.testing.assert_array_equal(out['a'].values, ds['a'].values)


def test_integration_doc_quotes_the_example():
    """INTEGRATION.md's binding code IS examples/nd_binding.py: every marked section of the example
    appears verbatim in the document (tests/test_binding_example_gpu.py executes the example)."""
    import re
    ex = open(os.path.join(ROOT, 'examples', 'nd_binding.py')).read()
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sections = dict(re.findall(r"# --8<-- \[(\w+)\]\n(.*?)\n# --8<-- \[end\]", ex, re.S))
    assert set(sections) == {'load', 'omnibus', 'omnibus_ml', 'convolve', 'nlmeans', 'gaussian'}
    for name, code in sections.items():
        assert code.strip('\n') in doc, 'INTEGRATION.md does not quote section [%s] verbatim' % name
    # the header says what the example does with the optional tile arguments
    hdr = open(os.path.join(ROOT, 'include', 'nd_amd.h')).read()
    assert 'NULL for any of the four' in hdr
