"""On-disk tiling (nd_amd/tiling.py, the role of nd/tiling.py without NetCDF): tile -> map -> merge
equals the whole-raster result, an interrupted run resumes, `.part` leftovers are never mistaken for
tiles (nd/tiling.py:95-100; reference tests: nd/tests/test_tiling.py:62-127)."""
import os

import numpy as np
import pytest

from nd_amd import tiling, xr_lite


def _dataset(ny=37, nx=50, k=5, seed=0):
    rng = np.random.default_rng(seed)
    ds = xr_lite.Dataset(coords={'y': np.arange(ny) * 10.0, 'x': np.arange(nx) * 10.0 + 5,
                                 'time': np.arange(k)}, attrs={'crs': 'EPSG:32650', 'n': 3})
    ds['C11'] = (('y', 'x', 'time'), rng.gamma(4.0, 0.25, (ny, nx, k)).astype(np.float32))
    ds['C12'] = (('y', 'x', 'time'), (rng.normal(size=(ny, nx, k)) + 1j * rng.normal(size=(ny, nx, k))).astype(np.complex64))
    ds['mask'] = (('y', 'x'), rng.random((ny, nx)) < 0.5)
    return ds


def _box(ds, w=5):
    """a windowed function with scipy.ndimage arithmetic on the host (the oracle's boxcar)"""
    from oracle import oracle as O
    out = xr_lite.Dataset(coords=ds.coords, attrs=ds.attrs)
    kern = np.ones((w, w, 1)) / float(w * w)
    a = np.ascontiguousarray(ds['C11'].values)
    out['C11'] = (('y', 'x', 'time'), O.convolve(a, kern))
    return out


def test_tile_roundtrip_and_names(tmp_path):
    ds = _dataset()
    paths = tiling.tile(ds, str(tmp_path / 'tiles'), chunks={'y': 16, 'x': 20}, buffer={'y': 2, 'x': 3})
    assert len(paths) == 3 * 3
    assert os.path.basename(paths[0]) == 'part.y_0_18.x_0_23.envi'
    assert os.path.basename(paths[-1]) == 'part.y_30_37.x_37_50.envi'
    t = tiling.open_tile(paths[4])                       # the middle tile: y 14..34, x 17..43
    assert t['C11'].shape == (20, 26, 5) and t['C12'].dtype == np.complex64
    assert t['mask'].dtype == np.bool_ and t['mask'].shape == (20, 26)
    np.testing.assert_array_equal(t['C11'].values, ds['C11'].values[14:34, 17:43])
    np.testing.assert_array_equal(np.asarray(t.coords['y']), np.asarray(ds.coords['y'])[14:34])
    # ENVI header of a variable: (time*y? no:) the last two axes are lines x samples
    hdr = open(os.path.join(paths[4], 'C11.hdr')).read()
    assert 'samples = 5' in hdr and 'lines = 26' in hdr and 'bands = 20' in hdr and 'data type = 4' in hdr
    merged = tiling.auto_merge(paths)
    for v in ('C11', 'C12', 'mask'):
        np.testing.assert_array_equal(merged[v].values, ds[v].values)
    np.testing.assert_array_equal(np.asarray(merged.coords['x']), np.asarray(ds.coords['x']))
    assert merged.attrs['crs'] == 'EPSG:32650'
    # a second call skips what exists
    before = {p: os.path.getmtime(os.path.join(p, 'tile.json')) for p in paths}
    tiling.tile(ds, str(tmp_path / 'tiles'), chunks={'y': 16, 'x': 20}, buffer={'y': 2, 'x': 3})
    assert before == {p: os.path.getmtime(os.path.join(p, 'tile.json')) for p in paths}


def test_map_over_tiles_equals_whole_and_resumes(tmp_path, oracle):
    ds = _dataset(ny=41, nx=33, k=3, seed=2)
    want = _box(ds)['C11'].values
    tiles = tiling.tile(ds, str(tmp_path / 'in'), chunks={'y': 12}, buffer=2)      # halo = w // 2
    calls = []

    def flaky(t, fail_at=None):
        calls.append(1)
        if fail_at is not None and len(calls) == fail_at:
            raise RuntimeError('power cut')
        return _box(t)

    out_dir = str(tmp_path / 'out')
    with pytest.raises(RuntimeError):
        tiling.map_over_tiles(tiles, flaky, kwargs={'fail_at': 3}, path=out_dir, suffix='_box')
    done = sorted(os.listdir(out_dir))
    assert len(done) == 2 and all(d.endswith('_box.envi') for d in done)
    # simulate a result that died while being written
    os.makedirs(os.path.join(out_dir, 'part.y_22_38_box.envi.part'))
    calls.clear()
    merged = tiling.map_over_tiles(tiles, flaky, path=out_dir, suffix='_box')
    assert len(calls) == len(tiles) - 2                      # the two finished tiles were kept
    np.testing.assert_array_equal(merged['C11'].values, want)
    assert merged['C11'].dims == ('y', 'x', 'time')
    # overwrite=True recomputes everything; merge=False returns the paths
    calls.clear()
    paths = tiling.map_over_tiles(tiles, flaky, path=out_dir, suffix='_box', merge=False, overwrite=True)
    assert len(calls) == len(tiles) and all(os.path.isdir(p) for p in paths)
    np.testing.assert_array_equal(tiling.auto_merge(os.path.join(out_dir, '*_box.envi'))['C11'].values, want)
    # too small a buffer shows up as a difference (the halo matters)
    tiles0 = tiling.tile(ds, str(tmp_path / 'in0'), chunks={'y': 12}, buffer=0)
    bad = tiling.map_over_tiles(tiles0, _box, path=str(tmp_path / 'out0'))
    assert not np.array_equal(bad['C11'].values, want)


def test_errors(tmp_path):
    ds = _dataset()
    f = tmp_path / 'afile'
    f.write_text('x')
    with pytest.raises(ValueError, match='cannot be a file'):
        tiling.tile(ds, str(f), chunks={'y': 10})
    with pytest.raises(ValueError, match='no dimension'):
        tiling.tile(ds, str(tmp_path / 't'), chunks={'z': 10})
    with pytest.raises(ValueError, match='No files found'):
        tiling.auto_merge(str(tmp_path / 'nothing*'))
    paths = tiling.tile(ds, str(tmp_path / 't'), chunks={'y': 20})
    with pytest.raises(ValueError, match='overwrite its input'):
        tiling.map_over_tiles(paths, lambda t: t)


@pytest.mark.gpu
def test_tiled_pipeline_on_the_gpu_equals_whole_raster(tmp_path, oracle, device):
    """tile -> BoxcarFilter -> OmnibusTest per tile on the GPU -> merge == the whole raster at once."""
    from nd_amd.change import OmnibusTest
    from nd_amd.filters import BoxcarFilter
    from tests import synth
    planes = synth.omnibus_stack(seed=5, k=8, ny=70, nx=90, dtype=np.float32, change_frac=0.2)
    ds = xr_lite.Dataset()
    for v, p in zip(('C11', 'C12__re', 'C12__im', 'C22'), planes):
        ds[v] = (('y', 'x', 'time'), np.ascontiguousarray(np.moveaxis(p, 0, -1)))

    def pipeline(t):
        return OmnibusTest(n=9 * 9, alpha=0.9).apply(BoxcarFilter(w=3).apply(t))

    want = pipeline(ds).values
    tiles = tiling.tile(ds, str(tmp_path / 'in'), chunks={'y': 25, 'x': 40}, buffer=1)
    merged = tiling.map_over_tiles(tiles, pipeline, path=str(tmp_path / 'out'))
    np.testing.assert_array_equal(merged['change'].values, want)
    assert want.any()


def test_map_over_tiles_twice_next_to_the_inputs(tmp_path, oracle):
    """Results written beside the inputs (path=None) match the same glob on the next call: they
    must be recognised as results, not filtered a second time (round-2 advisor finding)."""
    ds = _dataset(ny=30, nx=21, k=2, seed=5)
    want = _box(ds)['C11'].values
    d = str(tmp_path / 'tiles')
    tiling.tile(ds, d, chunks={'y': 10}, buffer=2)
    pattern = os.path.join(d, '*.envi')
    first = tiling.map_over_tiles(pattern, _box, suffix='_f')
    np.testing.assert_array_equal(first['C11'].values, want)
    n_after_first = len(os.listdir(d))
    calls = []
    second = tiling.map_over_tiles(pattern, lambda t: calls.append(1) or _box(t), suffix='_f')
    assert calls == []                                        # every result already existed
    assert len(os.listdir(d)) == n_after_first                # no *_f_f tiles
    np.testing.assert_array_equal(second['C11'].values, want)
