"""Pins of the omnibus oracle (CPU): the reference's own known answers, the chi-square CDF the
reference itself uses at its second call site (scipy.stats.chi2, nd/_change.pyx:123-124), mpmath,
the anchors recorded in SURVEY.md section 8c, and the committed golden vectors."""
import os

import numpy as np
import pytest

from tests import synth

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'omnibus_kat.npz')


def test_chisq_cdf_against_scipy(oracle):
    from scipy.stats import chi2
    worst = 0.0
    for nu in [4, 8, 12, 20, 40, 92, 96, 188, 400, 9, 27, 45]:
        for x in np.geomspace(1e-3, 5 * nu + 50, 300):
            a, b = oracle.cdf_chisq_P(x, nu), chi2.cdf(x, nu)
            if b > 1e-290:
                worst = max(worst, abs(a - b) / b)
    assert worst < 2e-12


def test_chisq_cdf_against_mpmath(oracle):
    import mpmath
    mpmath.mp.dps = 40
    for nu, x in [(4, 0.5), (4, 30.0), (92, 60.0), (92, 92.0), (92, 140.0), (96, 126.3), (9, 3.3),
                  (188, 250.0), (8, 1e-3)]:
        want = float(mpmath.gammainc(nu / 2.0, 0, x / 2.0, regularized=True))
        assert abs(oracle.cdf_chisq_P(x, nu) - want) <= 2e-13 * max(want, 1e-300)


def test_chisq_gsl_branches(oracle):
    """One pin per branch of gsl_cdf_gamma_P as restated in oracle/nd_oracle.c (cdf/gamma.c: B1
    x <= 0, B2 y > a -> 1 - Q, B3 -> P) and per edge value of the call sites nd/_change.pyx:147-148."""
    from scipy.stats import chi2
    P = oracle.cdf_chisq_P
    # B1: x <= 0 returns 0 before anything is evaluated (also for -inf and -0.0)
    for x in (0.0, -0.0, -1e-300, -3.0, -float('inf')):
        assert P(x, 4) == 0.0 and P(x, 92) == 0.0
    # B3: y = x / 2 <= a = nu / 2 (lower tail, the P form), B2: y > a (upper tail, 1 - Q)
    for nu in (4, 8, 48, 92, 96):                # f = 4 (j - 1) and f + 4 of the dual-pol test
        a = nu / 2.0
        for x in (1e-12, 0.5 * nu, nu * (1 - 1e-12)):          # B3, up to the switch point
            assert 2 * a >= x
            assert P(x, nu) == pytest.approx(chi2.cdf(x, nu), rel=2e-12, abs=1e-300)
        for x in (nu * (1 + 1e-12), 2.0 * nu, 10.0 * nu, 1e3 * nu):   # B2
            assert x / 2 > a
            assert P(x, nu) == pytest.approx(chi2.cdf(x, nu), rel=2e-12)
        # continuity across the switch y = a
        assert abs(P(nu * (1 + 1e-9), nu) - P(nu * (1 - 1e-9), nu)) < 1e-8    # (pdf x 2e-9 nu: no jump)
    # large finite x: Q underflows smoothly, P -> 1 exactly, never above
    for x in (1e4, 1e6, 2e6, 1e12, 1e300):
        assert P(x, 4) == 1.0 and P(x, 92) == 1.0
    # x = +inf: GSL's large-x form evaluates exp(a ln x - x - ...) = exp(inf - inf): NaN (scipy: 1)
    assert np.isnan(P(float('inf'), 4)) and np.isnan(P(float('inf'), 92))
    assert chi2.cdf(float('inf'), 4) == 1.0
    # NaN stays NaN through every comparison
    assert np.isnan(P(float('nan'), 4))


def test_chisq_edge_semantics(oracle):
    # gsl_cdf_chisq_P: 0 for x <= 0 (GSL cdf/gamma.c); NaN propagates
    assert oracle.cdf_chisq_P(0.0, 4) == 0.0
    assert oracle.cdf_chisq_P(-3.0, 4) == 0.0
    assert np.isnan(oracle.cdf_chisq_P(float('nan'), 4))
    assert np.isnan(oracle.cdf_chisq_P(float('inf'), 4))


def test_survey_scalar_anchors(oracle):
    """(p=2, k=24, n=9)."""
    rho = oracle.rho(2, 24, 9)
    assert rho == 0.9324845679012346
    assert oracle.omega2(2, 24, 9, rho) == 0.04979228465313219
    assert oracle.f_dof(2, 24, 9) == 92.0


def _kat_values():
    dims = {'y': 5, 'x': 5, 'time': 10}
    d1 = synth.reference_test_dataset(dims, [1, 0, 0, 1], 0.1)
    d2 = synth.reference_test_dataset(dims, [10, 0, 0, 10], 0.1)
    ds = {v: np.concatenate([d1[v][..., :5], d2[v][..., 5:]], axis=2) for v in d1}
    return np.stack([ds['C11'], ds['C12__re'], ds['C12__im'], ds['C22']], axis=-1)


def test_reference_known_answer(oracle):
    """nd/tests/test_change_omnibus.py:6-19: change at time index 5 for every pixel, exactly one
    change per pixel; float32 and float64 agree."""
    values = _kat_values()
    ch64, z64, P64 = oracle.change_detection(values, 0.9, 9, stats=True)
    assert ch64[:, :, 5].all()
    assert (ch64.sum(axis=2) == 1).all()
    ch32 = oracle.change_detection(values.astype(np.float32), 0.9, 9)
    np.testing.assert_array_equal(ch32, ch64)
    # anchors recorded by the survey's probe of the reference
    assert z64[0, 0] == 190.88359667561778
    assert P64[0, 0] == 1.0
    P5, _ = oracle.single_pixel_omnibus(values[0, 0, :5], 9)
    assert abs(P5 - 4.998458137782973e-4) < 1e-17
    z32 = oracle.change_detection(values.astype(np.float32), 0.9, 9, stats=True)[1]
    assert float(z32[0, 0]) == pytest.approx(190.8836212158203, abs=0)
    P5f, _ = oracle.single_pixel_omnibus(values[0, 0, :5].astype(np.float32), 9)
    assert float(P5f) == pytest.approx(4.998564254492521e-4, rel=1e-7)


def test_nan_path(oracle):
    """nd/tests/test_change_common.py:21-32: N(0,1) data, default alpha/n -> no change anywhere."""
    dn = synth.reference_test_dataset({'y': 20, 'x': 30, 'time': 10}, 0, 1)
    vn = np.stack([dn['C11'], dn['C12__re'], dn['C12__im'], dn['C22']], axis=-1)
    ch, z, P = oracle.change_detection(vn, 0.01, 1, stats=True)
    assert not ch.any()
    assert np.isnan(z[0, 0]) and np.isnan(P[0, 0])


def test_golden_vectors(oracle):
    g = np.load(GOLD)
    np.testing.assert_array_equal(g['kat_values'], _kat_values())
    for dt, tag in ((np.float64, 'f64'), (np.float32, 'f32')):
        ch, z, P = oracle.change_detection(g['kat_values'].astype(dt), 0.9, 9, stats=True)
        np.testing.assert_array_equal(ch, g['kat_change_' + tag])
        np.testing.assert_array_equal(z, g['kat_z_' + tag])
        np.testing.assert_array_equal(P, g['kat_P_' + tag])
    ch, z, P = oracle.change_detection(g['nan_values'], 0.01, 1, stats=True)
    np.testing.assert_array_equal(ch, g['nan_change'])
    for alpha in (0.9, 0.99, 0.9999):
        tag = ('%g' % alpha).replace('.', 'p')
        ch = oracle.change_detection(g['wishart_values_f32'], alpha, 9)
        np.testing.assert_array_equal(ch, g['wishart_change_' + tag])


def test_P_definition_in_double(oracle):
    """P = P1 + omega2 (P2 - P1) with P_i = chi2cdf(z; f + 4 i) (doc/change/omnibus.rst:51-60)."""
    from scipy.stats import chi2
    rng = np.random.default_rng(0)
    planes = synth.wishart_c2(rng, (14, 1, 1), 9, np.float64)
    ts = np.stack([p[:, 0, 0] for p in planes], axis=-1)
    for j in (2, 5, 14):
        P, z = oracle.single_pixel_omnibus(ts[:j], 9)
        f = oracle.f_dof(2, j, 9)
        rho = oracle.rho(2, j, 9)
        w2 = oracle.omega2(2, j, 9, rho)
        P1, P2 = chi2.cdf(z, f), chi2.cdf(z, f + 4)
        assert P == pytest.approx(P1 + w2 * (P2 - P1), rel=1e-11, abs=1e-300)
        # z = -2 rho ln Q from its definition
        dets = ts[:j, 0] * ts[:j, 3] - (ts[:j, 1] ** 2 + ts[:j, 2] ** 2)
        s = ts[:j].sum(axis=0)
        lnQ = 9 * (2 * j * np.log(j) + np.log(dets).sum() - j * np.log(s[0] * s[3] - s[1] ** 2 - s[2] ** 2))
        assert z == pytest.approx(-2 * rho * lnQ, rel=1e-10)


def test_threads_and_strides_do_not_matter(oracle):
    planes = synth.omnibus_stack(seed=2, k=9, ny=17, nx=13, dtype=np.float32, change_frac=0.2)
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    a = oracle.change_detection_planes(yxt, 0.9, 9, njobs=1)
    b = oracle.change_detection_planes(yxt, 0.9, 9, njobs=4)
    strided = [np.moveaxis(p, 0, -1) for p in planes]       # views of planar (t, y, x)
    c = oracle.change_detection_planes(strided, 0.9, 9, njobs=2)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)
    assert a.sum() > 0


def test_generic_p_equals_c2_path_for_p2(oracle):
    """The generic-p restatement with p = 2 reproduces the C2 oracle bit for bit (shared formulas)."""
    for dt in (np.float32, np.float64):
        planes = synth.omnibus_stack(seed=4, k=11, ny=14, nx=9, dtype=dt, change_frac=0.3)
        yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
        a = oracle.change_detection_planes(yxt, 0.9, 9, stats=True)
        b = oracle.change_detection_pol(yxt, 2, 0.9, 9, stats=True)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
        assert a[0].sum() > 0


def test_generic_p3_against_numpy_definition(oracle):
    """p = 3 (no reference implementation: parity unpinned).  The oracle is checked against the
    textbook statistic in float64: ln Q = n (p k ln k + sum ln det C_i - k ln det sum C_i),
    z = -2 rho ln Q, P = P1 + omega2 (P2 - P1) with chi-square CDFs from scipy."""
    from scipy.stats import chi2
    planes = synth.omnibus_stack_c3(seed=6, k=9, ny=3, nx=4, dtype=np.float64, change_frac=0.5)
    yxt = [np.ascontiguousarray(np.moveaxis(p, 0, -1)) for p in planes]
    ch, z, P = oracle.change_detection_pol(yxt, 3, 0.9, 9, stats=True)
    k = 9

    def mats(iy, ix):
        v = [p[iy, ix] for p in yxt]
        C = np.zeros((k, 3, 3), complex)
        C[:, 0, 0], C[:, 1, 1], C[:, 2, 2] = v[0], v[1], v[2]
        C[:, 0, 1] = v[3] + 1j * v[4]; C[:, 1, 0] = np.conj(C[:, 0, 1])
        C[:, 0, 2] = v[5] + 1j * v[6]; C[:, 2, 0] = np.conj(C[:, 0, 2])
        C[:, 1, 2] = v[7] + 1j * v[8]; C[:, 2, 1] = np.conj(C[:, 1, 2])
        return C
    for iy in range(3):
        for ix in range(4):
            C = mats(iy, ix)
            lnQ = 9 * (3 * k * np.log(k) + np.log(np.linalg.det(C).real).sum()
                       - k * np.log(np.linalg.det(C.sum(axis=0)).real))
            rho = oracle.rho(3, k, 9)
            w2 = oracle.omega2(3, k, 9, rho)
            f = oracle.f_dof(3, k, 9)
            assert f == 72.0
            zz = -2 * rho * lnQ
            assert z[iy, ix] == pytest.approx(zz, rel=1e-9)
            P1, P2 = chi2.cdf(zz, f), chi2.cdf(zz, f + 4)
            assert P[iy, ix] == pytest.approx(P1 + w2 * (P2 - P1), rel=1e-9, abs=1e-300)
    # half-integer a = f/2 occurs for even k: chi-square CDF with odd dof
    for nu in (9, 27, 63, 81):
        for x in (0.5, 10.0, float(nu), 3.0 * nu):
            assert oracle.cdf_chisq_P(x, nu) == pytest.approx(chi2.cdf(x, nu), rel=2e-12)
