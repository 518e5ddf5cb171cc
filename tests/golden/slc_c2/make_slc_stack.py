"""k-date dual-pol series from the reference's bundled single-date C2 raster (slc_c2.npz).

    planes = slc_stack(k=24, looks=9, seed=1)      # [C11, C12re, C12im, C22], each (k, 206, 500) float32

Every pixel's bundled matrix S = [[c11, c12], [conj c12, c22]] is taken as the true covariance; a date
is the mean of `looks` outer products s s^H with s = L z, S = L L^H (2 x 2 Cholesky), z ~ CN(0, I).
Pixels whose bundled matrix is zero stay exactly zero (the scene's nodata margin); rank-deficient
single-look matrices (c11 c22 = |c12|^2) give rank-deficient samples scaled by chi-square noise.
Rows 60-140 x columns 150-350 get a x4 power step from date k // 2 on."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def bundled():
    g = np.load(os.path.join(HERE, 'slc_c2.npz'))
    return g['C11'], g['C12__re'], g['C12__im'], g['C22']


def slc_stack(k=24, looks=9, seed=1, step=4.0):
    c11, c12r, c12i, c22 = (a.astype(np.float64) for a in bundled())
    ny, nx = c11.shape
    rng = np.random.default_rng(seed)
    l11 = np.sqrt(np.maximum(c11, 0.0))
    safe = np.where(l11 > 0, l11, 1.0)
    l21 = (c12r - 1j * c12i) / safe                      # conj(c12) / l11
    l21 = np.where(l11 > 0, l21, 0.0)
    l22 = np.sqrt(np.maximum(c22 - np.abs(l21) ** 2, 0.0))
    out = [np.empty((k, ny, nx), np.float32) for _ in range(4)]
    for t in range(k):
        z = (rng.normal(size=(2, looks, ny, nx)) + 1j * rng.normal(size=(2, looks, ny, nx))) / np.sqrt(2.0)
        s1 = l11 * z[0]
        s2 = l21 * z[0] + l22 * z[1]
        g = np.ones((ny, nx))
        if t >= k // 2:
            g[60:140, 150:350] = step
        c = (s1 * np.conj(s2)).mean(axis=0)
        out[0][t] = (np.abs(s1) ** 2).mean(axis=0) * g
        out[1][t] = c.real * g
        out[2][t] = c.imag * g
        out[3][t] = (np.abs(s2) ** 2).mean(axis=0) * g
    return out


if __name__ == '__main__':
    p = slc_stack()
    print([a.shape for a in p], [float(a.max()) for a in p], float((p[0] == 0).mean()))
