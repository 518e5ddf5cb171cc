#!/usr/bin/env python3
"""
oracle/build_ref.py -- build the REAL reference for the part of the path that
compiles from its own source in this image.  TEST INFRASTRUCTURE ONLY.

What is built
-------------
`/root/reference/nd/_filters.pyx` (the reference's non-local-means kernel,
`_pixelwise_nlmeans_3d` / `find_weight`) is translated by the Cython that is in
this image and compiled with gcc, *from where it lies* under /root/reference,
unmodified.  This mirrors the reference's own setup.py:78-82
(`Extension("nd._filters", ["nd/_filters.pyx"], extra_compile_args=['-O3'])`),
without running the reference's build system.  Output:

    oracle/_ref/nd/__init__.py              (empty, ours)
    oracle/_ref/nd/_filters.<abi>.so        (git-ignored; not gpurun-ignored)

The generated C file is written to a temporary directory outside the repo and
deleted; no reference source is copied into the repo.

What is NOT built
-----------------
`/root/reference/nd/_change.pyx` needs `cython_gsl` (CythonGSL) and libgsl,
neither of which is in this image, and the reference's own setup.py:60-95 skips
the module when GSL is missing.  No stand-in for GSL is written: the omnibus
path is treated as unbuildable here (DESIGN.md "Oracle").

`scipy.ndimage.convolve` (the boxcar/convolution arithmetic, called from
nd/filters.py:256-267) is a third-party dependency that IS installed here, so
it is used directly as the reference for that path (tests/golden/make_golden.py).

Run:  python3 oracle/build_ref.py        (no-op when /root/reference is absent)
"""
import glob
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('ND_REFERENCE', '/root/reference')
OUT = os.path.join(HERE, '_ref', 'nd')


def have_ref():
    return bool(glob.glob(os.path.join(OUT, '_filters*.so')))


def main():
    pyx = os.path.join(REF, 'nd', '_filters.pyx')
    if not os.path.exists(pyx):
        print('build_ref: %s not present -- nothing to do' % pyx)
        return 0
    try:
        import Cython  # noqa: F401
        import numpy
    except ImportError as e:
        print('build_ref: %s -- cannot build the reference here' % e)
        return 0
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, '__init__.py'), 'a').close()
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    so = os.path.join(OUT, '_filters' + ext)
    tmp = tempfile.mkdtemp(prefix='nd_ref_build_')
    try:
        cfile = os.path.join(tmp, '_filters.c')
        subprocess.check_call([sys.executable, '-m', 'cython', '-3', pyx, '-o', cfile])
        inc = ['-I' + sysconfig.get_paths()['include'], '-I' + numpy.get_include()]
        subprocess.check_call(
            ['gcc', '-O3', '-fPIC', '-shared', '-fwrapv', '-fno-strict-aliasing',
             '-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION', '-w']
            + inc + [cfile, '-o', so, '-lm'])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print('build_ref: built', so)
    return 0


if __name__ == '__main__':
    sys.exit(main())
