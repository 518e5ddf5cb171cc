"""
nd_amd/build.py -- compile the HIP library in-tree:  nd_amd/libnd_amd.so

    python -m nd_amd.build [--force] [--verbose]

One hipcc invocation per translation unit (object files under nd_amd/csrc/_build,
git-ignored), then one link.  gfx950 only.  -ffp-contract=off is deliberate: the
kernels reproduce the reference's rounding points, and the few places that want a
fused multiply-add call fma() explicitly.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_build')
LIB = os.path.join(HERE, 'libnd_amd.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'

FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fno-fast-math', '-Wall', '-Wno-unused-function']


# per-translation-unit additions.  nlmeans.hip: the SLP vectoriser pairs the scalar additions of the
# cross-lane row sums (nlmeans_patch2_kernel) into v_pk_add_f32, which cannot take a DPP operand, so
# every wave-shifted value then costs a v_mov_b32_dpp of its own; the kernel's packed arithmetic is
# written with vector types and does not depend on that pass (config 3, signed mode: 54.8 -> 51.7 ms)
# omnibus.hip: the same pass packs pairs of the scalar float32 products / sums of the determinants
# into v_pk_mul_f32 / v_pk_add_f32, whose operands must sit in aligned register pairs: two to four
# v_mov per packed instruction to assemble and take apart the pairs (45 of the 140 vector
# instructions of one date of the streaming search were moves).  Without it: streaming search
# 2.155 -> 2.089 ms, pass A 1.117 -> 1.082 ms (24 x 4096^2, tools/exp_ablate.py noslp).
PER_FILE = {'nlmeans.hip': ['-fno-slp-vectorize'] + os.environ.get('ND_AMD_NLM_FLAGS', '').split(),
            'omnibus.hip': ['-fno-slp-vectorize'] + os.environ.get('ND_AMD_OMNI_FLAGS', '').split(),
            # (omnibus_c3.hip: the pass moved 15 values per date between registers to pair scalar operations
            #  of the 3 x 3 determinants; without it the streaming search needs 133 instead of 155 registers)
            'omnibus_c3.hip': ['-fno-slp-vectorize'],
            'omnibus_ml.hip': ['-fno-slp-vectorize']}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp')]
    hdr.append(os.path.join(HERE, '..', 'include', 'nd_amd.h'))
    hdr.append(os.path.abspath(__file__))
    return hdr


def build(force=False, verbose=False, extra_flags=(), out=None, objdir=None):
    global OBJ, LIB
    if out is not None:
        LIB_ = out
        OBJ_ = objdir or (out + '.obj')
        force = True
    else:
        LIB_, OBJ_ = LIB, OBJ
    return _build(force, verbose, extra_flags, LIB_, OBJ_)


def _build(force, verbose, extra_flags, LIB, OBJ):
    os.makedirs(OBJ, exist_ok=True)
    dep_mtime = max(os.path.getmtime(p) for p in _deps())
    objs = []
    rebuilt = []
    procs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) >= max(os.path.getmtime(src), dep_mtime)):
            continue
        cmd = [HIPCC] + FLAGS + PER_FILE.get(os.path.basename(src), []) + list(extra_flags) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        rebuilt.append(os.path.basename(src))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    linked = False
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        linked = True
    # say what this call actually did: objects are reused by mtime, so "nothing compiled" is a
    # legitimate outcome that the build log should show
    print('nd_amd.build: hipcc --offload-arch=%s rebuilt=%s linked=%s lib=%s'
          % (ARCH, rebuilt, linked, os.path.relpath(LIB)))
    return LIB


if __name__ == '__main__':
    lib = build(force='--force' in sys.argv, verbose='--verbose' in sys.argv)
    print(lib)
