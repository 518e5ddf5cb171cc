"""Timing / traffic variants of single translation units, built HERE (no GPU) as _variants/lib_<name>.so
(other objects come from nd_amd/csrc/_build) and selected on the GPU box with ND_AMD_LIB.

    python tools/variants_r5.py build [names...]

A variant = (file, [(old text, new text), ...], [extra compiler flags]).  Round 5's experiments; the ones that won are in the
sources, the table stays as the record of what was tried."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, '_variants')

VARIANTS = {
    # pixel-major dense form: C11 / C22 pieces with plain (temporal) loads instead of non-temporal ones
    # pm_temporal (C11 / C22 pieces with plain instead of non-temporal loads): adopted in round 5,
    # 1.86 -> 1.49 ms, 8.19 -> 6.44 GB of traffic (gpurun_out/r5_exp1)
    # pm_direct_c12 (C12 straight into registers as well, no LDS image; -DND_PM_DIRECT_C12, code since removed):
    # 1.49 against 1.45 ms -- no gain (gpurun_out/r5_exp2)
    # mlw_* (the wave-private form of the fused multilooking kernel, tools/experiments/omnibus_mlw.hip, as
    # omnibus_mlw.hip in csrc at the time): 6 waves x 2 slots 2.69 ms, 4 waves x 3 slots 2.30 ms, two waves
    # per SIMD 2.67 ms, against 2.04 ms for the block form on the same box (gpurun_out/r5_ml)
    # the block form: L2 prefetch by the idle waves, steps ahead of the transfers
    # non-local means as of round 4 (A/B of the stream3 kernel's skipped edge planes on one box)
    'nlm_r04': ('nlmeans.hip', 'git:d879eb2', []),
    'ml_trace': ('omnibus_ml.hip', [], ['-DND_ML_TRACE']),
    'ml_pf0': ('omnibus_ml.hip', [], ['-DND_ML_PREFETCH=0']),
    'ml_pf1': ('omnibus_ml.hip', [], ['-DND_ML_PREFETCH=1']),
    'ml_pf2': ('omnibus_ml.hip', [], ['-DND_ML_PREFETCH=2']),
    'ml_pf5': ('omnibus_ml.hip', [], ['-DND_ML_PREFETCH=5']),
    'ml_pf8': ('omnibus_ml.hip', [], ['-DND_ML_PREFETCH=8']),
}


def build(names):
    from nd_amd import build as B
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for name in names:
        fname, patches, flags = VARIANTS[name]
        d = os.path.join(VDIR, 'src_' + name)
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(B.CSRC, d, ignore=shutil.ignore_patterns('_build'))
        hp = os.path.join(d, 'common.hpp')
        h = open(hp).read().replace('../../include/nd_amd.h', os.path.join(ROOT, 'include', 'nd_amd.h'))
        open(hp, 'w').write(h)
        p = os.path.join(d, fname)
        s = open(p).read()
        if isinstance(patches, str) and patches.startswith('git:'):
            s = subprocess.check_output(['git', 'show', '%s:nd_amd/csrc/%s' % (patches[4:], fname)], cwd=ROOT).decode()
            patches = []
        for old, new in patches:
            assert s.count(old) == 1, (name, s.count(old), old[:70])
            s = s.replace(old, new)
        open(p, 'w').write(s)
        obj = os.path.join(VDIR, '%s_%s.o' % (fname[:-4], name))
        cmd = [B.HIPCC] + B.FLAGS + B.PER_FILE.get(fname, []) + flags + ['-c', p, '-o', obj]
        procs.append((name, fname, obj, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    for name, fname, obj, pr in procs:
        assert pr.wait() == 0, name
        objs = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ))
                if f.endswith('.o') and f != fname[:-4] + '.o']
        so = os.path.join(VDIR, 'lib_%s.so' % name)
        subprocess.check_call([B.HIPCC, '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', so, obj] + objs)
        os.remove(obj)
        shutil.rmtree(os.path.join(VDIR, 'src_' + name))
        print('built', so)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:] or list(VARIANTS))
