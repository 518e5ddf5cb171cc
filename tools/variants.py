"""Timing / traffic variants of single translation units, built HERE (no GPU) as _variants/lib_<name>.so
(other objects come from nd_amd/csrc/_build) and selected on the GPU box with ND_AMD_LIB.

    python tools/variants.py build [names...]

A variant = (file, [(old text, new text), ...], [extra compiler flags]).  """
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, '_variants')

VARIANTS = {
    # ---- round 5 (the ones that won are in the sources; the table stays as the record of what was tried) ----
    # pm_temporal (C11 / C22 pieces with plain instead of non-temporal loads): adopted, 1.86 -> 1.49 ms
    # pm_direct_c12 (C12 straight into registers, no LDS image): 1.49 against 1.45 ms -- no gain, code removed
    # mlw_* (wave-private form of the fused multilooking kernel): 2.30 - 2.69 ms against 2.04 ms -- lost, code removed
    # ml_pf* (L2 prefetch by the idle waves, -DND_ML_PREFETCH=n): 2.14 / 3.27 ms against 2.03 / 3.13 -- lost
    # ---- round 6: the time-split pass A of the full-pol test (omnibus_c3_retain_kernel) ----
    # timing only, NOT a correct screen: no per-date PSD check, no exponent tracking -- what the checks cost
    'c3_nocheck': ('omnibus_c3.hip', [
        ("            bad = bad | !((dmin > (T)0) & (mmin >= (T)0) & (det > (T)0));\n", ""),
        ("            const int e = __builtin_amdgcn_frexp_exp(prod);\n            emin = e < emin ? e : emin;\n"
         "            emax = e > emax ? e : emax;\n", ""),
    ], []),
    # plain instead of non-temporal loads in the time-split pass A
    'c3_plain_loads': ('omnibus_c3.hip', [
        ("            v[tt][c] = __builtin_nontemporal_load(\n                reinterpret_cast<const T *>(reinterpret_cast<const char *>(g.pl[c] + uo) + lo));\n",
         "            v[tt][c] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(g.pl[c] + uo) + lo);\n"),
    ], []),
    # XCD-aware order: the blocks one XCD receives (every eighth) walk a contiguous eighth of the raster
    'c3_xcd': ('omnibus_c3.hip', [
        ("    const int w = wv % NW, pg = wv / NW;\n    const int64_t b = blockIdx.x;\n",
         "    const int w = wv % NW, pg = wv / NW;\n    const int64_t nb8 = (int64_t)gridDim.x / 8;\n"
         "    const int64_t b = (int64_t)blockIdx.x < 8 * nb8 ? ((int64_t)blockIdx.x % 8) * nb8 + (int64_t)blockIdx.x / 8 : (int64_t)blockIdx.x;\n"),
    ], []),
    # headline pass A (omnibus_c2_retain_kernel, EXACT): the 96 loads issued from date 0 / 6 / 12 / 18 on
    # (blockIdx.x & 3, wave-uniform: four straight runs behind a scalar switch); the fold stays in time order
    'c2_issue_rotation': ('omnibus.hip', [
        ('''#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const unsigned soff = (unsigned)t * sstep;
            v[t][0] = buffer_load<T>(r11, voff, soff);
            v[t][1] = buffer_load<T>(r12r, voff, soff);
            v[t][2] = buffer_load<T>(r12i, voff, soff);
            v[t][3] = buffer_load<T>(r22, voff, soff);
        }
''', '''#define ND_ISSUE_FROM(ROT)                                                   \\
    _Pragma("unroll") for (int u = 0; u < KMAX; ++u) {                       \\
        constexpr int dummy = 0; (void)dummy;                                \\
        const int t = (u + (ROT)) % KMAX;                                    \\
        const unsigned soff = (unsigned)t * sstep;                           \\
        v[t][0] = buffer_load<T>(r11, voff, soff);                           \\
        v[t][1] = buffer_load<T>(r12r, voff, soff);                          \\
        v[t][2] = buffer_load<T>(r12i, voff, soff);                          \\
        v[t][3] = buffer_load<T>(r22, voff, soff);                           \\
    }
        switch ((int)(blockIdx.x & 3)) {
        case 0: ND_ISSUE_FROM(0) break;
        case 1: ND_ISSUE_FROM(KMAX / 4) break;
        case 2: ND_ISSUE_FROM(KMAX / 2) break;
        default: ND_ISSUE_FROM(3 * KMAX / 4) break;
        }
#undef ND_ISSUE_FROM
''', 'first'),
    ], []),
    # where the cost of the z / P rasters goes (timing only): the chi-square pair skipped
    'stats_nochisq': ('omnibus.hip', [
        ("chisq_pair<1>(zd, 4 * (k - 1), g.e.lgam, P1, P2);", "P1[0] = P2[0] = zd[0] * 1e-3;", 'all'),
    ], []),
    # full-pol search on the dump: chunks of eight instead of four dates (traffic / time of partially used lines)
    'c3_search_cd8': ('omnibus_c3.hip', [
        ("    constexpr int CD = 4;                                // dates per chunk", "    constexpr int CD = 8;                                // dates per chunk"),
        ('''                    ND_C3_PICK(2)
                default:
                    ND_C3_PICK(3)
''', '''                    ND_C3_PICK(2)
                    ND_C3_PICK(3)
                    ND_C3_PICK(4)
                    ND_C3_PICK(5)
                    ND_C3_PICK(6)
                default:
                    ND_C3_PICK(7)
'''),
    ], []),
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
        for patch in patches:
            old, new = patch[0], patch[1]
            if len(patch) > 2 and patch[2] in ('first', 'all'):   # the first of several occurrences / every one
                assert s.count(old) >= 1, (name, old[:70])
                s = s.replace(old, new, 1) if patch[2] == 'first' else s.replace(old, new)
                continue
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
