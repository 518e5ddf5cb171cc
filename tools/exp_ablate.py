"""Tuning experiment (GPU box): ablations of the omnibus kernels. Builds patched copies of
nd_amd/csrc/omnibus.hip into gpurun_out/exp and times pass A / pass B with the library's event timers."""
import os, subprocess, sys, json, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'gpurun_out', 'exp')
os.makedirs(OUT, exist_ok=True)
HIPCC = '/opt/rocm/bin/hipcc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math']

def child():
    import torch
    from nd_amd import _lib, kernels, synth
    dev = torch.device('cuda:0')
    st = synth.wishart_c2_stack(24, 4096, 4096, seed=1234, device=dev, change_frac=0.01)
    for _ in range(3):
        kernels.change_detection(st[0], st[1], st[2], st[3], alpha=float(os.environ.get('EXP_ALPHA', '0.99')), n=9)
    _lib.timing_enable(64)
    for _ in range(10):
        kernels.change_detection(st[0], st[1], st[2], st[3], alpha=float(os.environ.get('EXP_ALPHA', '0.99')), n=9)
    torch.cuda.synchronize()
    kt = _lib.timing_collect()
    by = {}
    for n, ms in kt:
        by.setdefault(n, []).append(ms)
    print(json.dumps({n: round(sum(v) / len(v), 4) for n, v in by.items()}))

def build_variant(name, patches, extra=()):
    d = os.path.join(OUT, 'src_' + name)
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, 'nd_amd', 'csrc'), d, ignore=shutil.ignore_patterns('_build'))
    # the sources include ../../include/nd_amd.h
    os.makedirs(os.path.join(OUT, 'include'), exist_ok=True)
    shutil.copy(os.path.join(ROOT, 'include', 'nd_amd.h'), os.path.join(OUT, 'include', 'nd_amd.h'))
    p = os.path.join(d, 'omnibus.hip')
    s = open(p).read()
    for old, new in patches:
        assert old in s, (name, old[:60])
        s = s.replace(old, new)
    s = s.replace('#include "common.hpp"', '#include "common.hpp"')
    open(p, 'w').write(s)
    hp = os.path.join(d, 'common.hpp')
    h = open(hp).read().replace('../../include/nd_amd.h', os.path.join(OUT, 'include', 'nd_amd.h'))
    open(hp, 'w').write(h)
    so = os.path.join(OUT, 'lib_%s.so' % name)
    srcs = [os.path.join(d, f) for f in os.listdir(d) if f.endswith('.hip')]
    subprocess.check_call([HIPCC] + FLAGS + list(extra) + ['-shared', '-o', so] + srcs)
    return so

# name -> [(text in omnibus.hip, replacement)]; each patch must still match the current source
VARIANTS = {
    'base': [],
    # pass B: stop after the first sweep / never run marginal tests / never evaluate any test
    'b_onesweep': [("                        if (l >= k - 1) {\n                            done = true;                   // :256",
                    "                        if (true) {\n                            done = true;                   // :256")],
    'b_nofirst': [("const bool need = (jj >= 2) && (fire_at < 0 || last);", "const bool need = (jj >= 2) && last;")],
    'b_notests': [("const bool need = (jj >= 2) && (fire_at < 0 || last);", "const bool need = false && (jj >= 2) && (fire_at < 0 || last);")],
    # dense kernel: occupancy hints
    'd_w3': [("__launch_bounds__(64) omnibus_c2_dense_kernel", "__launch_bounds__(64, 3) omnibus_c2_dense_kernel")],
    'd_w4': [("__launch_bounds__(64) omnibus_c2_dense_kernel", "__launch_bounds__(64, 4) omnibus_c2_dense_kernel")],
    # pass A: product of determinants as a float32 sum of logs (timing only, not exact)
    'a_logsum': [('    // ---- fold in time order ----\n    Accum<T> A;\n    A.reset();\n#pragma unroll\n    for (int t = 0; t < KMAX; ++t)\n        if (EXACT || t < k) A.step(v[t][0], v[t][1], v[t][2], v[t][3]);\n\n    bool flag;\n    if (STATS) {', '    // ---- fold in time order ----\n    Accum<T> A;\n    A.reset();\n    float lsum = 0.f;\n    if (STATS) {\n#pragma unroll\n    for (int t = 0; t < KMAX; ++t)\n        if (EXACT || t < k) A.step(v[t][0], v[t][1], v[t][2], v[t][3]);\n    } else {\n#pragma unroll\n    for (int t = 0; t < KMAX; ++t)\n        if (EXACT || t < k) {\n            const T a_ = v[t][0], b_ = v[t][1], c_ = v[t][2], d_ = v[t][3];\n            const T det = (a_ * d_) - ((b_ * b_) + (c_ * c_));\n            lsum = lsum + __log2f(fabsf((float)det));\n            A.s11 = A.s11 + a_; A.s12r = A.s12r + b_; A.s12i = A.s12i + c_; A.s22 = A.s22 + d_;\n        }\n    A.prod = exp2((double)lsum);\n    }\n\n    bool flag;\n    if (STATS) {')],
    # pass A: each XCD (block index mod 8) / each of 32 groups walks its own contiguous part of the raster
    'a_xcd8': [('omnibus_c2_retain_kernel(const OmniGlobalArgs<T> g, const OmniTab tab)\n{\n    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int64_t b = blockIdx.x;', 'omnibus_c2_retain_kernel(const OmniGlobalArgs<T> g, const OmniTab tab)\n{\n    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int64_t nb_ = gridDim.x, per_ = nb_ >> 3;\n    const int64_t b = (nb_ & 7) ? (int64_t)blockIdx.x : ((int64_t)(blockIdx.x & 7) * per_ + (blockIdx.x >> 3));')],
    'a_grp32': [('omnibus_c2_retain_kernel(const OmniGlobalArgs<T> g, const OmniTab tab)\n{\n    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int64_t b = blockIdx.x;', 'omnibus_c2_retain_kernel(const OmniGlobalArgs<T> g, const OmniTab tab)\n{\n    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int64_t nb_ = gridDim.x, per_ = nb_ >> 5;\n    const int64_t b = (nb_ & 31) ? (int64_t)blockIdx.x : ((int64_t)(blockIdx.x & 31) * per_ + (blockIdx.x >> 5));')],
    # pass A block size
    't128': [("#define ND_RETAIN_THREADS 256", "#define ND_RETAIN_THREADS 128")],
}

# name -> extra hipcc flags
EXTRA = {'noslp': ['-fno-slp-vectorize']}
VARIANTS['noslp'] = []

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(); sys.exit(0)
    names = sys.argv[1:] or list(VARIANTS)
    for name in names:
        so = build_variant(name, VARIANTS[name], EXTRA.get(name, ()))
        env = dict(os.environ, ND_AMD_LIB=so)
        r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
        print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:])
        sys.stdout.flush()
