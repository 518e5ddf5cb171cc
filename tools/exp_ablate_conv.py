"""Tuning experiment (GPU box): ablations of the tiled convolution kernel."""
import os, subprocess, sys, json, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'exp'); os.makedirs(OUT, exist_ok=True)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-shared']
def build(name, patches):
    d = os.path.join(OUT, 'cv_' + name); shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, 'nd_amd', 'csrc'), d, ignore=shutil.ignore_patterns('_build'))
    os.makedirs(os.path.join(OUT, 'include'), exist_ok=True)
    shutil.copy(os.path.join(ROOT, 'include', 'nd_amd.h'), os.path.join(OUT, 'include', 'nd_amd.h'))
    hp = os.path.join(d, 'common.hpp')
    h = open(hp).read().replace('../../include/nd_amd.h', os.path.join(OUT, 'include', 'nd_amd.h')); open(hp, 'w').write(h)
    p = os.path.join(d, 'correlate.hip'); s = open(p).read()
    for old, new in patches:
        assert old in s, (name, old[:50]); s = s.replace(old, new)
    open(p, 'w').write(s)
    so = os.path.join(OUT, 'lib_cv_%s.so' % name)
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', so] + [os.path.join(d, f) for f in os.listdir(d) if f.endswith('.hip')])
    return so
V = {
    'base': [],
    'nocompute': [("    for (int r = 0; r < kOY + kh - 1; ++r) {\n        double v[kOX + KW - 1];", "    for (int r = 0; r < (a.kh > 1000 ? kOY + kh - 1 : 1); ++r) {\n        double v[kOX + KW - 1];")],
    'noloads': [("                buf[i] = plane[(int64_t)ymap[r] * a.sin_y + xmap[c]];", "                buf[i] = (T)(ymap[r] + xmap[c]);")],
    'nostore': [("        if (y < a.ny) {\n            T *orow", "        if (y < a.ny && a.kh > 1000) {\n            T *orow")],
}
for name in (sys.argv[1:] or list(V)):
    so = build(name, V[name])
    for w in ('3', '5'):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_filters.py'), '--what', 'boxcar', '--w', w, '--steps', '10'], env=dict(os.environ, ND_AMD_LIB=so), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        print(name, 'w=' + w, round(json.loads(line[-1])['ms'], 3) if line else r.stderr[-300:]); sys.stdout.flush()
