"""Ablation variants of omnibus.hip, built HERE (no GPU needed) and timed on the GPU box.

    python tools/exp_variants.py build [names...]     # in the container: _variants/lib_<name>.so
    python tools/exp_variants.py run [names...]       # on the GPU box: one JSON line per variant and alpha

A variant is a list of (text, replacement) patches of nd_amd/csrc/omnibus.hip; only that unit is
recompiled, the other objects come from nd_amd/csrc/_build.  The variants are timing probes: most of
them compute wrong maps on purpose."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, '_variants')

VARIANTS = {
    'base': [],
    # streaming search: no stores of the change map in dense waves
    's_nostore': [("        if (in) {\n            uint8_t *res = wob + (int64_t)lane * k;\n            if ((k & 3) == 0 && ((uintptr_t)res & 3) == 0) {\n                uint32_t *w = reinterpret_cast<uint32_t *>(res);\n                for (int q = 0; q < (k >> 2); ++q)\n                    w[q] = (mask_nibble(mask, q) * 0x00204081u) & 0x01010101u;",
                   "        if (in && g.k < 0) {\n            uint8_t *res = wob + (int64_t)lane * k;\n            if ((k & 3) == 0 && ((uintptr_t)res & 3) == 0) {\n                uint32_t *w = reinterpret_cast<uint32_t *>(res);\n                for (int q = 0; q < (k >> 2); ++q)\n                    w[q] = (mask_nibble(mask, q) * 0x00204081u) & 0x01010101u;")],
    # ... no deep searches (a pixel that needs one just stops)
    's_nodeep': [("                        } else if (DP & bit) {\n                            deep = true;", "                        } else if (DP & bit) {\n                            done = true;")],
    # ... no walk at all
    's_nowalk': [("            while (__any(!done)) {\n                bool deep = false;", "            done = true;\n            while (__any(!done)) {\n                bool deep = false;")],
    # ... no 2- / 3-date tests
    's_nomarg': [("        if constexpr (sizeof(T) == 4) {                     // marginal tests over 2 and 3 dates", "        if constexpr (false) {} else if constexpr (false) {                     // marginal tests over 2 and 3 dates"),
                 ("            prod12 = prod2;\n            det1 = ds;\n        } else {\n            T s11 = q.a + d1.a", "            prod12 = prod2;\n            det1 = ds;\n        } else if constexpr (false) {\n            T s11 = q.a + d1.a")],
}


def build(names):
    from nd_amd import build as B
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for name in names:
        d = os.path.join(VDIR, 'src_' + name)
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(B.CSRC, d, ignore=shutil.ignore_patterns('_build'))
        hp = os.path.join(d, 'common.hpp')
        h = open(hp).read().replace('../../include/nd_amd.h', os.path.join(ROOT, 'include', 'nd_amd.h'))
        open(hp, 'w').write(h)
        p = os.path.join(d, 'omnibus.hip')
        s = open(p).read()
        for old, new in VARIANTS[name]:
            assert old in s, (name, old[:70])
            s = s.replace(old, new)
        open(p, 'w').write(s)
        obj = os.path.join(VDIR, 'omnibus_%s.o' % name)
        cmd = [B.HIPCC] + B.FLAGS + B.PER_FILE['omnibus.hip'] + ['-c', p, '-o', obj]
        procs.append((name, obj, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    for name, obj, pr in procs:
        assert pr.wait() == 0, name
        objs = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith('.o') and f != 'omnibus.o']
        so = os.path.join(VDIR, 'lib_%s.so' % name)
        subprocess.check_call([B.HIPCC, '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', so, obj] + objs)
        os.remove(obj)
        shutil.rmtree(os.path.join(VDIR, 'src_' + name))
        print('built', so)


def child():
    import torch
    from nd_amd import _lib, kernels, synth
    dev = torch.device('cuda:0')
    st = synth.wishart_c2_stack(24, 4096, 4096, seed=1234, device=dev, change_frac=0.01)
    for alpha in [float(x) for x in os.environ.get('EXP_ALPHAS', '1e-4,0.01').split(',')]:
        for _ in range(3):
            kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
        _lib.timing_enable(128)
        for _ in range(10):
            kernels.change_detection(st[0], st[1], st[2], st[3], alpha=alpha, n=9)
        torch.cuda.synchronize()
        kt = _lib.timing_collect()
        _lib.timing_enable(0)
        by = {}
        for n, ms in kt:
            by.setdefault(n, []).append(ms)
        print(json.dumps({'alpha': alpha, **{n: round(sum(v) / len(v), 4) for n, v in by.items()}}))


if __name__ == '__main__':
    cmd = sys.argv[1]
    names = sys.argv[2:] or list(VARIANTS)
    if cmd == 'build':
        build(names)
    elif cmd == 'child':
        child()
    else:
        for name in names:
            so = os.path.join(VDIR, 'lib_%s.so' % name)
            env = dict(os.environ, ND_AMD_LIB=so)
            r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
            for line in r.stdout.strip().splitlines() or [r.stderr[-800:]]:
                print(name, line)
            sys.stdout.flush()
