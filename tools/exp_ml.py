"""Timing variants of omnibus_ml.hip (macro switches), built HERE and timed on the GPU box.

    python tools/exp_ml.py build name=-DFLAG[,-DFLAG2] ...   # in the container: _variants/libml_<name>.so
    python tools/exp_ml.py run [names...]                    # on the GPU box

Only omnibus_ml.hip is recompiled; the other objects come from nd_amd/csrc/_build.  Most variants
compute wrong maps on purpose (timing probes)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, '_variants')


def build(specs):
    from nd_amd import build as B
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition('=')
        obj = os.path.join(VDIR, 'omnibus_ml_%s.o' % name)
        cmd = [B.HIPCC] + B.FLAGS + ['-fno-slp-vectorize'] + [f for f in flags.split(',') if f] + \
              ['-c', os.path.join(B.CSRC, 'omnibus_ml.hip'), '-o', obj]
        procs.append((name, obj, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    for name, obj, pr in procs:
        assert pr.wait() == 0, name
        objs = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith('.o') and f != 'omnibus_ml.o']
        so = os.path.join(VDIR, 'libml_%s.so' % name)
        subprocess.check_call([B.HIPCC, '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', so, obj] + objs)
        os.remove(obj)
        print('built', so)


def child():
    import torch
    from nd_amd import _lib, kernels, synth
    dev = torch.device('cuda:0')
    k = int(os.environ.get('EXP_K', '24'))
    st = synth.wishart_c2_stack(k, 4096, 4096, looks=1, seed=1234, device=dev, change_frac=0.01)
    for ml in [int(x) for x in os.environ.get('EXP_MLS', '3,5').split(',')]:
        for alpha in [float(x) for x in os.environ.get('EXP_ALPHAS', '0.99,0.01').split(',')]:
            f = lambda: kernels.change_detection_multilooked(st[0], st[1], st[2], st[3], alpha=alpha, ml=ml)
            for _ in range(3):
                f()
            _lib.timing_enable(128)
            for _ in range(8):
                f()
            torch.cuda.synchronize()
            kt = _lib.timing_collect()
            _lib.timing_enable(0)
            by = {}
            for n, ms in kt:
                by.setdefault(n, []).append(ms)
            print(json.dumps({'ml': ml, 'alpha': alpha, **{n: round(sum(v) / len(v), 4) for n, v in by.items()}}))


if __name__ == '__main__':
    cmd = sys.argv[1]
    if cmd == 'build':
        build(sys.argv[2:])
    elif cmd == 'child':
        child()
    else:
        names = sys.argv[2:] or sorted(f[6:-3] for f in os.listdir(VDIR) if f.startswith('libml_'))
        for name in names:
            so = os.path.join(VDIR, 'libml_%s.so' % name)
            env = dict(os.environ, ND_AMD_LIB=so)
            r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
            for line in r.stdout.strip().splitlines() or [r.stderr[-800:]]:
                print(name, line)
            sys.stdout.flush()
