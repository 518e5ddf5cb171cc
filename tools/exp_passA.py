"""Tuning experiment (GPU box): pass-A bandwidth under layout / build variants + raw read probes."""
import ctypes as C, os, subprocess, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'gpurun_out', 'exp')
os.makedirs(OUT, exist_ok=True)

def child(argv):
    import torch
    from nd_amd import _lib, kernels, synth
    pad = int(argv[0])
    k, ny, nx = 24, 4096, 4096
    dev = torch.device('cuda:0')
    # planar stack with the date stride padded by `pad` elements
    st = ny * nx + pad
    buf = torch.empty(4 * k * st, dtype=torch.float32, device=dev)
    src = synth.wishart_c2_stack(k, ny, nx, seed=1234, device=dev, change_frac=0.01)
    planes = []
    for v in range(4):
        pl = buf[v * k * st:(v + 1) * k * st].view(k, st)[:, :ny * nx].view(k, ny, nx)
        pl.copy_(src[v])
        planes.append(pl)
    del src
    for _ in range(3):
        kernels.change_detection(*planes, alpha=0.99, n=9)
    _lib.timing_enable(64)
    for _ in range(10):
        kernels.change_detection(*planes, alpha=0.99, n=9)
    torch.cuda.synchronize()
    kt = _lib.timing_collect()
    a = [ms for n, ms in kt if n == 'omnibus_c2_global']
    b = [ms for n, ms in kt if n == 'omnibus_c2_search']
    print(json.dumps({'pad': pad, 'passA_ms': sum(a) / len(a), 'passB_ms': sum(b) / len(b),
                      'passA_GBs': k * 16 * ny * nx / (sum(a) / len(a) * 1e-3) / 1e9}))

def probes():
    import torch
    so = os.path.join(OUT, 'probe.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC',
                           os.path.join(ROOT, 'tools', 'probe.hip'), '-o', so])
    L = C.CDLL(so)
    dev = torch.device('cuda:0')
    k, npix = 24, 4096 * 4096
    out = torch.zeros(4, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def timeit(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    nbytes = 4 * k * npix * 4
    buf = torch.randn(nbytes // 4 + 4 * k * 4096, device=dev)
    for blocks in (2048, 8192, 65536):
        ms = timeit(lambda: L.probe_read_linear(C.c_void_p(buf.data_ptr()), C.c_int64(nbytes), blocks, C.c_void_p(out.data_ptr()), stream))
        print('linear read blocks=%d: %.3f ms  %.0f GB/s' % (blocks, ms, nbytes / ms / 1e6))
    for pad in (0, 64, 1024, 4096 + 64):
        st = npix + pad
        for tch in (2, 4, 8):
            ms = timeit(lambda: L.probe_read_planes(C.c_void_p(buf.data_ptr()), C.c_int64(npix), k, C.c_int64(st), C.c_int64(k * st), tch, C.c_void_p(out.data_ptr()), stream))
            print('planes read pad=%d tch=%d: %.3f ms  %.0f GB/s' % (pad, tch, ms, nbytes / ms / 1e6))

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(sys.argv[2:])
        sys.exit(0)
    probes()
    from nd_amd import build
    variants = {'base': [], 'tch2': ['-DND_TIME_CHUNK=2'], 'tch8': ['-DND_TIME_CHUNK=8']}
    for name, flags in variants.items():
        so = os.path.join(OUT, 'libnd_amd_%s.so' % name)
        build.build(extra_flags=flags, out=so)
        for pad in ((0, 64, 1024) if name == 'base' else (0,)):
            env = dict(os.environ, ND_AMD_LIB=so)
            r = subprocess.run([sys.executable, __file__, 'child', str(pad)], env=env, capture_output=True, text=True)
            print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
