import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'exp'); os.makedirs(OUT, exist_ok=True)
import torch
so = os.path.join(OUT, 'probe.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-Rpass-analysis=kernel-resource-usage',
                       os.path.join(ROOT, 'tools', 'probe.hip'), '-o', so], stderr=open(os.path.join(OUT,'probe_build.log'),'w'))
os.system("grep -E 'Function Name|VGPRs:|Occupancy' %s | sed 's/.*remark: //; s/\\[-Rpass.*//' | paste - - - " % os.path.join(OUT,'probe_build.log'))
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
buf = torch.randn(nbytes // 4 + 4 * k * 8192, device=dev)
p = C.c_void_p(buf.data_ptr())
for pad in (0, 64):
    st = npix + pad
    for w in (1, 2, 4):
        ms = timeit(lambda: L.probe_retain(p, C.c_int64(npix), C.c_int64(st), C.c_int64(k * st), w, C.c_void_p(out.data_ptr()), stream))
        print('retain pad=%d width=%d floats: %.3f ms  %.0f GB/s' % (pad, w, ms, nbytes / ms / 1e6))
    for tch in (1, 2, 4):
        ms = timeit(lambda: L.probe_pipe(p, C.c_int64(npix), k, C.c_int64(st), C.c_int64(k * st), tch, C.c_void_p(out.data_ptr()), stream))
        print('pipe   pad=%d tch=%d: %.3f ms  %.0f GB/s' % (pad, tch, ms, nbytes / ms / 1e6))
