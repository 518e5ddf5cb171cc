"""Tuning experiment (GPU box): per-wave iteration counts and durations of omnibus pass B.  Builds a
patched copy of omnibus.hip whose search kernel records, per block, the number of loop iterations
and the elapsed wall clock in the unused tail of its shard's list segment, then prints the
distribution."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

def child():
    import ctypes as C
    import numpy as np, torch
    from nd_amd import _lib, synth
    dev = torch.device('cuda:0')
    k, ny, nx = 24, 4096, 4096
    st = synth.wishart_c2_stack(k, ny, nx, seed=1234, device=dev, change_frac=0.01)
    L = _lib.lib()
    mb = C.c_size_t(0)
    nbytes = L.nd_amd_omnibus_c2_workspace_bytes(0, ny, nx, k, C.byref(mb))
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    change = torch.empty((ny, nx, k), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(2):
        rc = L.nd_amd_omnibus_c2(p(st[0]), p(st[1]), p(st[2]), p(st[3]), 0, ny, nx, k, nx, 1, st.stride(1),
                                 9, C.c_double(0.99), p(change), None, None, p(ws), C.c_size_t(nbytes),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    w32 = ws.view(torch.int32).cpu().numpy().view(np.uint32)
    # layout: [128 counters x 32 words][table][list 128 x seg]
    counts = w32[0:128 * 32:32]
    off_tab = 128 * 32 * 4
    al = lambda v: (v + 255) // 256 * 256
    off_idx = off_tab + al((k + 1) * 64)
    npix = ny * nx
    nb256 = -(-npix // 256) + ny
    seg = 2 * nb256 + (2 * ny + 256) * 4 + 256
    lst = w32[off_idx // 4: off_idx // 4 + 128 * seg].reshape(128, seg)
    rec = lst[:, seg - 256:].reshape(128, 64, 4)      # [shard][lblock][iters, cycles, batches, -]
    it = rec[:, :, 0].astype(np.int64); cy = rec[:, :, 1].astype(np.int64); nb = rec[:, :, 2]
    act = nb > 0
    print('candidates', int(counts.sum()), 'active blocks', int(act.sum()))
    print('iterations per active block: mean %.1f  p50 %d  p90 %d  p99 %d  max %d' % (
        it[act].mean(), *np.percentile(it[act], [50, 90, 99]).astype(int), it[act].max()))
    print('clock ticks (100 MHz) per active block: mean %.1f p50 %d p90 %d p99 %d max %d' % (
        cy[act].mean(), *np.percentile(cy[act], [50, 90, 99]).astype(int), cy[act].max()))
    print('ticks per iteration (active): %.2f' % (cy[act].sum() / it[act].sum()))
    print('ticks of empty blocks: mean %.1f max %d' % (cy[~act].mean(), cy[~act].max()))

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(); sys.exit(0)
    import exp_ablate
    patches = [
        ("    for (uint32_t base = lblock * 64u; base < n; base += nlblock * 64u) {",
         "    const unsigned long long dbg_t0 = wall_clock64();\n    unsigned dbg_it = 0, dbg_nb = 0;\n"
         "    for (uint32_t base = lblock * 64u; base < n; base += nlblock * 64u) {\n        ++dbg_nb;"),
        ("        while (__any(!done)) {\n            if (!done) {\n                load_step(A, t);",
         "        while (__any(!done)) {\n            ++dbg_it;\n            if (!done) {\n                load_step(A, t);"),
    ]
    # the record goes to the last 256 words of the shard's list segment (never reached by the list)
    src = open(os.path.join(ROOT, 'nd_amd', 'csrc', 'omnibus.hip')).read()
    tail = "                        done = true;                       // :241-242\n                    }\n                }\n            }\n        }\n    }\n}"
    assert tail in src
    patches.append((tail, tail[:-2] + "    if (lane == 0 && lblock < 64) {\n        uint32_t *r = const_cast<uint32_t *>(s.flag_idx) + (size_t)shard * s.seg + s.seg - 256 + lblock * 4;\n"
                    "        r[0] = dbg_it; r[1] = (uint32_t)(wall_clock64() - dbg_t0); r[2] = dbg_nb;\n    }\n}"))
    so = exp_ablate.build_variant('b_trace', patches)
    env = dict(os.environ, ND_AMD_LIB=so)
    r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
    print(r.stdout[-3000:], r.stderr[-2000:])
