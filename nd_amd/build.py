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


# Scratch guard.  omnibus_c2_ml_kernel waits for its LDS-DMA transfers by COUNT (s_waitcnt vmcnt(4 | 16),
# omnibus_ml.hip ml_wait_vm), and on gfx9 scratch spills and reloads count in vmcnt as well.  Loads
# return in order, so a spill or reload anywhere around stage() can only make a counted wait stricter
# (the N newest operations it leaves outstanding then include the scratch access, never an OLDER
# transfer), i.e. cost time, not correctness -- but time is what this kernel is about, and a compiler
# or flag change that pushes the step loop into scratch should not pass unnoticed.  The build reads the
# compiler's own resource report (-Rpass-analysis=kernel-resource-usage) and warns (fails under
# ND_AMD_STRICT_SCRATCH=1) if an instantiation exceeds its budget: none for the sparse form (pass A of the benchmark regime), what the fused-search
# and statistics forms are known to spill in their tails (bytes per lane) otherwise.
NO_SCRATCH = {'omnibus_ml.hip': 'omnibus_c2_ml_kernel'}
# bytes per lane by the kernel's template arguments <K, KMAX, STATS, CHAIN>, keyed (STATS, CHAIN)
SCRATCH_BUDGET = {(False, False): 0, (False, True): 96, (True, False): 256, (True, True): 384}


def _template_bools(mangled, symbol):
    """(STATS, CHAIN) of an omnibus_c2_ml_kernel instantiation from its mangled name
    (...omnibus_c2_ml_kernelILi3ELi24ELb0ELb1EEEv...), or None if the name does not parse."""
    import re
    m = re.search(re.escape(symbol) + r'I((?:L[ib]\d+E)+)E', mangled)
    if not m:
        return None
    flags = re.findall(r'Lb(\d)E', m.group(1))
    return (flags[0] == '1', flags[1] == '1') if len(flags) == 2 else None


def check_no_scratch(remarks_path, symbol, obj_path=None):
    """-> [(function, bytes per lane)] of the kernels whose mangled name contains `symbol`.  An instantiation
    over its budget is a SPEED regression (the counted waits only get stricter), so it is reported as a
    warning; ND_AMD_STRICT_SCRATCH=1 (CI) turns it into an error.  A report that is missing or older than
    the object it describes says nothing and is skipped."""
    import re
    if not os.path.exists(remarks_path) or (obj_path and os.path.exists(obj_path)
                                            and os.path.getmtime(remarks_path) + 1.0 < os.path.getmtime(obj_path)):
        return []
    found, cur = [], None
    for ln in open(remarks_path, errors='replace'):
        m = re.search(r'Function Name: (\S+)', ln)
        if m:
            cur = m.group(1)
            continue
        m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', ln)
        if m and cur is not None and symbol in cur:
            found.append((cur, int(m.group(1))))
    strict = os.environ.get('ND_AMD_STRICT_SCRATCH', '') == '1'
    if not found:
        msg = ('%s: no resource report for %s (is -Rpass-analysis=kernel-resource-usage on?)'
               % (remarks_path, symbol))
        if strict:
            raise RuntimeError(msg)
        sys.stderr.write('nd_amd.build: warning: ' + msg + '\n')
        return found
    bad = []
    for f, n in found:
        key = _template_bools(f, symbol)
        budget = SCRATCH_BUDGET.get(key, 0) if key is not None else 0
        if n > budget:
            bad.append((f, n, budget))
    if bad:
        msg = 'scratch use (bytes per lane) over budget: %s' % bad
        if strict:
            raise RuntimeError(msg)
        sys.stderr.write('nd_amd.build: warning: ' + msg + ' (speed only; ND_AMD_STRICT_SCRATCH=1 makes this an error)\n')
    return found


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
        err = None
        if os.path.basename(src) in NO_SCRATCH:
            cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
            err = open(obj + '.remarks', 'w')
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stderr=err), err))
        rebuilt.append(os.path.basename(src))
    for src, p, err in procs:
        rc = p.wait()
        if err is not None:
            err.close()
            if rc != 0 or verbose:
                sys.stderr.write(''.join(ln for ln in open(err.name, errors='replace')
                                         if 'remark:' not in ln or verbose))
        if rc != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    for src in sources():
        sym = NO_SCRATCH.get(os.path.basename(src))
        if sym:
            objp = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
            check_no_scratch(objp + '.remarks', sym, objp)
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
