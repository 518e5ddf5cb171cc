#!/usr/bin/env python3
"""Summarise rocprofv3 output (kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes) for the
nd_amd kernels into a text file under profiles/.

    python tools/summarize_prof.py gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write \
        profiles/r01_omnibus_rocprof.txt "<command that was profiled>"

PMC correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE counts 64 B per 128-B
request, i.e. half the bytes of a coalesced streaming read; it is doubled here.  The factor is
checked against this kernel's own known byte count (every input byte is read exactly once).
Counter unit: KiB.
"""
import collections
import csv
import glob
import sys


def kernel_stats(d):
    f = (glob.glob(d + '/*/*_kernel_stats.csv') + glob.glob(d + '/*kernel_stats.csv'))[0]
    return [r for r in csv.DictReader(open(f)) if 'nd_amd' in r['Name']]


def kernel_trace_split(d):
    """{kernel name: (n_gated, avg_us_gated, n_working, avg_us_working)} for the kernels whose launches are of two
    kinds -- the device-side gate (omni_gate_skip: both forms of a call are launched, the unfavoured one returns at
    once) makes launches of a few microseconds of kernels that otherwise take a millisecond, and one average over
    both says nothing.  From the per-dispatch trace; a launch counts as gated below a tenth of the longest."""
    fs = glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*kernel_trace.csv')
    if not fs:
        return {}
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'nd_amd' in r['Kernel_Name']:
            acc[r['Kernel_Name']].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    out = {}
    for k, v in acc.items():
        cut = 0.1 * max(v)
        lo, hi = [x for x in v if x < cut], [x for x in v if x >= cut]
        if lo and hi and max(v) > 100.0:
            out[k] = (len(lo), sum(lo) / len(lo), len(hi), sum(hi) / len(hi))
    return out


def pmc(d, counter):
    f = (glob.glob(d + '/*/*_counter_collection.csv') + glob.glob(d + '/*counter_collection.csv'))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'nd_amd' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    dstats, dfetch, dwrite, out, cmd = sys.argv[1:6]
    lines = ['rocprofv3 summary for the nd_amd kernels (MI355X, gfx950)', 'command: ' + cmd, '']
    lines.append('--kernel-trace --stats (per kernel):')
    lines.append('%-86s %6s %12s %12s %12s' % ('kernel', 'calls', 'avg_us', 'min_us', 'max_us'))
    for r in kernel_stats(dstats):
        lines.append('%-86s %6s %12.1f %12.1f %12.1f' % (
            r['Name'].split('(')[0][-86:], r['Calls'], float(r['AverageNs']) / 1e3,
            float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
    split = kernel_trace_split(dstats)
    if split:
        lines.append('')
        lines.append('kernels launched both as the working form and as the form the device-side gate sends back at once')
        lines.append('(the secondary block runs the headline stack at alpha = 0.01, where the sparse pass A is launched and returns):')
        lines.append('%-86s %6s %12s %8s %12s' % ('kernel', 'calls', 'avg_us', 'gated', 'avg_us'))
        for k, (nl, al, nh, ah) in split.items():
            lines.append('%-86s %6d %12.1f %8d %12.1f' % (k.split('(')[0][-86:], nh, ah, nl, al))
    lines.append('')
    lines.append('--pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), averages per launch:')
    fe, wr = pmc(dfetch, 'FETCH_SIZE'), pmc(dwrite, 'WRITE_SIZE')
    for k in fe:
        f_kib, nf = fe[k]
        w_kib, nw = wr.get(k, (float('nan'), 0))
        rd = 2.0 * f_kib * 1024
        wt = w_kib * 1024
        lines.append('%s' % k.split('(')[0][-86:])
        lines.append('    FETCH_SIZE %.0f KiB (n=%d) -> read bytes (x2 gfx950 correction) %.4e' % (f_kib, nf, rd))
        lines.append('    WRITE_SIZE %.0f KiB (n=%d) -> write bytes %.4e' % (w_kib, nw, wt))
        lines.append('    HBM traffic per launch %.4e B' % (rd + wt))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
