#!/usr/bin/env python3
"""<dir> <commit> KEY... -> JSON on stdout: per bench workload, per nd_amd kernel, the HBM bytes per
launch from the FETCH_SIZE / WRITE_SIZE passes of tools/collect_traffic.sh.

Units and correction as in MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB;
on gfx950 FETCH_SIZE tallies 64 B per request.

THE RULE (calibrated on this pool with tools/probe_fetch.hip, a 2 GiB buffer read exactly once per
pattern; profiles/r05_fetch_calibration.json):
  * coalesced reads -- 16 B or 4 B per lane, global_load / buffer_load to registers or `... lds`
    (LDS-DMA) alike, and a lane's own 16-byte pieces at a 96-byte pitch with plain loads --
    FETCH_SIZE = 0.500 x the bytes read: requests of 128 B tallied at 64 B.  Factor 2.
  * isolated 4-byte reads (one per 64 B, 128 B, 256 B or 4 KiB): FETCH_SIZE = 64.0 B per read, i.e.
    ONE request per read.  How many bytes that request moves is not observable from this counter;
    doubling it would claim 128 B per 4-byte read.  Kernels whose reads are gathers (GATHER_KERNELS
    below: the change-point searches that pick listed pixels out of the planes) are reported with
    factor 1 = 64 B per request, and `fetch_factor` says so per kernel.
  * non-temporal 16-byte pieces at a 96-byte pitch: 0.62 (the line is dropped after its first use and
    fetched again) -- the re-fetch is real traffic, the factor stays 2.
"""
import collections
import csv
import json
import sys


# kernels whose memory reads are gathers of isolated elements: factor 1 (see THE RULE)
# (the dual-pol searches read the compact dump pass A wrote, 16 bytes per date and lane: coalesced)
GATHER_KERNELS = ('omnibus_c3_search_kernel', 'omnibus_c3_search_starts_kernel')


def fetch_factor(kernel_name, workload=''):
    """ by kernel AND workload: on pixel-major inputs (c3_pm_*) omnibus_c3_search_kernel's image
    form reads contiguous per-pixel runs with 16-byte loads -- factor 2 like every coalesced form; since round 6
    the sparse-regime searches (omnibus_c3_search_dump_kernel) read 16-byte pieces of contiguous runs
    everywhere and are not in the gather list."""
    if workload.startswith('c3_pm'):
        return 2.0
    return 1.0 if any(g in kernel_name for g in GATHER_KERNELS) else 2.0


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if 'nd_amd' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    return acc


def main():
    d, commit, keys = sys.argv[1], sys.argv[2], sys.argv[3:]
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {'commit': commit, 'csrc_sha': bench.csrc_sha(), 'unit': 'bytes per launch',
           'method': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of '
                     '`python3 bench.py --traffic-run KEY`; traffic = fetch_factor x FETCH_SIZE x 1024 + WRITE_SIZE '
                     'x 1024, fetch_factor = 2 for coalesced reads, 1 for the gather kernels (calibration: '
                     'profiles/r05_fetch_calibration.json, rule: tools/summarize_traffic.py), mean over the '
                     'launches of the run',
           'workloads': {}}
    for k in keys:
        try:
            fe = per_kernel('%s/%s_FETCH_SIZE.csv' % (d, k), 'FETCH_SIZE')
            wr = per_kernel('%s/%s_WRITE_SIZE.csv' % (d, k), 'WRITE_SIZE')
        except Exception as e:          # noqa: BLE001
            out['workloads'][k] = []
            out.setdefault('errors', {})[k] = repr(e)
            continue
        rows = []
        for name, v in fe.items():
            f = sum(v) / len(v) * 1024.0
            w = wr.get(name, [0.0])
            w = sum(w) / len(w) * 1024.0
            ff = fetch_factor(name, k)
            rows.append({'kernel': name, 'launches': len(v), 'fetch_size_bytes': f, 'fetch_factor': ff,
                         'fetch_bytes': ff * f, 'write_bytes': w, 'traffic_bytes': ff * f + w})
        out['workloads'][k] = sorted(rows, key=lambda r: -r['traffic_bytes'])
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
