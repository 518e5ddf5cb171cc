#!/usr/bin/env python3
"""<dir> <commit> KEY... -> JSON on stdout: per bench workload, per nd_amd kernel, the HBM bytes per
launch from the FETCH_SIZE / WRITE_SIZE passes of tools/collect_traffic.sh.

Units and correction as in MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB;
on gfx950 FETCH_SIZE tallies 64 B per 128-B request of a coalesced streaming read, so it is doubled.
"""
import collections
import csv
import json
import sys


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
                     '`python3 bench.py --traffic-run KEY`; traffic = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 '
                     '(gfx950 correction of MI355X_MICROARCH.md), mean over the launches of the run',
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
            rows.append({'kernel': name, 'launches': len(v), 'fetch_bytes_x2': 2.0 * f, 'write_bytes': w,
                         'traffic_bytes': 2.0 * f + w})
        out['workloads'][k] = sorted(rows, key=lambda r: -r['traffic_bytes'])
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
