#!/usr/bin/env python3
"""<counter csv of FETCH_SIZE> <stdout of probe_fetch> -> table: FETCH_SIZE (KiB -> bytes) of every probe
kernel against the bytes it read (tools/probe_fetch.hip): the factor to multiply FETCH_SIZE by for that
access pattern.  Gathers: bytes fetched per isolated 4-byte read."""
import collections
import csv
import json
import sys


def main():
    rows = collections.defaultdict(list)
    order = []
    for r in csv.DictReader(open(sys.argv[1])):
        if r['Counter_Name'] != 'FETCH_SIZE' or 'probe_' not in r['Kernel_Name']:
            continue
        key = (r['Kernel_Name'].split('(')[0], r['Dispatch_Id'])
        if key not in order:
            order.append(key)
        rows[key].append(float(r['Counter_Value']))
    known = {}
    reads = {}
    for ln in open(sys.argv[2]):
        p = ln.split()
        if len(p) >= 2 and p[0].startswith('probe_'):
            (reads if len(p) == 3 else known)[p[0]] = int(p[1])
    gathers = ['probe_gather_4096', 'probe_gather_256', 'probe_gather_128', 'probe_gather_64']
    out = []
    gi = 0
    seen = collections.Counter()
    for name, disp in order:
        fetch = sum(rows[(name, disp)]) * 1024.0
        short = name.replace('void ', '')
        seen[short] += 1
        e = {'kernel': short, 'dispatch': int(disp), 'FETCH_SIZE_bytes': fetch}
        if 'gather' in short:
            g = gathers[gi % 4]
            gi += 1
            e['pattern'] = g
            e['reads'] = reads[g]
            e['FETCH_bytes_per_read'] = fetch / reads[g]
        else:
            base = short.split('<')[0]
            nb = known.get('probe_pitch96' if 'pitch96' in base else base)
            if nb:
                e['bytes_read'] = nb
                e['FETCH_over_bytes'] = fetch / nb
                e['factor_to_apply'] = nb / fetch if fetch else None
        out.append(e)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
