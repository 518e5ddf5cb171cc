"""Merge the traffic.json files of several tools/collect_traffic.sh runs (groups of workload keys, one gpurun call
each) into one:  python tools/merge_traffic.py out.json in1.json in2.json ...   The kernel sources must be the same."""
import json
import sys

out, ins = sys.argv[1], sys.argv[2:]
tabs = [json.load(open(f)) for f in ins]
shas = {t['csrc_sha'] for t in tabs}
assert len(shas) == 1, 'the runs are of different kernel sources: %r' % shas
merged = dict(tabs[0])
merged['workloads'] = {}
for t in tabs:
    merged['workloads'].update(t['workloads'])
    if t.get('errors'):
        merged.setdefault('errors', {}).update(t['errors'])
json.dump(merged, open(out, 'w'), indent=1)
print(sorted(merged['workloads']))
