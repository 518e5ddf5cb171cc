#!/bin/bash
# headline step under ND_AMD_SLABS = 1 / 2 / 3 / 4 / 8 (row slabs: pass B of one under pass A of the next)
out=${1:-gpurun_out/slabs}
mkdir -p $out
for s in 1 2 4 8 3 1 4; do
  ND_AMD_SLABS=$s timeout -k 10 200 python3 bench.py --no-extra --cpu-rows 0 --steps 40 --warmup 10 > $out/slabs_$s.json 2> $out/slabs_$s.err || exit 1
  python3 - $out/slabs_$s.json $s <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print('slabs', sys.argv[2], 'value %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'step_ms', {k: round(v, 4) for k, v in d['step_ms'].items() if k != 'note'},
      'passA/launch %.4f x %s' % (r['kernel_ms'], r.get('launches_per_step')), 'frac %.3f' % r['frac'], 'check', d.get('matches_oracle_whole_raster', d.get('check')), flush=True)
P
done
