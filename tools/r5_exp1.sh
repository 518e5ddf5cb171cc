#!/bin/bash
# round 5, GPU call 2: FETCH_SIZE calibration probe; pixel-major dense form with temporal C11 / C22
# loads (time + FETCH); C3 at alpha = 0.99 with the streaming fused search.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r5_exp1; mkdir -p $OUT
# 1. calibration
hipcc --offload-arch=gfx950 -O3 -o $OUT/probe_fetch $R/tools/probe_fetch.hip || exit 1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o p --output-format csv -- $OUT/probe_fetch > $OUT/probe_stdout.txt 2> $OUT/probe_stderr.txt
F=$(find $OUT/pf -name 'p_counter_collection.csv' | head -1)
python3 $R/tools/summarize_fetch_probe.py "$F" $OUT/probe_stdout.txt > $OUT/fetch_calibration.json
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/pfs -o p --output-format csv -- $OUT/probe_fetch > /dev/null 2>&1
S=$(find $OUT/pfs -name 'p_kernel_stats.csv' | head -1); cp "$S" $OUT/probe_kernel_stats.csv
rm -rf $OUT/pf $OUT/pfs $OUT/probe_fetch
echo "calibration done"
# 2. pixel-major dense form: base vs temporal loads
cd $R
for V in base pm_temporal; do
  if [ $V = base ]; then unset ND_AMD_LIB; else export ND_AMD_LIB=$R/_variants/lib_$V.so; fi
  timeout -k 5 200 python3 tools/exp_bench_extra.py pm_a0.01 pm_a0.99 > $OUT/pm_$V.txt 2>&1
  ( cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmf_$V -o p --output-format csv -- python3 $R/bench.py --traffic-run pm_a0.01 > $OUT/pm_${V}_fetch.log 2>&1 )
  F=$(find /tmp/pmf_$V -name 'p_counter_collection.csv' | head -1)
  python3 - "$F" > $OUT/pm_${V}_fetch.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'FETCH_SIZE' and 'nd_amd' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:70]].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, len(v), 'FETCH_SIZE KiB mean', sum(v) / len(v), 'x2 GB', 2 * sum(v) / len(v) * 1024 / 1e9)
PY
  echo "pm $V done"
done
unset ND_AMD_LIB
# 3. C3 at 0.99: sparse design vs the streaming fused search
timeout -k 5 200 python3 tools/exp_bench_extra.py c3_a0.99 > $OUT/c3_sparse.txt 2>&1
ND_AMD_C3_FUSED_ALPHA=2 timeout -k 5 200 python3 tools/exp_bench_extra.py c3_a0.99 > $OUT/c3_fused.txt 2>&1
echo "c3 done"
tail -3 $OUT/pm_base.txt $OUT/pm_pm_temporal.txt $OUT/pm_base_fetch.txt $OUT/pm_pm_temporal_fetch.txt $OUT/c3_sparse.txt $OUT/c3_fused.txt
