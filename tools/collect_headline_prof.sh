#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline bench command only (no counter passes):
#   bash tools/collect_headline_prof.sh gpurun_out/<dir>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-extra --cpu-rows 0 > $OUT/stats.log 2>&1
F=$(ls $OUT/stats/*/p_kernel_stats.csv $OUT/stats/p_kernel_stats.csv 2>/dev/null | head -1)
grep "nd_amd\|Name" "$F" | cut -c1-200 > $OUT/kernel_stats.csv
grep '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_line_under_rocprof.json
rm -rf $OUT/stats
cat $OUT/kernel_stats.csv
