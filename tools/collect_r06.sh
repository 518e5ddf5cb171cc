#!/bin/bash
# Round-6 profile collection (run on the GPU box): writes everything under gpurun_out/<dir>.
#   bash tools/collect_r06.sh gpurun_out/r06prof <commit>
# 1. rocprofv3 --kernel-trace --stats of the headline bench command + separate --pmc FETCH_SIZE / WRITE_SIZE
#    passes of the same command -> omnibus_rocprof.txt
# 2. HBM traffic of every bench workload's kernels (tools/collect_traffic.sh) -> traffic.json, copied to
#    profiles/r06_traffic.json on the box so that the bench runs below report it
# 3. the bench line as the driver runs it, and the --extras run -> bench_line.json, bench_detail.json,
#    bench_extras.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; COMMIT=${2:-unknown}; mkdir -p $OUT
CMD="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5"
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- $CMD > $OUT/stats.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-rows 0 --no-secondary > $OUT/fetch.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-rows 0 --no-secondary > $OUT/write.log 2>&1
python3 $R/tools/summarize_prof.py $OUT/stats $OUT/fetch $OUT/write $OUT/omnibus_rocprof.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5   (the driver's command, secondary block included; PMC passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE with --kernel-trace, --steps 5 --warmup 1); commit $COMMIT" > /dev/null
cp $(find $OUT/stats -name 'p_kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/stats $OUT/fetch $OUT/write
echo "headline profiled"
# (the traffic of the other workloads: tools/collect_traffic.sh in groups of keys -- a gpurun call ends after 20 minutes --
#  merged by tools/merge_traffic.py into profiles/r06_traffic.json, which bench.py reads)
cat $OUT/omnibus_rocprof.txt | head -40
