#!/bin/bash
# Round-3 profile collection (run on the GPU box): writes everything under gpurun_out/<dir>.
#   bash tools/collect_profiles.sh gpurun_out/r03prof <commit>
# 1. rocprofv3 --kernel-trace --stats of the headline bench command, and separate --pmc FETCH_SIZE /
#    WRITE_SIZE passes of the same command (HBM traffic, the guide's recipe) -> omnibus_rocprof.txt
# 2. HBM traffic of every bench workload's kernels (tools/collect_traffic.sh) -> traffic.json
# 3. SQ counters of the headline's kernels (tools/pmc_cmd.sh) -> headline_pmc.txt
# 4. the bench line itself, with the traffic file in place -> bench_line.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; COMMIT=${2:-unknown}; mkdir -p $OUT
CMD="python3 $R/bench.py --steps 20 --warmup 5 --no-extra --cpu-rows 0"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-extra --cpu-rows 0 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-extra --cpu-rows 0 > $OUT/write.log 2>&1
python3 $R/tools/summarize_prof.py $OUT/stats $OUT/fetch $OUT/write $OUT/omnibus_rocprof.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-extra --cpu-rows 0   (PMC passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE with --kernel-trace, --steps 5 --warmup 1); commit $COMMIT" > /dev/null
cp $OUT/stats/*/p_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null || cp $OUT/stats/p_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/stats $OUT/fetch $OUT/write
echo "headline profiled"
bash $R/tools/collect_traffic.sh $1/traffic $COMMIT > $OUT/traffic.log 2>&1
cp $OUT/traffic/traffic.json $OUT/traffic.json
mkdir -p $R/profiles && cp $OUT/traffic.json $R/profiles/r03_traffic.json     # bench.py reads it from there
echo "traffic collected"
bash $R/tools/pmc_cmd.sh $1/pmc bench.py --steps 3 --warmup 1 --no-extra --cpu-rows 0 > /dev/null 2>&1
cp $OUT/pmc/summary.txt $OUT/headline_pmc.txt 2>/dev/null
(cd $R && python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err)
ls $OUT
cat $OUT/omnibus_rocprof.txt
