#!/bin/bash
# Round-2 profile collection (run on the GPU box): writes everything under gpurun_out/<dir>.
#   bash tools/collect_profiles.sh gpurun_out/r02prof
# 1. rocprofv3 --kernel-trace --stats of the headline bench command
# 2. separate --pmc FETCH_SIZE / WRITE_SIZE passes of the same command (HBM traffic, guide's recipe)
# 3. the same three for the dense-threshold run (bench_dense.py --alphas 0.01)
# 4. SQ counters of the dense-threshold kernels (tools/pmc_dense.sh) and of the nlmeans patch kernels
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; mkdir -p $OUT
CMD="python3 $R/bench.py --steps 20 --warmup 5 --no-extra --cpu-rows 0"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-extra --cpu-rows 0 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-extra --cpu-rows 0 > $OUT/write.log 2>&1
python3 $R/tools/summarize_prof.py $OUT/stats $OUT/fetch $OUT/write $OUT/omnibus_rocprof.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-extra --cpu-rows 0   (PMC passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE with --kernel-trace, --steps 5 --warmup 1)" > /dev/null
DC="python3 $R/tools/bench_dense.py --alphas 0.01 --steps 5 --cpu-rows 0"
rocprofv3 --kernel-trace --stats -d $OUT/dstats -o p --output-format csv -- $DC > $OUT/dstats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/dfetch -o p --output-format csv -- $DC > $OUT/dfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/dwrite -o p --output-format csv -- $DC > $OUT/dwrite.log 2>&1
python3 $R/tools/summarize_prof.py $OUT/dstats $OUT/dfetch $OUT/dwrite $OUT/omnibus_dense_rocprof.txt "rocprofv3 --kernel-trace --stats -- python3 tools/bench_dense.py --alphas 0.01 --steps 5 --cpu-rows 0 (and --pmc FETCH_SIZE / WRITE_SIZE passes)" > /dev/null
bash $R/tools/pmc_dense.sh $1/dense_pmc 0.01 > /dev/null 2>&1
bash $R/tools/pmc_nlm.sh $1/nlm_pmc > /dev/null 2>&1
# 5. kernel stats of the filters (boxcar 3x3 / 5x5, fused Gaussian) and of the tutorial pipeline
: > $OUT/filters_kernel_stats.txt
for args in "--what boxcar --w 3" "--what boxcar --w 5" "--what boxcar --w 7" "--what gaussian --sigma 1.0"; do
  rm -rf /tmp/fprof; rocprofv3 --kernel-trace --stats -d /tmp/fprof -o p --output-format csv -- python3 $R/tools/bench_filters.py $args --steps 10 > /tmp/fprof.log 2>&1
  echo "== rocprofv3 --kernel-trace --stats -- python3 tools/bench_filters.py $args --steps 10" >> $OUT/filters_kernel_stats.txt
  grep "nd_amd" /tmp/fprof/*/p_kernel_stats.csv /tmp/fprof/p_kernel_stats.csv 2>/dev/null | cut -d: -f2- | cut -c1-220 >> $OUT/filters_kernel_stats.txt
done
rm -rf /tmp/fprof; rocprofv3 --kernel-trace --stats -d /tmp/fprof -o p --output-format csv -- python3 $R/tools/bench_pipeline.py --nx 16384 --steps 3 --alpha 1e-4 > /tmp/fprof.log 2>&1
echo "== rocprofv3 --kernel-trace --stats -- python3 tools/bench_pipeline.py --nx 16384 --steps 3 --alpha 1e-4" >> $OUT/filters_kernel_stats.txt
grep "nd_amd" /tmp/fprof/*/p_kernel_stats.csv /tmp/fprof/p_kernel_stats.csv 2>/dev/null | cut -d: -f2- | cut -c1-220 >> $OUT/filters_kernel_stats.txt
bash $R/tools/pmc_nlm_window.sh $1/nlmwin > /dev/null 2>&1
(cd $R && python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err)
cp $OUT/stats/p_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
cp $OUT/dstats/p_kernel_stats.csv $OUT/dense_kernel_stats.csv 2>/dev/null
cp $OUT/dense_pmc/summary.txt $OUT/dense_pmc_summary.txt 2>/dev/null
cp $OUT/nlm_pmc/summary.txt $OUT/nlm_pmc_summary.txt 2>/dev/null
# the raw traces are large (the merge back is limited to 64 MiB): keep the summaries only
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/dstats $OUT/dfetch $OUT/dwrite $OUT/dense_pmc $OUT/nlm_pmc
ls $OUT
cat $OUT/omnibus_rocprof.txt $OUT/omnibus_dense_rocprof.txt
