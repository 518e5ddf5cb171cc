#!/bin/bash
# Round-4 end-of-round evidence (run on the GPU box): bash tools/collect_r04.sh gpurun_out/r04prof <commit>
#   1. rocprofv3 --kernel-trace --stats of the headline bench command + separate FETCH_SIZE / WRITE_SIZE passes
#      -> omnibus_rocprof.txt   2. the default bench line -> bench_line.json   3. the fuzz campaign -> fuzz.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; COMMIT=${2:-unknown}; mkdir -p $OUT
CMD="python3 $R/bench.py --steps 20 --warmup 5 --no-extra --cpu-rows 0"
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- $CMD > $OUT/stats.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-extra --cpu-rows 0 > $OUT/fetch.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-extra --cpu-rows 0 > $OUT/write.log 2>&1
python3 $R/tools/summarize_prof.py $OUT/stats $OUT/fetch $OUT/write $OUT/omnibus_rocprof.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-extra --cpu-rows 0   (PMC passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE with --kernel-trace, --steps 5 --warmup 1); commit $COMMIT" > /dev/null
grep '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_line_under_rocprof.json
rm -rf $OUT/stats $OUT/fetch $OUT/write
echo "headline profiled"
(cd $R && timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err)
echo "bench line rc=$?"
cd $R
for spec in "300 94001 omnibus" "100 94002 c3" "100 94003 omnibus_ml" "60 94004 correlate,gaussian" "60 94005 nlmeans"; do
  set -- $spec
  echo "  --seconds $1 --seed $2 --what $3" >> $OUT/fuzz.txt
  timeout -k 10 $(( $1 + 120 )) python3 tools/fuzz_parity.py --seconds $1 --seed $2 --what $3 2>&1 | grep -v amdgpu.ids | tail -2 | sed 's/^/     /' >> $OUT/fuzz.txt
  echo "fuzz $3 done"
done
cat $OUT/omnibus_rocprof.txt | head -12; cat $OUT/fuzz.txt
