#!/bin/bash
# HBM traffic of every bench workload's kernels: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
# passes (with --kernel-trace only, as MI355X_MICROARCH.md prescribes) over `bench.py --traffic-run KEY`.
#   bash tools/collect_traffic.sh gpurun_out/traffic [commit] [KEY ...]    (on the GPU box)
# writes <outdir>/traffic.json -- copy it to profiles/r05_traffic.json, which bench.py reads.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; COMMIT=${2:-unknown}; shift; shift
KEYS="$@"
[ -z "$KEYS" ] && KEYS="headline omnibus_a0.01 omnibus_a0.0001 omnibus_a0.2 ml3 ml5 pm_a0.99 pm_a0.01 c3_a0.99 c3_a0.01 c3_pm_a0.99 boxcar3 boxcar5 gauss1 nlm_pm0 nlm_pm1 pipeline"
mkdir -p $OUT
# (a heartbeat: a run that writes nothing for seven minutes is taken to be hung)
( while true; do sleep 60; date >> $OUT/heartbeat.log; done ) &
HB=$!
for K in $KEYS; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr_$C
    # (counters of this library's kernels only: the synthesis of a full-pol stack is 17 000 torch launches, at which a
    #  counter pass over every dispatch gives up; with the filter the REAL workload can be profiled -- ND_AMD_TRAFFIC_FULL_SYNTH)
    ND_AMD_TRAFFIC_FULL_SYNTH=1 timeout -k 5 1000 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex nd_amd -d /tmp/tr_$C -o p --output-format csv -- python3 $R/bench.py --traffic-run $K > $OUT/run_${K}_$C.log 2>&1
    F=$(ls /tmp/tr_$C/*/p_counter_collection.csv /tmp/tr_$C/p_counter_collection.csv 2>/dev/null | head -1)
    cp "$F" $OUT/${K}_$C.csv
  done
  echo "collected $K"
done
kill $HB 2>/dev/null
python3 $R/tools/summarize_traffic.py $OUT $COMMIT $KEYS > $OUT/traffic.json
rm -f $OUT/*_FETCH_SIZE.csv $OUT/*_WRITE_SIZE.csv
cat $OUT/traffic.json | head -c 3000
