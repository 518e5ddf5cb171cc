#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05b
bash tools/collect_traffic.sh gpurun_out/r05b/traffic $1 pipeline > gpurun_out/r05b/traffic.log 2>&1
echo "pipeline traffic done"; tail -c 900 gpurun_out/r05b/traffic/traffic.json
timeout -k 10 500 python tools/fuzz_parity.py --seconds 400 --seed 50 > gpurun_out/r05b/fuzz.log 2>&1; tail -3 gpurun_out/r05b/fuzz.log
timeout -k 10 300 python tools/exp_cliffs.py filters > gpurun_out/r05b/cliffs_filters.txt 2>&1; tail -5 gpurun_out/r05b/cliffs_filters.txt
