#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r5_exp3; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_nlmeans_gpu.py tests/test_config_share_gpu.py tests/test_api_gpu.py -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -3 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python tools/fuzz_parity.py --seconds 90 --what nlmeans --seed 7 > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
timeout -k 5 300 python3 tools/exp_bench_extra.py pipeline > $OUT/pipeline.txt 2>&1; tail -1 $OUT/pipeline.txt
timeout -k 5 300 python3 tools/exp_bench_extra.py pipeline > $OUT/pipeline2.txt 2>&1; tail -1 $OUT/pipeline2.txt
