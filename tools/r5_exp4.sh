#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r5_exp4; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_omnibus_gpu.py -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python tools/bench_pm_long.py --k 48 > $OUT/pm_long_48.txt 2>&1; cat $OUT/pm_long_48.txt | tail -2
timeout -k 10 300 python tools/bench_pm_long.py --k 96 --ny 1024 > $OUT/pm_long_96.txt 2>&1; cat $OUT/pm_long_96.txt | tail -2
timeout -k 10 300 python tools/bench_pm_long.py --k 32 > $OUT/pm_long_32.txt 2>&1; cat $OUT/pm_long_32.txt | tail -2
