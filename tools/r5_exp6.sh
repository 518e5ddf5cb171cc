#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r5_exp6; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_correlate_gpu.py -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 400 python tools/fuzz_parity.py --seconds 150 --what correlate --seed 9 > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
timeout -k 10 400 python tools/fuzz_parity.py --seconds 60 --what gaussian,correlate --seed 12 > $OUT/fuzz2.log 2>&1; tail -2 $OUT/fuzz2.log
