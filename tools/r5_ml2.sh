#!/bin/bash
# round 5: block form of the fused multilooking kernel with the L2 prefetch of the idle waves
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r5_ml2; mkdir -p $OUT; cd $R
timeout -k 10 600 python -m pytest tests/test_omnibus_ml_gpu.py -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -3 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 200 python tools/bench_ml.py --alphas 0.99 > $OUT/bench_base.txt 2>&1; tail -2 $OUT/bench_base.txt
for V in "$@"; do
  ND_AMD_LIB=$R/_variants/lib_$V.so timeout -k 10 200 python tools/bench_ml.py --alphas 0.99 --no-two-step > $OUT/bench_$V.txt 2>&1; echo $V; tail -2 $OUT/bench_$V.txt
done
timeout -k 10 200 python tools/fuzz_parity.py --seconds 60 --what omnibus_ml --seed 6 > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
