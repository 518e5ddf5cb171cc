#!/bin/bash
# round 5: the wave form of the fused multilooking kernel -- tests first, then timings of both forms
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r5_ml; mkdir -p $OUT; cd $R
timeout -k 10 600 python -m pytest tests/test_omnibus_ml_gpu.py -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -15 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python tools/fuzz_parity.py --seconds 120 --what omnibus_ml --seed 5 > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
ND_AMD_ML_FORM=0 timeout -k 10 200 python tools/bench_ml.py --alphas 0.99 > $OUT/bench_block.txt 2>&1; cat $OUT/bench_block.txt | tail -2
ND_AMD_ML_FORM=1 timeout -k 10 200 python tools/bench_ml.py --alphas 0.99,0.01 > $OUT/bench_wave.txt 2>&1; cat $OUT/bench_wave.txt | tail -4
for V in "$@"; do
  ND_AMD_LIB=$R/_variants/lib_$V.so timeout -k 10 200 python tools/bench_ml.py --alphas 0.99 --no-two-step > $OUT/bench_$V.txt 2>&1; echo $V; tail -2 $OUT/bench_$V.txt
done
