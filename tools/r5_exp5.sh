#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r5_exp5; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_correlate_gpu.py tests/test_gaussian_gpu.py -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python tools/fuzz_parity.py --seconds 60 --what correlate --seed 9 > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
timeout -k 10 300 python tools/bench_conv3d.py > $OUT/conv3d_tiled.txt 2>&1; cat $OUT/conv3d_tiled.txt | tail -4
ND_AMD_NO_TILED=1 timeout -k 10 300 python tools/bench_conv3d.py > $OUT/conv3d_generic.txt 2>&1; cat $OUT/conv3d_generic.txt | tail -4
timeout -k 5 300 python3 tools/exp_bench_extra.py boxcar3,boxcar5 > $OUT/boxcar.txt 2>&1; tail -2 $OUT/boxcar.txt
for i in 1 2; do
timeout -k 5 300 python3 tools/exp_bench_extra.py pipeline > $OUT/pipeline_new$i.txt 2>&1; tail -1 $OUT/pipeline_new$i.txt
ND_AMD_LIB=$R/_variants/lib_nlm_r04.so timeout -k 5 300 python3 tools/exp_bench_extra.py pipeline > $OUT/pipeline_r04_$i.txt 2>&1; tail -1 $OUT/pipeline_r04_$i.txt
done
