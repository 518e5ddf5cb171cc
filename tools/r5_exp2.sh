#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r5_exp2; mkdir -p $OUT; cd $R
hipcc --offload-arch=gfx950 -O3 -o $OUT/probe_tiles tools/probe_tiles.hip || exit 1
timeout -k 5 200 $OUT/probe_tiles > $OUT/probe_tiles.txt 2>&1; cat $OUT/probe_tiles.txt; rm -f $OUT/probe_tiles
timeout -k 5 200 python3 tools/exp_bench_extra.py pm_a0.01 > $OUT/pm_base.txt 2>&1; tail -1 $OUT/pm_base.txt
ND_AMD_LIB=$R/_variants/lib_pm_direct_c12.so timeout -k 5 200 python3 tools/exp_bench_extra.py pm_a0.01 > $OUT/pm_direct_c12.txt 2>&1; tail -1 $OUT/pm_direct_c12.txt
