#!/bin/bash
# PMC counters of the pixel-major streaming search (reference layout, alpha = 0.01) next to the planar one.
# usage (on the GPU box): bash tools/pmc_pm_dense.sh <outdir>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; RAW=/tmp/pmdense_raw; mkdir -p $OUT $RAW
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $RAW/a -o p --output-format csv -- python3 $R/tools/bench_pixel_major.py --alpha 0.01 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM -d $RAW/b -o p --output-format csv -- python3 $R/tools/bench_pixel_major.py --alpha 0.01 > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --stats -d $RAW/t -o p --output-format csv -- python3 $R/tools/bench_pixel_major.py --alpha 0.01 > $OUT/t.log 2>&1
python3 $R/tools/pmc_summary.py $RAW/a/p_counter_collection.csv $RAW/b/p_counter_collection.csv > $OUT/summary.txt
grep "nd_amd" $RAW/t/*/p_kernel_stats.csv $RAW/t/p_kernel_stats.csv 2>/dev/null | cut -d: -f2- | cut -c1-200 >> $OUT/summary.txt
cat $OUT/summary.txt
