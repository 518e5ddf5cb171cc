#!/bin/bash
# PMC counters of the pixel-major search kernels (reference layout, alpha = 0.01 by default).
# usage (on the GPU box): bash tools/pmc_pm_dense.sh <outdir> [alpha]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; RAW=/tmp/pmdense_raw; mkdir -p $OUT $RAW
A=${2:-0.01}
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TA_BUSY" \
           "TA_ADDR_STALLED_BY_TC_CYCLES TA_BUFFER_READ_LDS_WAVEFRONTS TA_TOTAL_WAVEFRONTS TCP_READ_TAGCONFLICT_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_BUSY"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set -d $RAW/p$i -o p --output-format csv -- python3 $R/tools/bench_pixel_major.py --alpha $A > $OUT/run$i.log 2>&1
  echo "pass $i rc=$?"
done
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $RAW/t -o p --output-format csv -- python3 $R/tools/bench_pixel_major.py --alpha $A > $OUT/t.log 2>&1
python3 $R/tools/pmc_summary.py $(find $RAW -name p_counter_collection.csv | sort) > $OUT/summary.txt
grep -h "nd_amd" $(find $RAW/t -name p_kernel_stats.csv) 2>/dev/null | cut -c1-200 >> $OUT/summary.txt
cat $OUT/summary.txt
