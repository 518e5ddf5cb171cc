#!/bin/bash
# PMC counters of the omnibus kernels in the dense regime (two passes: the SQ block has 8 slots).
# usage (on the GPU box): bash tools/pmc_dense.sh <outdir> [alpha]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; A=${2:-0.01}; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $OUT/pmc1 -o p --output-format csv -- python3 $R/tools/bench_dense.py --alphas $A --steps 2 --cpu-rows 0 > $OUT/run1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_IFETCH -d $OUT/pmc2 -o p --output-format csv -- python3 $R/tools/bench_dense.py --alphas $A --steps 2 --cpu-rows 0 > $OUT/run2.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc1/p_counter_collection.csv $OUT/pmc2/p_counter_collection.csv > $OUT/summary.txt
cat $OUT/summary.txt
