#!/bin/bash
# SQ counters (two passes: the SQ block has 8 slots) of every nd_amd kernel a python command launches.
# usage (on the GPU box): bash tools/pmc_cmd.sh <outdir under the repo> <script.py> [args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; shift; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $OUT/pmc1 -o p --output-format csv -- python3 $R/"$@" > $OUT/run1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_IFETCH -d $OUT/pmc2 -o p --output-format csv -- python3 $R/"$@" > $OUT/run2.log 2>&1
python3 $R/tools/pmc_summary.py $(ls $OUT/pmc1/*/p_counter_collection.csv $OUT/pmc1/p_counter_collection.csv 2>/dev/null | head -1) $(ls $OUT/pmc2/*/p_counter_collection.csv $OUT/pmc2/p_counter_collection.csv 2>/dev/null | head -1) > $OUT/summary.txt
rm -rf $OUT/pmc1 $OUT/pmc2
cat $OUT/summary.txt
