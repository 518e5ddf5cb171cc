#!/bin/bash
# PMC counters of the true-patch non-local means kernels (C-C: 7x7 patch / 21x21 search, 12 x 4096^2):
# the cross-lane form (default) and the one-column-per-lane form (ND_AMD_NLM_PATCH1=1).
# usage (on the GPU box): bash tools/pmc_nlm.sh <outdir> ["patch2 patch1"]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; mkdir -p $OUT
for form in ${2:-patch2 patch1}; do
  if [ $form = patch1 ]; then export ND_AMD_NLM_PATCH1=1; else unset ND_AMD_NLM_PATCH1; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d /tmp/nlmpmc_${form}_a -o p --output-format csv -- python3 $R/tools/bench_filters.py --what nlmeans --k 12 --steps 1 --warmup 1 > $OUT/${form}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d /tmp/nlmpmc_${form}_b -o p --output-format csv -- python3 $R/tools/bench_filters.py --what nlmeans --k 12 --steps 1 --warmup 1 > $OUT/${form}_b.log 2>&1
  echo "=== $form" >> $OUT/summary.txt
  python3 $R/tools/pmc_summary.py /tmp/nlmpmc_${form}_a/p_counter_collection.csv /tmp/nlmpmc_${form}_b/p_counter_collection.csv | grep -A9 "patch" >> $OUT/summary.txt
done
cat $OUT/summary.txt
