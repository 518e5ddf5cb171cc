#!/bin/bash
# PMC counters of the unit-weight window kernels on the tutorial filter (config 5's share,
# 24 x 2048 x 16384 x 4 variables): ring form, streaming form with 4 and with 2 rows per thread.
# usage (on the GPU box): bash tools/pmc_nlm_window.sh <outdir>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; RAW=/tmp/nlmwin_raw; mkdir -p $OUT $RAW   # raw traces stay on the box
for form in ring s3_oy4 s3_oy2; do
  unset ND_AMD_NLM_NOSTREAM3 ND_AMD_NLM_S3_OY4
  if [ $form = ring ]; then export ND_AMD_NLM_NOSTREAM3=1; fi
  if [ $form = s3_oy4 ]; then export ND_AMD_NLM_S3_OY4=1; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $RAW/${form}_a -o p --output-format csv -- python3 $R/tools/bench_pipeline.py --nx 16384 --steps 1 --alpha 1e-4 > $OUT/${form}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $RAW/${form}_b -o p --output-format csv -- python3 $R/tools/bench_pipeline.py --nx 16384 --steps 1 --alpha 1e-4 > $OUT/${form}_b.log 2>&1
  rocprofv3 --kernel-trace --stats -d $RAW/${form}_t -o p --output-format csv -- python3 $R/tools/bench_pipeline.py --nx 16384 --steps 3 --alpha 1e-4 > $OUT/${form}_t.log 2>&1
  echo "=== $form" >> $OUT/summary.txt
  python3 $R/tools/pmc_summary.py $RAW/${form}_a/p_counter_collection.csv $RAW/${form}_b/p_counter_collection.csv | grep -A9 "nlmeans_window" >> $OUT/summary.txt
  grep "nlmeans_window" $RAW/${form}_t/p_kernel_stats.csv | cut -c1-200 >> $OUT/summary.txt
done
cat $OUT/summary.txt
