#!/bin/bash
# memory-side and SQ counters of the fused multilooking kernel next to the plain pass A.
# usage (on the GPU box): bash tools/pmc_ml.sh <outdir under the repo> [ml] [alpha] [reps]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/$1; shift; mkdir -p $OUT
i=0
for set in "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_BUSY TCC_CYCLE" \
           "TCC_HIT TCC_MISS TCC_TAG_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL" \
           "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TA_BUSY" \
           "TA_ADDR_STALLED_BY_TC_CYCLES TA_BUFFER_READ_LDS_WAVEFRONTS TA_TOTAL_WAVEFRONTS TCP_READ_TAGCONFLICT_STALL_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p --output-format csv -- python3 $R/tools/ml_run.py "$@" > $OUT/run$i.log 2>&1
  echo "pass $i rc=$?"
done
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/ps -o p --output-format csv -- python3 $R/tools/ml_run.py "$@" > $OUT/runs.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob('$OUT/p[0-9]*')):
    f = glob.glob(p + '/**/p_counter_collection.csv', recursive=True)
    if not f: print(p, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        kn = r['Kernel_Name'][:48]
        if 'omnibus' not in kn: continue
        acc[kn][r['Counter_Name']] += float(r['Counter_Value'])
        key = (kn, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); n[kn] += 1
    for kn, d in acc.items():
        print(kn, 'dispatches', n[kn])
        for c, v in sorted(d.items()): print('    %-40s %.5g per dispatch' % (c, v / n[kn]))
f = glob.glob('$OUT/ps/**/p_kernel_stats.csv', recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if 'omnibus' in r['Name']: print('stats', r['Name'][:60], r['Calls'], r['AverageNs'])
PY
rm -rf $OUT/p[0-9]* $OUT/ps
