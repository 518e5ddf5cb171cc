#!/bin/bash
# Rehearsal of the driver's N > 1 launch line of bench.py on a one-GPU box (ranks share the GPU over
# gloo).  Timings per variant go to gpurun_out/rehearse.log.
export ND_AMD_BENCH_REHEARSE=gloo
mkdir -p gpurun_out
for v in "omnibus" "omnibus --scaling strong" "pipeline" "c3"; do
  set -- $v
  wl=$1; shift
  if [ "$wl" = pipeline ]; then sz="--ny 64 --nx 512 --k 6"; else sz="--ny 96 --nx 512 --k 8"; fi
  t0=$(date +%s.%N)
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --workload $wl $sz "$@" \
      > gpurun_out/rehearse_$wl$#.out 2> gpurun_out/rehearse_$wl$#.err
  rc=$?
  t1=$(date +%s.%N)
  echo "$v rc=$rc $(echo "$t1 - $t0" | bc) s" >> gpurun_out/rehearse.log
done
cat gpurun_out/rehearse.log
