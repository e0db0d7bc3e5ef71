#!/bin/bash
# rehearsal of the multi-rank benchmark path on the one-GPU box: the supervisor starts 2 (then 4) ranks that share GPU 0
# and reduce their gradient buckets through the peer-memory two-shot kernel (NOT a result: the ranks split one device)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 PDA_BENCH_SHARE_GPU=1 PDA_BENCH_STAGE_TIMEOUT=120
for n in 2 4; do
  timeout -k 10 400 python -u bench.py --gpus $n --comm p2p2 --batch-size 64 --steps 6 --warmup 4 > gpurun_out/rehearsal_$n.log 2> gpurun_out/rehearsal_$n.err; rc=$?
  tail -n 1 gpurun_out/rehearsal_$n.log | cut -c1-1200
  [ $rc -eq 0 ] || { tail -n 30 gpurun_out/rehearsal_$n.err; exit $rc; }
done
