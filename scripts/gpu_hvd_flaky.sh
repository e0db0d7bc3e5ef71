#!/bin/bash
# the two-rank Horovod-wrapper worker a few times in a row: which parameters, if any, differ between the ranks
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3 4 5 6 7 8 9 10; do
  port=$((29600 + i))
  timeout -k 10 100 python -u scripts/repro_hvd_p2p_two_ranks.py 1 2 $port > gpurun_out/hvdf_r1.log 2>&1 &
  pid=$!
  timeout -k 10 100 python -u scripts/repro_hvd_p2p_two_ranks.py 0 2 $port > gpurun_out/hvdf_r0.log 2>&1
  rc0=$?
  wait $pid; rc1=$?
  echo "run $i rc $rc0 $rc1: $(grep '^rank 0:' gpurun_out/hvdf_r0.log | cut -c1-260)"
  if [ $rc0 -ge 124 ] || [ $rc1 -ge 124 ]; then echo "stopping: a rank timed out or crashed"; tail -n 5 gpurun_out/hvdf_r0.log; exit 1; fi
done
