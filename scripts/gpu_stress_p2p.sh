#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29811
timeout -k 10 200 python -u scripts/stress_p2p_collectives.py 1 2 $port > gpurun_out/stress_r1.log 2>&1 &
pid=$!
timeout -k 10 200 python -u scripts/stress_p2p_collectives.py 0 2 $port > gpurun_out/stress_r0.log 2>&1
rc0=$?; wait $pid; rc1=$?
echo "rc $rc0 $rc1"; tail -n 6 gpurun_out/stress_r0.log | cut -c1-220; tail -n 2 gpurun_out/stress_r1.log | cut -c1-220
