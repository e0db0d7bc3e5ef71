#!/bin/bash
# bench modes beyond the default: hvd engine (+SyncBN, +eval), selftest, bucket sweep, graphed DDP; shared-GPU rehearsals
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
S=gpurun_out/r3_modes_summary.txt; : > $S
run() { # name, command...
  local name=$1; shift
  timeout -k 10 400 "$@" > gpurun_out/r3_mode_$name.log 2>&1; local rc=$?
  echo "[$name] rc $rc: $(grep '^{' gpurun_out/r3_mode_$name.log | tail -n 1 | cut -c1-1800)" >> $S
  if [ $rc -ne 0 ]; then tail -n 25 gpurun_out/r3_mode_$name.log; echo "FAILED $name"; exit 1; fi
}
timeout -k 10 900 python -u -m pytest tests/test_p2p_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_p2p_tests.log 2>&1 || { tail -n 40 gpurun_out/r3_p2p_tests.log; exit 1; }
tail -n 2 gpurun_out/r3_p2p_tests.log
run native_1gpu python -u bench.py --steps 20 --warmup 10
run hvd_1gpu python -u bench.py --engine hvd --steps 20 --warmup 10 --eval-steps 10
run hvd_syncbn_1gpu python -u bench.py --engine hvd --sync-bn --steps 20 --warmup 10 --eval-steps 10
export PDA_BENCH_SHARE_GPU=1 PDA_BENCH_STAGE_TIMEOUT=120
SMALL="--arch resnet18 --batch-size 64 --image-size 64 --steps 6 --warmup 3"
run share2_selftest_sweep python -u bench.py --gpus 2 --comm p2p2 --selftest --bucket-mb 8,16,25 $SMALL
run share2_graph python -u bench.py --gpus 2 --comm p2p2 --graph $SMALL
run share2_hvd python -u bench.py --gpus 2 --comm p2p2 --engine hvd --sync-bn --eval-steps 3 --selftest $SMALL
unset PDA_BENCH_SHARE_GPU
timeout -k 10 300 python -u benchmarks/host_phases.py --precision fp16 > gpurun_out/r3_host_phases_fp16.txt 2>&1 || { tail -n 20 gpurun_out/r3_host_phases_fp16.txt; exit 1; }
tail -n 12 gpurun_out/r3_host_phases_fp16.txt
cat $S | cut -c1-700
