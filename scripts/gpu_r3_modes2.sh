#!/bin/bash
# shared-GPU rehearsals of the bench modes (two ranks on GPU 0, peer-memory collectives)
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
S=gpurun_out/r3_modes2_summary.txt; : > $S
run() { local name=$1; shift
  timeout -k 10 170 "$@" > gpurun_out/r3_mode_$name.log 2>&1; local rc=$?
  echo "[$name] rc $rc: $(grep '^{' gpurun_out/r3_mode_$name.log | tail -n 1 | cut -c1-2500)" >> $S
  if [ $rc -ne 0 ]; then tail -n 25 gpurun_out/r3_mode_$name.log | cut -c1-400; echo "FAILED $name"; exit 1; fi
  echo "ok $name"
}
export PDA_BENCH_SHARE_GPU=1 PDA_BENCH_STAGE_TIMEOUT=50
SMALL="--arch resnet18 --batch-size 64 --image-size 64 --steps 6 --warmup 3"
run share2_hvd python -u bench.py --gpus 2 --comm p2p2 --engine hvd --sync-bn --eval-steps 3 --selftest $SMALL
unset PDA_BENCH_SHARE_GPU
timeout -k 10 170 python -u benchmarks/host_phases.py --precision fp16 > gpurun_out/r3_host_phases_fp16.txt 2>&1 || { tail -n 20 gpurun_out/r3_host_phases_fp16.txt; exit 1; }
tail -n 12 gpurun_out/r3_host_phases_fp16.txt
cut -c1-900 $S
bash scripts/gpu_prof_b400.sh one && head -n 40 gpurun_out/prof_b400_one_stream_step.txt | cut -c1-160
