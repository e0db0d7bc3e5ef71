#!/bin/bash
# round-3 evidence: step profile (two-stream as benchmarked + one-stream), per-layer conv table with the tuner's
# candidates, and the bench in its modes
set -o pipefail
mkdir -p gpurun_out
bash scripts/gpu_prof_b400.sh || exit 1
timeout -k 10 600 python -u benchmarks/conv_bench.py --batch 400 --iters 10 --cfgs 0,1,2,9,12,5,6,7,13,33,34 --wcfgs 0,1,12,65,66,68,69,70 --out gpurun_out/conv_layers_b400_r3.csv > gpurun_out/conv_layers_b400_r3.txt 2>&1 || { tail -n 20 gpurun_out/conv_layers_b400_r3.txt; exit 1; }
tail -n 1 gpurun_out/conv_layers_b400_r3.txt
run() { # name, args
  local name=$1; shift
  timeout -k 10 400 python -u bench.py "$@" > gpurun_out/r3f_$name.log 2> gpurun_out/r3f_$name.err || { tail -n 20 gpurun_out/r3f_$name.err; exit 1; }
  echo "[$name] $(tail -n 1 gpurun_out/r3f_$name.log)" >> gpurun_out/r3_final_modes.txt
  echo "[$name] $(grep -o '"value": [0-9.]*' gpurun_out/r3f_$name.log | tail -n 1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r3f_$name.log | tail -n 1) $(grep -o '"vs_stock": [0-9.a-z]*' gpurun_out/r3f_$name.log | tail -n 1)"
}
rm -f gpurun_out/r3_final_modes.txt
run native --steps 30 --warmup 10 || exit 1
run hvd --steps 30 --warmup 10 --engine hvd --eval-steps 10 || exit 1
run hvd_syncbn --steps 30 --warmup 10 --engine hvd --sync-bn || exit 1
run force_ddp --steps 30 --warmup 10 --force-ddp || exit 1
run bf16 --steps 30 --warmup 10 --precision bf16 --no-compare-stock || exit 1
run r18_eager --arch resnet18 --image-size 32 --batch-size 800 --steps 100 --warmup 20 --no-compare-stock || exit 1
run r18_graph --arch resnet18 --image-size 32 --batch-size 800 --steps 100 --warmup 20 --graph --no-compare-stock || exit 1
run r18_ddp --arch resnet18 --image-size 32 --batch-size 800 --steps 100 --warmup 20 --force-ddp --no-compare-stock || exit 1
run r18_ddp_graph --arch resnet18 --image-size 32 --batch-size 800 --steps 100 --warmup 20 --force-ddp --graph --no-compare-stock || exit 1
