#!/bin/bash
# ResNet-18 on 32x32 input, batch 800 (the launch-bound workload): A/B of the shortcut / block-output fusions, eager and graph
set -o pipefail
mkdir -p gpurun_out
run() { # label, env..., -- args
  local label=$1; shift
  env "$@" timeout -k 10 300 python -u bench.py --arch resnet18 --image-size 32 --batch-size 800 --steps 100 --warmup 20 --no-compare-stock $EXTRA > gpurun_out/r18ab.log 2>&1 || { tail -n 20 gpurun_out/r18ab.log; exit 1; }
  echo "$label $EXTRA $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r18ab.log | tail -n 1)" | tee -a gpurun_out/r18_ab.txt
}
rm -f gpurun_out/r18_ab.txt
for EXTRA in "" "--graph"; do
  run default PDA_NOP=1 || exit 1
  run no_stem_fusion PDA_FUSE_STEM_BWD=0 || exit 1
  run no_shortcut_links PDA_DEFER_SHORTCUT=0 PDA_COMPACT_SHORTCUT_GRAD=0 || exit 1
  run no_tail PDA_FUSE_TAIL_BWD=0 || exit 1
  run default PDA_NOP=1 || exit 1
done
