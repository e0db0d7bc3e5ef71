#!/bin/bash
# kernel-time breakdown of one training step of selected model-zoo families: bash scripts/gpu_prof_zoo.sh "vgg16 mobilenet_v2"
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for arch in $1; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_zoo_$arch -o r -- python3 $R/benchmarks/zoo_bench.py --archs $arch --batch 128 --steps 5 --warmup 6 > $R/gpurun_out/prof_zoo_$arch.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_zoo_$arch.log; exit 1; }
  find $R/gpurun_out/prof_zoo_$arch -name "*kernel_trace.csv" -delete
  tail -n 1 $R/gpurun_out/prof_zoo_$arch.log | cut -c1-200
done
