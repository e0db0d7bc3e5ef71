#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in 28,512,128,1,1 14,256,256,3,1 56,256,64,1,1; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dgf_$L -o r -- python3 $R/benchmarks/dgrad_fusion_bench.py --batch 256 --layer $L > $R/gpurun_out/prof_dgf_$L.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_dgf_$L.log; exit 1; }
done
find $R/gpurun_out -name "*kernel_trace.csv" -delete
echo done
