#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
L=28,512,128,1,1
for V in nobv nogred; do
export PDA_KERNELS_LIB=$R/pytorch_distributed_amd/lib/libpda_kernels_$V.so
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dgfv_$V -o r -- python3 $R/benchmarks/dgrad_fusion_bench.py --batch 256 --layer $L > $R/gpurun_out/prof_dgfv_$V.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_dgfv_$V.log; exit 1; }
done
find $R/gpurun_out -name "*kernel_trace.csv" -delete
echo done
