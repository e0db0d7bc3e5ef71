#!/bin/bash
# A/B of two kernel-library builds inside one session: $1 = variant name, rest = conv_bench args
set -o pipefail
mkdir -p gpurun_out
V=$1; shift
for rep in 1 2; do
for lib in default $V; do
if [ $lib = default ]; then unset PDA_KERNELS_LIB; else export PDA_KERNELS_LIB=$GRAFT_REPO_ROOT/pytorch_distributed_amd/lib/libpda_kernels_$lib.so; fi
timeout -k 10 300 python -u benchmarks/conv_bench.py --batch 128 "$@" > gpurun_out/ab_$lib.log 2>&1 || { tail -n 20 gpurun_out/ab_$lib.log; exit 1; }
echo "$lib: $(tail -n 1 gpurun_out/ab_$lib.log)"
done
done
