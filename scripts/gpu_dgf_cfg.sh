#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u benchmarks/dgrad_fusion_bench.py --batch 256 --cfgs 0,9,10,2 > gpurun_out/dgf_cfg_main.txt 2>&1 || { tail -n 20 gpurun_out/dgf_cfg_main.txt; exit 1; }
PDA_KERNELS_LIB=$GRAFT_REPO_ROOT/pytorch_distributed_amd/lib/libpda_kernels_w3.so timeout -k 10 300 python -u benchmarks/dgrad_fusion_bench.py --batch 256 --cfgs 0,9,10,2 > gpurun_out/dgf_cfg_w3.txt 2>&1 || { tail -n 20 gpurun_out/dgf_cfg_w3.txt; exit 1; }
echo ok
