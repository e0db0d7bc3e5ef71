#!/bin/bash
# per-phase shader-clock stamps of the igemm workgroups (diagnostic library: the ablation flags are live there)
set -o pipefail
export PDA_KERNELS_LIB=$PWD/pytorch_distributed_amd/lib/libpda_kernels_diag.so
for cfg in ${ST_CFGS:-1 5 7}; do
  for ab in ${ST_AB:-0 1 2 3}; do
    echo "== ablate $ab"; PDA_IGEMM_ABLATE=$ab PDA_IGEMM_STAMPS=1 timeout -k 10 60 python -u benchmarks/igemm_stamps.py --cfg $cfg ${ST_ARGS} 2>&1 | grep "cfg\|main loop\|write-out\|lifetime" || exit 1
  done
done
