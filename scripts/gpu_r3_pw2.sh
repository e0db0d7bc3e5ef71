#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for n in 0 64 128 256; do
  echo "== PDA_PW_NCH=$n" | tee -a gpurun_out/pw_nch.txt
  PDA_PW_NCH=$n timeout -k 10 200 python -u benchmarks/pw_bench.py --cfgs 33 2>&1 | tee -a gpurun_out/pw_nch.txt || exit 1
done
echo "== tiled" | tee -a gpurun_out/pw_nch.txt
timeout -k 10 300 python -u benchmarks/pw_bench.py --cfgs 0,1,2,9,10,12 2>&1 | tee -a gpurun_out/pw_nch.txt
