#!/bin/bash
# conv kernel iteration: numerics of the LDS-DMA ring tiles, then the per-layer table with every tile A/B in one process
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 200 --timeout-method thread -k "lds_dma_ring or tile_configs" > gpurun_out/ci_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/ci_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -u benchmarks/conv_bench.py --batch ${CONV_BATCH:-128} --cfgs ${CONV_CFGS:-0,1,5,6,7} --only fd > gpurun_out/ci_conv_ab.txt 2>&1; rc=$?
python - <<'PY'
import ast
for l in open("gpurun_out/ci_conv_ab.txt"):
    if l.startswith("{"):
        d = ast.literal_eval(l)
        print(d["H"], d["Cin"], d["Cout"], d["k"], d["s"], "| fprop", d.get("fprop_all"), "| dgrad", d.get("dgrad_all"))
    elif l.startswith("per-step"):
        print(l.strip())
PY
exit $rc
