#!/bin/bash
# per-layer table of the conv kernels at the benchmark's batch (400): best of the tuner's candidates per layer
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u benchmarks/conv_bench.py --batch 400 --cfgs 0,1,2,3,9,10,11,12,5,6,7 --wcfgs 0,1,2,3,9,10,11,12,13 --only fdw --iters 10 --out gpurun_out/conv_b400.csv > gpurun_out/conv_b400.txt 2>&1; rc=$?
python - <<'PY'
import ast
for l in open("gpurun_out/conv_b400.txt"):
    if l.startswith("{"):
        d = ast.literal_eval(l)
        print(f'{d["H"]:3d} {d["Cin"]:5d} {d["Cout"]:5d} k{d["k"]} s{d["s"]} x{d["count"]} io_floor {d["io_floor_us"]:6.0f} GF {d["GFLOP"]:6.1f} | f {d.get("fprop_us")} ({d.get("fprop_TF")}TF c{d.get("fprop_cfg")}) d {d.get("dgrad_us")} ({d.get("dgrad_TF")}TF c{d.get("dgrad_cfg")}) w {d.get("wgrad_us")} ({d.get("wgrad_TF")}TF c{d.get("wgrad_cfg")})')
    elif l.startswith("per-step"):
        print(l.strip())
PY
exit $rc
