#!/bin/bash
# kernel-time breakdown of one model-zoo architecture: bash scripts/gpu_zoo_prof.sh <arch>
set -o pipefail
arch=${1:-mobilenet_v2}
mkdir -p gpurun_out/zoo_prof_$arch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/zoo_prof_$arch -o z -- python3 $R/benchmarks/zoo_bench.py --archs $arch --steps 5 --warmup 5 > $R/gpurun_out/zoo_prof_$arch/run.log 2>&1 || { tail -n 20 $R/gpurun_out/zoo_prof_$arch/run.log; exit 1; }
grep "img/s" $R/gpurun_out/zoo_prof_$arch/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/zoo_prof_$arch/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:22]:
    print(f'{r["Name"][:120]:120s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
PY
