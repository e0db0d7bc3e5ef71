#!/bin/bash
# kernel trace of one Linear layer's forward + backward (ours and the library), direct fp32 gradient sink
set -o pipefail
mkdir -p gpurun_out/lin_trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lin_trace -o lin -- python3 $R/benchmarks/linear_bench.py --batch 128 --shapes 25088x4096 --direct-grad > $R/gpurun_out/lin_trace/run.log 2>&1 || { tail -n 20 $R/gpurun_out/lin_trace/run.log; exit 1; }
tail -n 2 $R/gpurun_out/lin_trace/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/lin_trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
