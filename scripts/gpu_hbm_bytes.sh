#!/bin/bash
# HBM traffic of the benchmark step: FETCH_SIZE and WRITE_SIZE (KB moved between L2 and memory) summed over every
# dispatch of a short run with the tuner off (each counter set in a run of its own, kernel trace only).
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/hbm_$c -o r -- python3 $R/bench.py --steps 4 --warmup 2 --no-autotune --no-compare-stock > $R/gpurun_out/hbm_$c.log 2>&1 || { tail -n 15 $R/gpurun_out/hbm_$c.log; exit 1; }
  grep '^{' $R/gpurun_out/hbm_$c.log | tail -n 1 | cut -c1-200
done
python3 - <<PY
import csv, glob
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("$R/gpurun_out/hbm_%s/**/*counter_collection.csv" % c, recursive=True)
    s = 0.0; n = 0
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                s += float(row["Counter_Value"]); n += 1
    tot[c] = s
    print(c, "sum over", n, "dispatches:", round(s / 1e6, 2), "GB (counter unit KB)")
steps = 6
print("per step (6 steps in the run, start-up kernels included): read %.2f GB, written %.2f GB" % (tot["FETCH_SIZE"] / 1e6 / steps, tot["WRITE_SIZE"] / 1e6 / steps))
PY
find $R/gpurun_out/hbm_FETCH_SIZE $R/gpurun_out/hbm_WRITE_SIZE -name "*.csv" -size +2M -delete
