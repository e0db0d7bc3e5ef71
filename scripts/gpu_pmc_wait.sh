#!/bin/bash
# where do the waves of the conv kernels spend their cycles? (one PMC pass per op, SQ block only)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for op in fprop dgrad wgrad; do
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d $R/gpurun_out/pmcw_$op -o r --output-format csv -- python3 $R/benchmarks/one_layer.py --op $op --h 14 --cin 256 --cout 256 --k 3 --batch 256 > $R/gpurun_out/pmcw_$op.log 2>&1 || exit 1
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for op in ("fprop", "dgrad", "wgrad"):
    f = glob.glob(f"gpurun_out/pmcw_{op}/**/r_counter_collection.csv", recursive=True)
    if not f:
        print(op, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    for k, d in agg.items():
        if "igemm" in k or "wgrad" in k:
            n = max(cnt[(k, c)] for c in d)
            print(op, k, {c: round(v / n) for c, v in d.items()})
PY
