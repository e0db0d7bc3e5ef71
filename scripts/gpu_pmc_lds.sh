#!/bin/bash
# LDS behaviour of the fprop tiles on the 14x14 256->256 3x3 layer (one PMC pass per run)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in ${PMC_CFGS:-1 5}; do
  timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD --kernel-trace -d $R/gpurun_out/pmcl_$cfg -o r --output-format csv -- python3 $R/benchmarks/one_layer.py --op fprop --h 14 --cin 256 --cout 256 --k 3 --batch 400 --cfg $cfg > $R/gpurun_out/pmcl_$cfg.log 2>&1 || exit 1
done
cd $R && python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/pmcl_*/")):
    f = glob.glob(d + "**/r_counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"][:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    for k, dd in agg.items():
        if "igemm" in k:
            n = max(cnt[(k, c)] for c in dd)
            print(os.path.basename(d.rstrip('/')), {c: round(x / n) for c, x in dd.items()})
PY
