#!/bin/bash
# wave-cycle breakdown and L2 behaviour of the fprop tiles on the 14x14 256->256 3x3 layer (one PMC pass per run)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${PMC_BATCH:-256}
for cfg in ${PMC_CFGS:-1 5 7}; do
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmct_sq_$cfg -o r --output-format csv -- python3 $R/benchmarks/one_layer.py --op ${PMC_OP:-fprop} --h 14 --cin 256 --cout 256 --k 3 --batch $B --cfg $cfg > $R/gpurun_out/pmct_sq_$cfg.log 2>&1 || exit 1
  timeout -s KILL 90 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmct_tcc_$cfg -o r --output-format csv -- python3 $R/benchmarks/one_layer.py --op ${PMC_OP:-fprop} --h 14 --cin 256 --cout 256 --k 3 --batch $B --cfg $cfg > $R/gpurun_out/pmct_tcc_$cfg.log 2>&1 || exit 1
done
cd $R && python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/pmct_*_*/")):
    f = glob.glob(d + "**/r_counter_collection.csv", recursive=True)
    if not f:
        print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"][:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    for k, dd in agg.items():
        if "igemm" in k:
            n = max(cnt[(k, c)] for c in dd)
            v = {c: round(x / n) for c, x in dd.items()}
            extra = ""
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("SQ_BUSY_CYCLES"):
                # MFMA busy is summed over SIMDs in cycles; SQ_BUSY_CYCLES is per-SE busy time summed: report the raw ratio to wave cycles too
                extra = f" mfma/wave_quadcycles={v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * v['SQ_WAVE_CYCLES']):.3f}"
            print(os.path.basename(d.rstrip('/')), k[:48], v, extra)
PY
tail -2 gpurun_out/pmct_sq_*.log | grep "us/iter"
