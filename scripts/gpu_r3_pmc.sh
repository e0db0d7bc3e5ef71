#!/bin/bash
# PMC pass over the 14x14 256->256 3x3 layer (batch 256) with the ring kernels: fprop ring tiles after the
# builtin-wait loop, the ring wgrad, and the A-stationary pointwise kernel on the 14x14 256->1024 layer
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run() {  # name, args...
  local name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc3_$name -o r -- python3 $R/benchmarks/one_layer.py "$@" > $R/gpurun_out/pmc3_$name.log 2>&1 || { tail -n 5 $R/gpurun_out/pmc3_$name.log; exit 1; }
}
run f1 --op fprop --h 14 --cin 256 --cout 256 --k 3 --batch 256 --cfg 1 || exit 1
run f5 --op fprop --h 14 --cin 256 --cout 256 --k 3 --batch 256 --cfg 5 || exit 1
run f6 --op fprop --h 14 --cin 256 --cout 256 --k 3 --batch 256 --cfg 6 || exit 1
run f7 --op fprop --h 14 --cin 256 --cout 256 --k 3 --batch 256 --cfg 7 || exit 1
run w1 --op wgrad --h 14 --cin 256 --cout 256 --k 3 --batch 256 --cfg 1 || exit 1
run w68 --op wgrad --h 14 --cin 256 --cout 256 --k 3 --batch 256 --cfg 68 || exit 1
run w65 --op wgrad --h 14 --cin 256 --cout 256 --k 3 --batch 256 --cfg 65 || exit 1
run p33 --op fprop --h 14 --cin 256 --cout 1024 --k 1 --batch 256 --cfg 33 || exit 1
run p12 --op fprop --h 14 --cin 256 --cout 1024 --k 1 --batch 256 --cfg 12 || exit 1
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for d in sorted(glob.glob(R + "/gpurun_out/pmc3_*/")):
    f = glob.glob(d + "*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
    for k, c in agg.items():
        if "igemm" not in k and "wgrad" not in k and "pw_fprop" not in k:
            continue
        L = n[k]
        cyc = c["GRBM_GUI_ACTIVE"] / L / 8
        mf = c["SQ_VALU_MFMA_BUSY_CYCLES"] / L
        print(f"{os.path.basename(d[:-1]):10s} {k:60s} launches {L}: kernel cycles {cyc:9.0f}  MFMA busy {100 * mf / (cyc * 1024):5.1f} % of SIMD-cycles  "
              f"LDS active {100 * c['SQ_LDS_IDX_ACTIVE'] / L / (cyc * 256):5.1f} % of CU-cycles  bank conflicts {c['SQ_LDS_BANK_CONFLICT'] / L:9.0f}  wait-LDS {c['SQ_WAIT_INST_LDS'] / L:10.0f}  wave cycles {c['SQ_WAVE_CYCLES'] / L:11.0f}")
PY
