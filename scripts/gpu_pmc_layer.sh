#!/bin/bash
# PMC pass over one compute-bound layer (fprop / dgrad / wgrad of the 14x14 256->256 3x3 conv, batch 256)
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for op in fprop dgrad wgrad; do
timeout -s KILL 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$op -o r -- python3 $R/benchmarks/one_layer.py --op $op --h 14 --cin 256 --cout 256 --k 3 --batch 256 > $R/gpurun_out/pmc_$op.log 2>&1 || { tail -n 5 $R/gpurun_out/pmc_$op.log; exit 1; }
done
ls $R/gpurun_out/pmc_fprop
