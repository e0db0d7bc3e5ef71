#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for op in wgrad fprop dgrad; do
D=$R/gpurun_out/pmc_${op}_14x256k3
ARGS="--op $op --h 14 --cin 256 --cout 256 --k 3 --stride 1 --batch 128 --iters 4"
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $D -o a -- python3 $R/benchmarks/one_layer.py $ARGS > $D.log 2>&1 || { tail -n 5 $D.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $D -o b -- python3 $R/benchmarks/one_layer.py $ARGS >> $D.log 2>&1 || { tail -n 5 $D.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -o c -- python3 $R/benchmarks/one_layer.py $ARGS >> $D.log 2>&1 || { tail -n 5 $D.log; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_native6 -o r -- python3 $R/bench.py --batch-size 256 --steps 5 --warmup 3 > $R/gpurun_out/prof_native6.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_native6.log; exit 1; }
echo done
