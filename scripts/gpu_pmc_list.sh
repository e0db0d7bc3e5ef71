#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/pmc_avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*\|SQ_LDS[A-Z0-9_]*\|SQ_INSTS_LDS\|SQ_ACTIVE_INST_LDS\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INST_CYCLES_VMEM\|SQ_WAIT_INST_LDS\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES" $GRAFT_REPO_ROOT/gpurun_out/pmc_avail.txt | sort -u | head -40
