#!/bin/bash
# kernel-time breakdown of the default benchmark configuration (per-GPU batch 400): rocprofv3 statistics of the
# whole run plus the table of ONE steady-state step cut out of the kernel trace
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b400 -o r -- python3 $R/bench.py --batch-size 400 --steps 5 --warmup 3 > $R/gpurun_out/prof_b400.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_b400.log; exit 1; }
python3 $R/scripts/trace_last_step.py $R/gpurun_out/prof_b400 > $R/gpurun_out/prof_b400_step.txt 2>&1
find $R/gpurun_out -name "*kernel_trace.csv" -delete
tail -n 1 $R/gpurun_out/prof_b400.log | cut -c1-200
