#!/bin/bash
# kernel-time breakdown of the default benchmark configuration (per-GPU batch 400): rocprofv3 statistics of the
# whole run plus the table of ONE steady-state step cut out of the kernel trace - (a) as benchmarked, (b) with the
# weight gradients kept on the main stream (per-kernel times without co-running kernels), (c) single-rank DDP
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, extra bench args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -o r -- python3 $R/bench.py --batch-size 400 --steps 5 --warmup 3 "$@" > $R/gpurun_out/prof_$name.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_$name.log; exit 1; }
  PDA_TRACE_LAUNCHES=$R/gpurun_out/prof_${name}_launches.txt python3 $R/scripts/trace_last_step.py $R/gpurun_out/prof_$name > $R/gpurun_out/prof_${name}_step.txt 2>&1
  find $R/gpurun_out/prof_$name -name "*kernel_trace.csv" -delete
  echo "$name: $(head -n 1 $R/gpurun_out/prof_${name}_step.txt)"
}
if [ "$1" = "one" ]; then PDA_WGRAD_STREAM=0 prof b400_one_stream --no-compare-stock; exit 0; fi
prof b400 --no-compare-stock
PDA_WGRAD_STREAM=0 prof b400_one_stream --no-compare-stock
prof b400_ddp --force-ddp
