#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ddp -o r -- python3 $R/bench.py --batch-size 400 --steps 5 --warmup 3 > $R/gpurun_out/prof_ddp.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_ddp.log; exit 1; }
true
echo done
