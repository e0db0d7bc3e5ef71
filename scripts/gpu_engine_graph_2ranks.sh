#!/bin/bash
# two ranks of tests/workers/engine_graph_worker.py on GPU 0 with both outputs kept (a hang in one rank shows in the other's log)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
PORT=$((20000 + RANDOM % 20000))
timeout -k 5 90 python3 -u tests/workers/engine_graph_worker.py 0 2 $PORT > gpurun_out/eg_rank0.log 2>&1 &
P0=$!
timeout -k 5 90 python3 -u tests/workers/engine_graph_worker.py 1 2 $PORT > gpurun_out/eg_rank1.log 2>&1 &
P1=$!
wait $P0; R0=$?
wait $P1; R1=$?
echo "rc $R0 $R1"
for r in 0 1; do echo "---- rank $r"; grep -v "amdgpu.ids\|Gloo" gpurun_out/eg_rank$r.log | tail -n 25 | cut -c1-400; done
[ $R0 -eq 0 ] && [ $R1 -eq 0 ]
