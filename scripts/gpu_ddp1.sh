#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 > gpurun_out/bench_plain.log 2>&1 || { tail -n 20 gpurun_out/bench_plain.log; exit 2; }
echo "plain $(tail -n 1 gpurun_out/bench_plain.log | grep -o '"value": [0-9.]*')"
timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 30 --warmup 8 --force-ddp > gpurun_out/bench_ddp1.log 2>&1 || { tail -n 20 gpurun_out/bench_ddp1.log; exit 2; }
echo "torchrun force-ddp $(tail -n 1 gpurun_out/bench_ddp1.log | grep -o '"value": [0-9.]*')"
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --force-ddp --native-comm > gpurun_out/bench_ddp1n.log 2>&1 || { tail -n 20 gpurun_out/bench_ddp1n.log; exit 2; }
echo "force-ddp native comm $(tail -n 1 gpurun_out/bench_ddp1n.log | grep -o '"value": [0-9.]*')"
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --force-ddp --sync-bn > gpurun_out/bench_ddp1s.log 2>&1 || { tail -n 20 gpurun_out/bench_ddp1s.log; exit 2; }
echo "force-ddp sync-bn $(tail -n 1 gpurun_out/bench_ddp1s.log | grep -o '"value": [0-9.]*')"
