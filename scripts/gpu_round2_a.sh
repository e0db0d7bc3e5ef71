#!/bin/bash
# GPU session A: sanity bench, comm-stream hand-off probe, GPU tests, single-rank DDP bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u bench.py --steps 10 --warmup 5 > gpurun_out/a_bench_plain.log 2>&1 && tail -1 gpurun_out/a_bench_plain.log && \
timeout -k 10 500 python -u benchmarks/overlap_probe.py --out gpurun_out/overlap_probe.txt \
  --variants native_prio,native_normal,native_inorder,pystream_normal,torch_chain_normal,torch_chain_prio,pystream_prio \
  > gpurun_out/overlap_probe.log 2>&1 ; rc=$?; tail -30 gpurun_out/overlap_probe.txt; [ $rc -eq 0 ] && \
timeout -k 10 300 python -u bench.py --steps 10 --warmup 5 --force-ddp > gpurun_out/a_bench_ddp.log 2>&1 ; tail -2 gpurun_out/a_bench_ddp.log
