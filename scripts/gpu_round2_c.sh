#!/bin/bash
# GPU session C: smoke, two-shot peer-memory all-reduce tests, comm microbench, DDP benches
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 200 python -u __graft_entry__.py smoke > gpurun_out/c_smoke.log 2>&1; rc=$?; tail -2 gpurun_out/c_smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -u -m pytest tests/test_p2p_gpu.py -x -q --timeout 150 --timeout-method thread > gpurun_out/c_pytest_p2p.log 2>&1; rc=$?
tail -12 gpurun_out/c_pytest_p2p.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u -m pytorch_distributed_amd.launch --nproc_per_node 4 --master_port 29641 benchmarks/comm_bw.py --comm p2p2 --same-gpu --out gpurun_out/c_comm_bw_4r_samegpu.jsonl > gpurun_out/c_comm_bw.log 2>&1; rc=$?; tail -14 gpurun_out/c_comm_bw.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --force-ddp > gpurun_out/c_bench_ddp.log 2>&1; rc=$?; tail -2 gpurun_out/c_bench_ddp.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --force-ddp --comm torch > gpurun_out/c_bench_ddp_torch.log 2>&1; rc=$?; tail -1 gpurun_out/c_bench_ddp_torch.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --force-ddp --comm p2p2 > gpurun_out/c_bench_ddp_p2p2.log 2>&1; rc=$?; tail -1 gpurun_out/c_bench_ddp_p2p2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 > gpurun_out/c_bench_plain.log 2>&1; rc=$?; tail -1 gpurun_out/c_bench_plain.log
exit $rc
