#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu_k.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_gpu_k.log; [ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --list-avail > $R/gpurun_out/rocprof_avail.txt 2>&1 || true
grep -c . $R/gpurun_out/rocprof_avail.txt
for cfg in 0 16; do
timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_dgrad1x1_cfg$cfg -o kt -- python3 $R/benchmarks/one_layer.py --op dgrad --h 56 --cin 256 --cout 512 --k 1 --stride 2 --cfg $cfg > $R/gpurun_out/pmc_dgrad1x1_cfg$cfg.log 2>&1 || { tail -n 5 $R/gpurun_out/pmc_dgrad1x1_cfg$cfg.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_dgrad1x1_cfg$cfg -o sq -- python3 $R/benchmarks/one_layer.py --op dgrad --h 56 --cin 256 --cout 512 --k 1 --stride 2 --cfg $cfg >> $R/gpurun_out/pmc_dgrad1x1_cfg$cfg.log 2>&1 || { tail -n 5 $R/gpurun_out/pmc_dgrad1x1_cfg$cfg.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_dgrad1x1_cfg$cfg -o tcc -- python3 $R/benchmarks/one_layer.py --op dgrad --h 56 --cin 256 --cout 512 --k 1 --stride 2 --cfg $cfg >> $R/gpurun_out/pmc_dgrad1x1_cfg$cfg.log 2>&1 || { tail -n 5 $R/gpurun_out/pmc_dgrad1x1_cfg$cfg.log; exit 1; }
done
cd $R
timeout -k 10 300 python -u bench.py --batch-size 128 --steps 30 --warmup 8 > gpurun_out/bench_native_b128.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b128.log; exit 1; }
tail -n 1 gpurun_out/bench_native_b128.log | cut -c1-200
timeout -k 10 300 python -u bench.py --batch-size 256 --steps 30 --warmup 8 > gpurun_out/bench_native_b256.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b256.log; exit 1; }
tail -n 1 gpurun_out/bench_native_b256.log | cut -c1-200
