#!/bin/bash
# hvd + amp ordering fix: new tests, diagnostic record of one fixed run, repeated repro, bench sanity, torch-comm fallback
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -u -m pytest tests/test_p2p_gpu.py -x -v --timeout 200 --timeout-method thread \
  -k "hvd or overflow_check" > gpurun_out/r3_hvd_tests.log 2>&1 || { tail -n 40 gpurun_out/r3_hvd_tests.log; exit 1; }
tail -n 12 gpurun_out/r3_hvd_tests.log
rm -rf gpurun_out/hvd_diag && mkdir -p gpurun_out/hvd_diag
PDA_HVD_DIAG=gpurun_out/hvd_diag timeout -k 10 150 python -u scripts/repro_hvd_p2p_two_ranks.py 1 2 29590 > gpurun_out/diag_r1.log 2>&1 &
pid=$!
PDA_HVD_DIAG=gpurun_out/hvd_diag timeout -k 10 150 python -u scripts/repro_hvd_p2p_two_ranks.py 0 2 29590 > gpurun_out/diag_r0.log 2>&1
rc0=$?; wait $pid; rc1=$?
echo "diag run rc $rc0 $rc1"; tail -n 2 gpurun_out/diag_r0.log
[ $rc0 -ge 124 ] && exit 1
python scripts/hvd_diag_diff.py gpurun_out/hvd_diag > gpurun_out/hvd_diag_diff.txt 2>&1; tail -n 5 gpurun_out/hvd_diag_diff.txt
bash scripts/gpu_hvd_flaky.sh > gpurun_out/r3_hvd_flaky.log 2>&1 || { tail -n 20 gpurun_out/r3_hvd_flaky.log; exit 1; }
bash scripts/gpu_hvd_flaky.sh >> gpurun_out/r3_hvd_flaky.log 2>&1 || { tail -n 20 gpurun_out/r3_hvd_flaky.log; exit 1; }
cat gpurun_out/r3_hvd_flaky.log | cut -c1-200
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 > gpurun_out/r3_bench0.log 2>&1 || { tail -n 20 gpurun_out/r3_bench0.log; exit 1; }
tail -n 1 gpurun_out/r3_bench0.log | cut -c1-600
# last: the torch.distributed fallback of the engine's device path (used to hang in its first step)
PDA_REPRO_DEVICE_COMM=torch timeout -k 10 120 python -u scripts/repro_hvd_p2p_two_ranks.py 1 2 29591 > gpurun_out/torchcomm_r1.log 2>&1 &
pid=$!
PDA_REPRO_DEVICE_COMM=torch timeout -k 10 120 python -u scripts/repro_hvd_p2p_two_ranks.py 0 2 29591 > gpurun_out/torchcomm_r0.log 2>&1
rc0=$?; wait $pid; rc1=$?
echo "torch-comm fallback rc $rc0 $rc1"; tail -n 3 gpurun_out/torchcomm_r0.log
