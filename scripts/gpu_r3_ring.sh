#!/bin/bash
# ring-loop schedule (builtin waits, peeled steady state) against the previous loop; device-sequence peer kernels;
# graphed DDP step; bench sanity
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 300 --timeout-method thread \
  -k "ring or conv_igemm or dgrad or resnet50_bench or ddp_single" > gpurun_out/r3_ring_tests.log 2>&1 || { tail -n 30 gpurun_out/r3_ring_tests.log; exit 1; }
tail -n 3 gpurun_out/r3_ring_tests.log
for lib in new old; do
  if [ $lib = old ]; then export PDA_KERNELS_LIB=$GRAFT_REPO_ROOT/pytorch_distributed_amd/lib/libpda_kernels_ringold.so; else unset PDA_KERNELS_LIB; fi
  timeout -k 10 600 python -u benchmarks/conv_bench.py --batch 400 --cfgs 5,6,7 --only fd --iters 10 --out gpurun_out/ring_$lib.csv > gpurun_out/ring_$lib.txt 2>&1 || { tail -n 20 gpurun_out/ring_$lib.txt; exit 1; }
  grep "per-step" gpurun_out/ring_$lib.txt
done
unset PDA_KERNELS_LIB
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --no-compare-stock > gpurun_out/r3_bench1.log 2>&1 || { tail -n 20 gpurun_out/r3_bench1.log; exit 1; }
tail -n 1 gpurun_out/r3_bench1.log | cut -c1-300
PDA_KERNELS_LIB=$GRAFT_REPO_ROOT/pytorch_distributed_amd/lib/libpda_kernels_ringold.so timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --no-compare-stock > gpurun_out/r3_bench1_old.log 2>&1 || { tail -n 20 gpurun_out/r3_bench1_old.log; exit 1; }
tail -n 1 gpurun_out/r3_bench1_old.log | cut -c1-300
for extra in "" "--force-ddp" "--force-ddp --graph" "--graph"; do
  timeout -k 10 300 python -u bench.py --arch resnet18 --image-size 32 --batch-size 800 --steps 50 --warmup 15 --no-compare-stock $extra > gpurun_out/r3_r18.log 2>&1 || { tail -n 20 gpurun_out/r3_r18.log; exit 1; }
  echo "resnet18 32x32 b800 fp16 O2 [$extra]: $(tail -n 1 gpurun_out/r3_r18.log | cut -c1-220)" | tee -a gpurun_out/r3_r18_summary.txt
done
timeout -k 10 900 python -u -m pytest tests/test_p2p_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_p2p_tests.log 2>&1 || { tail -n 40 gpurun_out/r3_p2p_tests.log; exit 1; }
tail -n 3 gpurun_out/r3_p2p_tests.log
