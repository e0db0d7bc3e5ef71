mkdir -p gpurun_out
set -x
timeout -k 10 600 python -u -m pytest tests -m gpu -q --tb=line --timeout 120 --timeout-method thread -p no:cacheprovider -k "stem or resnet" > gpurun_out/pytest2.log 2>&1; tail -8 gpurun_out/pytest2.log | cut -c1-400
for b in 128 256; do
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --batch-size $b > gpurun_out/bench_native_b$b.log 2>&1 && tail -1 gpurun_out/bench_native_b$b.log | cut -c1-200 || exit 1
done
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --batch-size 128 --engine stock --channels-last > gpurun_out/bench_stock_cl_b128.log 2>&1 && tail -1 gpurun_out/bench_stock_cl_b128.log | cut -c1-200 || exit 1
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --batch-size 128 --engine stock > gpurun_out/bench_stock_b128.log 2>&1 && tail -1 gpurun_out/bench_stock_b128.log | cut -c1-200 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_native -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --batch-size 128 > $GRAFT_REPO_ROOT/gpurun_out/prof_native.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && find gpurun_out/prof_native -name "*kernel_stats.csv" | head -2
