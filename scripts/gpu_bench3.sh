mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -q --tb=line --timeout 120 --timeout-method thread -p no:cacheprovider -x > gpurun_out/pytest3.log 2>&1; tail -6 gpurun_out/pytest3.log | cut -c1-300
for b in 128 256; do
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --batch-size $b > gpurun_out/bench_native_b$b.log 2>&1 && tail -1 gpurun_out/bench_native_b$b.log | cut -c1-200 || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_native3 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --batch-size 128 > $GRAFT_REPO_ROOT/gpurun_out/prof_native3.log 2>&1 || exit 1
