mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -v --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest1.log 2>&1
rc=$?
echo "pytest rc=$rc"
tail -5 gpurun_out/pytest1.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python -u __graft_entry__.py smoke > gpurun_out/smoke1.log 2>&1 || { echo "smoke failed"; tail -20 gpurun_out/smoke1.log; exit 1; }
tail -2 gpurun_out/smoke1.log
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --batch-size 64 > gpurun_out/bench1.log 2>&1 || { echo "bench failed"; tail -20 gpurun_out/bench1.log; exit 1; }
tail -2 gpurun_out/bench1.log
