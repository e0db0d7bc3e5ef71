#!/bin/bash
# one conv layer, several tile configurations back to back: OL_ARGS="--h 14 --cin 256 --cout 256 --k 3 --batch 256" OL_CFGS="1 5 7"
set -o pipefail
for op in ${OL_OPS:-fprop dgrad}; do
  for cfg in ${OL_CFGS:-1 5 7 135 64}; do
    timeout -k 10 60 python -u benchmarks/one_layer.py --op $op ${OL_ARGS:---h 14 --cin 256 --cout 256 --k 3 --batch 256} --cfg $cfg --warmup 20 --iters 20 2>&1 | tail -1 || exit 1
  done
done
