#!/bin/bash
# compact table of the benchmark's other modes on the final tree (one box, one process each)
mkdir -p gpurun_out
S=gpurun_out/final_table.txt; : > $S
run() { local name=$1; shift
  timeout -k 10 300 python3 bench.py "$@" --no-compare-stock 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-34s %10.1f img/s %8.3f ms/step  host %6.2f ms' % ('$name', d['value'], d['ms_per_step'], c['host_issue_ms_per_step']))" >> $S || { echo "FAILED $name"; exit 1; }
}
run "resnet50 fp16 O2 b400" --steps 30 --warmup 10
run "resnet50 fp16 O2 b400 --force-ddp" --steps 30 --warmup 10 --force-ddp
run "resnet50 bf16 O2 b400" --steps 30 --warmup 10 --precision bf16
run "resnet50 fp32 b400" --steps 10 --warmup 5 --precision fp32
run "resnet101 fp16 O2 b256" --steps 20 --warmup 8 --arch resnet101 --batch-size 256
run "resnet18 32x32 b800 fp16 eager" --steps 100 --warmup 20 --arch resnet18 --image-size 32 --batch-size 800
run "resnet18 32x32 b800 fp16 --graph" --steps 100 --warmup 20 --arch resnet18 --image-size 32 --batch-size 800 --graph
run "resnet18 32x32 b800 fp16 ddp --graph" --steps 100 --warmup 20 --arch resnet18 --image-size 32 --batch-size 800 --graph --force-ddp
run "resnet18 32x32 b800 fp32 --graph" --steps 100 --warmup 20 --arch resnet18 --image-size 32 --batch-size 800 --graph --precision fp32
cat $S
