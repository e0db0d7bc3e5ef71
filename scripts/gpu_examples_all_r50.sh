#!/bin/bash
# every launch style on the benchmark's workload, one rank: does it run, and how fast
set -o pipefail
mkdir -p gpurun_out
A="--dataset imagenet --synthetic -a resnet50 -b 400 --epochs 1 --steps-per-epoch 40 -p 20 -j 4"
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/ex_$name.log 2>&1 || { echo "$name FAILED"; tail -n 15 gpurun_out/ex_$name.log | cut -c1-250; return 1; }; echo "$name: $(grep 'Epoch: \[0\]\[39/40\]' gpurun_out/ex_$name.log | grep -o 'Img/s *[0-9.]*' | head -1)  $(grep -c ' \* Acc@1' gpurun_out/ex_$name.log) eval"; }
L="python -u -m pytorch_distributed_amd.launch"
run 1_dp_fp16O2 python -u examples/1_dataparallel.py $A --precision fp16 --opt-level O2 &&
run 3_spawn_fp16O2 python -u examples/3_multiprocessing_distributed.py $A --precision fp16 --opt-level O2 &&
run 4_amp_O1 $L --nproc_per_node=1 --master_port 29551 --use_argv_local_rank examples/4_amp_distributed.py $A &&
run 4_amp_O2 $L --nproc_per_node=1 --master_port 29552 --use_argv_local_rank examples/4_amp_distributed.py $A --opt-level O2 &&
run 6_slurm_fp16O2 python -u examples/6_distributed_slurm.py $A --precision fp16 --opt-level O2 &&
run 2_ddp_bf16 $L --nproc_per_node=1 --master_port 29553 --use_argv_local_rank examples/2_distributed.py $A --precision bf16 --opt-level O2 &&
run 2_ddp_fp32 $L --nproc_per_node=1 --master_port 29554 --use_argv_local_rank examples/2_distributed.py $A
