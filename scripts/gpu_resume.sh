#!/bin/bash
# checkpoint + resume on the GPU path (amp O2, fused optimizer, DDP one rank)
set -o pipefail
mkdir -p /tmp/pda_resume && cd /tmp/pda_resume && rm -f *.pth.tar   # (checkpoints do not belong under gpurun_out/: it is copied back)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
A="--arch resnet18 --batch-size 128 --steps-per-epoch 30 --workers 0 --num-classes 10 -p 10 --synthetic --precision fp16 --opt-level O2 --lr 0.05 --seed 3"
L="python -u -m pytorch_distributed_amd.launch --nproc_per_node=1 --use_argv_local_rank"
timeout -k 10 200 $L --master_port 29571 $R/examples/2_distributed.py $A --epochs 2 > two_epochs.log 2>&1 || { tail -n 20 two_epochs.log | cut -c1-200; exit 1; }
grep "Epoch: \[1\]\[29/30\]\| \* Acc" two_epochs.log | cut -c1-160
rm -f *.pth.tar
timeout -k 10 200 $L --master_port 29572 $R/examples/2_distributed.py $A --epochs 1 > first.log 2>&1 || { tail -n 20 first.log | cut -c1-200; exit 2; }
ls *.pth.tar
timeout -k 10 200 $L --master_port 29573 $R/examples/2_distributed.py $A --epochs 2 --resume resnet18.checkpoint.pth.tar > second.log 2>&1 || { tail -n 20 second.log | cut -c1-200; exit 3; }
grep "resumed\|Epoch: \[1\]\[29/30\]\| \* Acc" second.log | cut -c1-160
