#!/bin/sh
# Launch commands of the six styles (counterparts of the reference's N.run.sh / start.sh).
# Select GPUs with HIP_VISIBLE_DEVICES=0,1,2,3 as the reference does with CUDA_VISIBLE_DEVICES.
python 1_dataparallel.py
python -m pytorch_distributed_amd.launch --nproc_per_node=4 --use_argv_local_rank 2_distributed.py
python 3_multiprocessing_distributed.py
python -m pytorch_distributed_amd.launch --nproc_per_node=4 --use_argv_local_rank 4_amp_distributed.py
python -m pytorch_distributed_amd.launch -np 4 -H localhost:4 --verbose 5_horovod_distributed.py
python -m pytorch_distributed_amd.launch -np 4 -H localhost:4 --verbose 5_2_horovod_mnist.py
# srun -N2 --gres gpu:8 python 6_distributed_slurm.py --dist-file dist_file
# python -m pytorch_distributed_amd.utils.smi --out distributed_log.csv --interval-ms 500   # statistics.sh
