#!/bin/bash
# launch-bound model families: eager vs whole-step hipGraph replay
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/zoo_graph.txt
for arch in mobilenet_v2 shufflenet_v2_x1_0 mnasnet1_0 squeezenet1_1 googlenet densenet121 resnet18; do
  timeout -k 10 200 python -u benchmarks/zoo_bench.py --archs $arch --out gpurun_out/zoo_graph.txt > gpurun_out/zoo_graph_$arch.log 2>&1 || { tail -n 12 gpurun_out/zoo_graph_$arch.log | cut -c1-300; exit 1; }
  timeout -k 10 200 python -u benchmarks/zoo_bench.py --archs $arch --graph --out gpurun_out/zoo_graph.txt > gpurun_out/zoo_graphed_$arch.log 2>&1 || { tail -n 12 gpurun_out/zoo_graphed_$arch.log | cut -c1-300; echo "$arch: graph capture failed" >> gpurun_out/zoo_graph.txt; }
  tail -n 2 gpurun_out/zoo_graph.txt
done
