#!/bin/bash
set -o pipefail
bash scripts/gpu_relufuse.sh && bash scripts/gpu_convfd.sh && bash scripts/gpu_dgf.sh | tail -n 24
