#!/bin/bash
# Build the host runtime (stores, shm collectives, reducer core, hvd engine) together with its stress
# driver under ThreadSanitizer and under AddressSanitizer+UBSan and run both (CPU only, ranks = threads).
#   scripts/sanitize_runtime.sh [ranks]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pytorch_distributed_amd/csrc/runtime
OUT=${TMPDIR:-/tmp}/pda_sanitize
RANKS=${1:-4}
mkdir -p "$OUT"
COMMON="-O1 -g -std=c++17 -pthread -fno-omit-frame-pointer -I$SRC"
g++ $COMMON -fsanitize=thread -o "$OUT/stress_tsan" "$SRC"/*.cc "$SRC"/tests/stress.cc -lrt
g++ $COMMON -fsanitize=address,undefined -fno-sanitize-recover=undefined -o "$OUT/stress_asan" "$SRC"/*.cc "$SRC"/tests/stress.cc -lrt
echo "== tsan"
TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1" "$OUT/stress_tsan" "$RANKS"
echo "== asan+ubsan"
ASAN_OPTIONS="detect_leaks=1" "$OUT/stress_asan" "$RANKS"
echo "sanitizers clean"
