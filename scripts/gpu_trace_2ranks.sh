#!/bin/bash
# kernel trace of rank 0 of a two-rank DDP run on ONE GPU (peer-memory all-reduce; both ranks are started by hand so
# that the profiled program is python itself): do the bucket all-reduce kernels run while backward kernels run?
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PDA_BENCH_SHARE_GPU=1 PDA_BENCH_STAGE_TIMEOUT=120
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 WORLD_SIZE=2
ARGS="--gpus 2 --comm p2p2 --batch-size 128 --steps 4 --warmup 3"
cd /tmp && export TMPDIR=/tmp
RANK=1 LOCAL_RANK=1 timeout -k 10 280 python3 -u $R/bench.py $ARGS > $R/gpurun_out/t2_rank1.log 2>&1 &
P1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/t2_prof -o r -- python3 $R/bench.py $ARGS > $R/gpurun_out/t2_rank0.log 2>&1; rc=$?
wait $P1; rc1=$?
tail -n 1 $R/gpurun_out/t2_rank0.log | cut -c1-300
[ $rc -eq 0 ] && [ $rc1 -eq 0 ] || { tail -n 15 $R/gpurun_out/t2_rank0.log; tail -n 15 $R/gpurun_out/t2_rank1.log; exit 1; }
python3 $R/scripts/trace_last_step.py $R/gpurun_out/t2_prof > $R/gpurun_out/t2_step.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/t2_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "mt_sgd" in r[2]]
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or marks[i + 1] != marks[i] + 1]
step = rows[ends[-2] + 1:ends[-1] + 1]
ar = [r for r in step if "p2p" in r[2] or "twoshot" in r[2] or "allreduce" in r[2].lower()]
other = [r for r in step if r not in ar]
tot = sum(e - s for s, e, *_ in ar)
ov = 0
for s, e, n, q in ar:
    cover = sorted((max(s, s2), min(e, e2)) for s2, e2, n2, q2 in other if q2 != q and s2 < e and e2 > s)
    cur = s
    for a, b in cover:
        if b > cur:
            ov += b - max(a, cur)
            cur = max(cur, b)
with open("$R/gpurun_out/t2_overlap.txt", "w") as out:
    print(f"rank 0, last step: {len(ar)} all-reduce kernel launches, {tot / 1e6:.3f} ms in total, of which {ov / 1e6:.3f} ms "
          f"while a compute kernel of another queue was running; queues: all-reduce {sorted(set(r[3] for r in ar))}, "
          f"others {sorted(set(r[3] for r in other))}", file=out)
    for s, e, n, q in ar:
        running = [n2.split('(')[0][:60] for s2, e2, n2, q2 in other if q2 != q and s2 < e and e2 > s][:3]
        print(f"  {n[:50]:50s} queue {q} {1e-3 * (e - s):8.1f} us  beside: {running}", file=out)
print(open("$R/gpurun_out/t2_overlap.txt").read())
PY
find $R/gpurun_out/t2_prof -name "*kernel_trace.csv" -delete
