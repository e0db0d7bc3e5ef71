"""Per-step kernel-time table from a rocprofv3 kernel trace: the kernels between the last two
optimizer launches (one steady-state training step, after autotuning and warm-up)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "mt_sgd"
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
extra = {}
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 r.get("Queue_Id", "?") + "/" + r.get("Stream_Id", "?")))
    extra[(int(r["Start_Timestamp"]), r["Kernel_Name"])] = (r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                                                            r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
                                                            r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
# one step = from just after the optimizer launch of step n-2 to the optimizer launch of step n-1
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or marks[i + 1] != marks[i] + 1]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
agg = collections.defaultdict(lambda: [0, 0])
per_queue = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in step:
    agg[n][0] += e - s
    agg[n][1] += 1
    per_queue[q][0] += e - s
    per_queue[q][1] += 1
# time during which kernels of at least two different queues were running
ev = sorted([(s, 1, q) for s, e, n, q in step] + [(e, -1, q) for s, e, n, q in step])
live = collections.Counter()
last_t, overlap, union = ev[0][0], 0, 0
for t_, d, q in ev:
    nq = sum(1 for v in live.values() if v > 0)
    if nq >= 1:
        union += t_ - last_t
    if nq >= 2:
        overlap += t_ - last_t
    live[q] += d
    last_t = t_
busy = sum(v[0] for v in agg.values())
wall = step[-1][1] - step[0][0]
print(f"one step: {len(step)} launches, kernel time {busy / 1e6:.3f} ms, wall {wall / 1e6:.3f} ms, "
      f"device busy {union / 1e6:.3f} ms, kernels of two queues running together {overlap / 1e6:.3f} ms")
for q, (t_, c) in sorted(per_queue.items(), key=lambda kv: -kv[1][0]):
    names = collections.Counter(n.split("(")[0][:40] for s_, e_, n, qq in step if qq == q).most_common(3)
    print(f"  queue/stream {q}: {c} launches, {t_ / 1e6:.3f} ms  (mostly {', '.join(k for k, _ in names)})")
groups = collections.defaultdict(float)
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{t / 1e6:8.3f} ms {c:5d}  {n[:140]}")
    key = ("igemm dgrad+epilogue" if "igemm" in n and "ELb1E" in n and ("ELi1ELb1" in n or "ELi2ELb1" in n) else
           "igemm dgrad plain" if "igemm" in n and ("ELi1ELb0" in n or "ELi2ELb0" in n) else
           "igemm fprop" if "igemm" in n else "wgrad" if "wgrad" in n else
           "bn" if n.startswith("_Z") and "bn_" in n else "other")
    groups[key] += t
print({k: round(v / 1e6, 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])})

import os
if os.environ.get("PDA_TRACE_LAUNCHES"):
    # every launch of the step in issue order: start offset, duration, grid / workgroup size, registers, LDS, name
    with open(os.environ["PDA_TRACE_LAUNCHES"], "w") as out:
        t0 = step[0][0]
        for s_, e_, n, q in step:
            g, w, v, l = extra.get((s_, n), ("?", "?", "?", "?"))
            out.write(f"{(s_ - t0) / 1e3:10.1f} us  {(e_ - s_) / 1e3:8.1f} us  q{q:8s} grid {g:>9s} wg {w:>4s} vgpr {v:>4s} lds {l:>7s}  {n[:150]}\n")
