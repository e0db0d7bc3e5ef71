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
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
# one step = from just after the optimizer launch of step n-2 to the optimizer launch of step n-1
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or marks[i + 1] != marks[i] + 1]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in step:
    agg[n][0] += e - s
    agg[n][1] += 1
busy = sum(v[0] for v in agg.values())
wall = step[-1][1] - step[0][0]
print(f"one step: {len(step)} launches, kernel time {busy / 1e6:.3f} ms, wall {wall / 1e6:.3f} ms")
groups = collections.defaultdict(float)
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{t / 1e6:8.3f} ms {c:5d}  {n[:140]}")
    key = ("igemm dgrad+epilogue" if "igemm" in n and "ELb1E" in n and ("ELi1ELb1" in n or "ELi2ELb1" in n) else
           "igemm dgrad plain" if "igemm" in n and ("ELi1ELb0" in n or "ELi2ELb0" in n) else
           "igemm fprop" if "igemm" in n else "wgrad" if "wgrad" in n else
           "bn" if n.startswith("_Z") and "bn_" in n else "other")
    groups[key] += t
print({k: round(v / 1e6, 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])})
