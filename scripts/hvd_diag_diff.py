"""Compare the per-rank records written with PDA_HVD_DIAG=<dir> (hvd/mpi_ops.py::_Diag).

For every fused device group: member order / offsets must match on all ranks, the checksums taken after the
all-reduce (b) and after the unpack (c) must be identical on all ranks, and the gradients handed to the wrapped
optimizer step (d) must equal (c).  Prints the first stage at which ranks disagree, per tensor.
usage: python scripts/hvd_diag_diff.py <dir>"""
import glob
import json
import os
import sys


def load(path):
    groups, steps = [], []
    for line in open(path):
        rec = json.loads(line)
        (steps if "step_grads" in rec else groups).append(rec)
    return groups, steps


def main(d):
    files = sorted(glob.glob(os.path.join(d, "rank*.jsonl")))
    if len(files) < 2:
        print(f"need at least two rank files in {d}, found {len(files)}")
        return 2
    data = [load(f) for f in files]
    g0, s0 = data[0]
    bad = 0
    for r, (g, s) in enumerate(data[1:], 1):
        if len(g) != len(g0):
            print(f"rank {r}: {len(g)} groups, rank 0: {len(g0)}")
            bad += 1
        for a, b in zip(g0, g):
            if a["names"] != b["names"] or a["offsets"] != b["offsets"]:
                print(f"group {a['group']}: member order / offsets differ: {a['names']} vs {b['names']}")
                bad += 1
                continue
            for stage in ("b_reduced", "c_unpacked"):
                for name, x, y in zip(a["names"], a[stage], b[stage]):
                    if x != y:
                        print(f"group {a['group']} {name}: stage {stage} differs between rank 0 and rank {r}: {x!r} {y!r}")
                        bad += 1
        for i, (a, b) in enumerate(zip(s0, s)):
            for name, x in a["step_grads"].items():
                y = b["step_grads"].get(name)
                if x != y:
                    print(f"step {i} {name}: gradient handed to step() differs between rank 0 and rank {r}: {x!r} {y!r}")
                    bad += 1
    # (c) of the group that carried a tensor in a step against what step() saw, per rank
    for r, (g, s) in enumerate(data):
        for st in s:
            seen = {}
            for rec in g:
                if rec.get("step") == st.get("step"):
                    for name, c in zip(rec["names"], rec["c_unpacked"]):
                        seen[name.split("allreduce.", 1)[-1]] = c
            for name, v in st["step_grads"].items():
                c = seen.get(name)
                if c is not None and c != v:
                    print(f"rank {r} step {st.get('step')} {name}: unpacked {c!r} but step() saw {v!r} (written after the reduction)")
                    bad += 1
    print(f"{len(g0)} groups, {len(s0)} steps on {len(files)} ranks: {'no differences' if not bad else str(bad) + ' differences'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
