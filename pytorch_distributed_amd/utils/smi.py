"""GPU utilisation / memory poller: the rocm-smi analogue of the reference's
``nvidia-smi --query-gpu=timestamp,index,memory.*,utilization.* -lms 500 -f x.csv``
(/root/reference/statistics.sh:1-4).  Run next to a job:

    python -m pytorch_distributed_amd.utils.smi --out distributed_log.csv --interval-ms 500
"""
from __future__ import annotations

import argparse
import csv
import glob
import os
import subprocess
import time


def _sysfs_cards():
    cards = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "gpu_busy_percent")):
            cards.append(d)
    return cards


def _read(path, default=""):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return default


def sample():
    """One row per GPU: (index, mem_total, mem_used, mem_free, util_gpu%, util_mem%)."""
    rows = []
    for i, d in enumerate(_sysfs_cards()):
        total = int(_read(os.path.join(d, "mem_info_vram_total"), "0") or 0)
        used = int(_read(os.path.join(d, "mem_info_vram_used"), "0") or 0)
        busy = _read(os.path.join(d, "gpu_busy_percent"), "0")
        mbusy = _read(os.path.join(d, "mem_busy_percent"), "0")
        rows.append((i, total >> 20, used >> 20, (total - used) >> 20, busy, mbusy))
    if rows:
        return rows
    try:  # fall back to the CLI
        out = subprocess.run(["rocm-smi", "--showuse", "--showmemuse", "--csv"], capture_output=True, text=True,
                             timeout=10).stdout
        for line in out.splitlines()[1:]:
            parts = line.split(",")
            if len(parts) >= 3:
                rows.append((parts[0], "", "", "", parts[1], parts[2]))
    except (OSError, subprocess.SubprocessError):
        pass
    return rows


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpu_log.csv")
    ap.add_argument("--interval-ms", type=int, default=500)
    ap.add_argument("--duration-s", type=float, default=0.0, help="0 = until interrupted")
    a = ap.parse_args(argv)
    t_end = time.time() + a.duration_s if a.duration_s > 0 else float("inf")
    with open(a.out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["timestamp", "index", "memory.total [MiB]", "memory.used [MiB]", "memory.free [MiB]",
                    "utilization.gpu [%]", "utilization.memory [%]"])
        try:
            while time.time() < t_end:
                ts = time.strftime("%Y/%m/%d %H:%M:%S") + ".%03d" % int((time.time() % 1) * 1000)
                for row in sample():
                    w.writerow([ts, *row])
                f.flush()
                time.sleep(a.interval_ms / 1000.0)
        except KeyboardInterrupt:
            pass


if __name__ == "__main__":
    main()
