"""Host runtime under ThreadSanitizer and AddressSanitizer+UBSan: `scripts/sanitize_runtime.sh` builds
the C++ runtime sources with their stress driver (ranks = threads of one process: TCP store traffic,
shared-memory collectives, reducer bookkeeping, the Horovod-style engine with per-rank enqueue orders)
and fails on any report or wrong result."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_runtime_is_clean_under_tsan_and_asan(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize_runtime.sh"), "4"], env=env,
                       capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    if "cannot find -ltsan" in tail or "cannot find -lasan" in tail:
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0, tail
    assert "sanitizers clean" in r.stdout
