"""In-tree native build for gfx950 (MI355X).

Two shared libraries are produced under ``pytorch_distributed_amd/lib``:

* ``libpda_kernels.so`` - every ``csrc/kernels/*.hip`` file compiled with
  ``hipcc --offload-arch=gfx950`` (device code is cross-compiled, no GPU needed)
  and exported through a plain C ABI that ``ops/_lib.py`` binds with ctypes.
* ``libpda_runtime.so`` - the host-side C++ runtime (``csrc/runtime/*.cc``:
  rendezvous stores, shared-memory collective backend, reducer core, Horovod-style
  background engine), compiled with g++.

* ``_pda_pyext.<abi>.so`` - a small CPython extension (``csrc/pyext/*.cc``): the per-parameter gradient-ready
  hook of ``DistributedDataParallel`` as a C callable (optional: the Python hooks are the fallback).

``python -m pytorch_distributed_amd._build`` rebuilds whatever is out of date.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "lib"
OBJ = PKG / "build"
ARCH = "gfx950"

HIPCC = os.environ.get("PDA_HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXX = os.environ.get("PDA_CXX") or shutil.which("g++") or "g++"

KERNEL_LIB = LIB / "libpda_kernels.so"
RUNTIME_LIB = LIB / "libpda_runtime.so"
import sysconfig  # noqa: E402
PYEXT_LIB = LIB / ("_pda_pyext" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _stamp(paths) -> str:
    h = hashlib.sha1()
    for p in sorted(paths):
        h.update(str(p.name).encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _run(cmd) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed:\n  %s\n%s\n%s" % (" ".join(map(str, cmd)), r.stdout, r.stderr))


def _needs(target: Path, stamp: str) -> bool:
    sf = target.with_suffix(target.suffix + ".stamp")
    return not (target.exists() and sf.exists() and sf.read_text() == stamp)


def _mark(target: Path, stamp: str) -> None:
    target.with_suffix(target.suffix + ".stamp").write_text(stamp)


def build_kernels(force: bool = False, jobs: int | None = None, verbose: bool = False) -> Path:
    srcs = sorted((CSRC / "kernels").glob("*.hip"))
    hdrs = sorted((CSRC / "kernels").glob("*.h"))
    stamp = _stamp(srcs + hdrs)
    if not force and not _needs(KERNEL_LIB, stamp):
        return KERNEL_LIB
    LIB.mkdir(exist_ok=True)
    OBJ.mkdir(exist_ok=True)
    jobs = jobs or min(8, os.cpu_count() or 4, int(os.environ.get("MAX_JOBS", "8")))
    hstamp = _stamp(hdrs)

    def one(src: Path) -> Path:
        obj = OBJ / (src.stem + ".o")
        st = _stamp([src]) + hstamp
        if force or _needs(obj, st):
            if verbose:
                print("[pda build] hipcc", src.name, flush=True)
            _run([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                  "-I", str(CSRC / "kernels"), "-c", str(src), "-o", str(obj)])
            _mark(obj, st)
        return obj

    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(one, srcs))
    _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(KERNEL_LIB)] + [str(o) for o in objs])
    _mark(KERNEL_LIB, stamp)
    return KERNEL_LIB


def build_variant(name: str, defines, verbose: bool = False) -> Path:
    """Experiment build: the kernel library compiled with extra ``-D`` flags into
    ``lib/libpda_kernels_<name>.so`` (select it at run time with ``PDA_KERNELS_LIB=<path>``), so two
    code-generation choices can be timed against each other inside one GPU session."""
    srcs = sorted((CSRC / "kernels").glob("*.hip"))
    out = LIB / f"libpda_kernels_{name}.so"
    objdir = PKG / f"build_{name}"
    objdir.mkdir(exist_ok=True)
    LIB.mkdir(exist_ok=True)

    def one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        if verbose:
            print("[pda build]", name, src.name, flush=True)
        _run([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
              "-I", str(CSRC / "kernels"), *[f"-D{d}" for d in defines], "-c", str(src), "-o", str(obj)])
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(one, srcs))
    _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(out)] + [str(o) for o in objs])
    return out


def build_runtime(force: bool = False, verbose: bool = False) -> Path | None:
    srcs = sorted((CSRC / "runtime").glob("*.cc"))
    hdrs = sorted((CSRC / "runtime").glob("*.h"))
    if not srcs:
        return None
    stamp = _stamp(srcs + hdrs)
    if not force and not _needs(RUNTIME_LIB, stamp):
        return RUNTIME_LIB
    LIB.mkdir(exist_ok=True)
    if verbose:
        print("[pda build] g++ runtime", flush=True)
    _run([CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fvisibility=hidden",
          "-I", str(CSRC / "runtime"), "-o", str(RUNTIME_LIB)] + [str(s) for s in srcs] + ["-lrt"])
    _mark(RUNTIME_LIB, stamp)
    return RUNTIME_LIB


def build_pyext(force: bool = False, verbose: bool = False) -> Path | None:
    """The CPython extension; skipped (None) where the interpreter's headers are not installed."""
    srcs = sorted((CSRC / "pyext").glob("*.cc"))
    inc = sysconfig.get_paths().get("include")
    if not srcs or not inc or not (Path(inc) / "Python.h").exists():
        return None
    stamp = _stamp(srcs)
    if not force and not _needs(PYEXT_LIB, stamp):
        return PYEXT_LIB
    LIB.mkdir(exist_ok=True)
    if verbose:
        print("[pda build] g++ pyext", flush=True)
    _run([CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I", inc, "-o", str(PYEXT_LIB)]
         + [str(s) for s in srcs])
    _mark(PYEXT_LIB, stamp)
    return PYEXT_LIB


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_runtime(force=force, verbose=verbose)
    build_pyext(force=force, verbose=verbose)
    build_kernels(force=force, verbose=verbose)


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m pytorch_distributed_amd._build --variant NAME DEFINE[=V] ...
        i = sys.argv.index("--variant")
        print("built:", build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
        sys.exit(0)
    build_all(force="--force" in sys.argv, verbose=True)
    print("built:", KERNEL_LIB, RUNTIME_LIB if RUNTIME_LIB.exists() else "(no runtime sources yet)")
