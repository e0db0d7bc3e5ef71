"""Loader of the optional CPython extension ``lib/_pda_pyext.<abi>.so`` (``csrc/pyext/ready_hook.cc``)."""
from __future__ import annotations

import importlib.machinery
import importlib.util
import os
import sysconfig
from pathlib import Path

_MOD = None
_TRIED = False


def module():
    """The extension module, or ``None`` when it is not built for this interpreter (or ``PDA_PYEXT=0``)."""
    global _MOD, _TRIED
    if _TRIED:
        return _MOD
    _TRIED = True
    if os.environ.get("PDA_PYEXT", "1") == "0":
        return None
    path = Path(__file__).resolve().parent.parent / "lib" / (
        "_pda_pyext" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if not path.exists():
        return None
    try:
        loader = importlib.machinery.ExtensionFileLoader("_pda_pyext", str(path))
        spec = importlib.util.spec_from_loader("_pda_pyext", loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        _MOD = mod
    except Exception:
        _MOD = None
    return _MOD
