"""ctypes binding of ``libpda_runtime.so`` (host-side C++ runtime: rendezvous stores,
shared-memory collectives, reducer core, Horovod-style engine)."""
from __future__ import annotations

import ctypes
from pathlib import Path

_LIB = None
_PATH = Path(__file__).resolve().parent.parent / "lib" / "libpda_runtime.so"

c_p, c_i, c_ll, c_cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_char_p
P_ll = ctypes.POINTER(ctypes.c_longlong)
P_i = ctypes.POINTER(ctypes.c_int)

_SIGS = {
    # name: (restype, argtypes)
    "pda_store_tcp_create": (c_p, [c_cp, c_i, c_i, c_ll]),
    "pda_store_tcp_port": (c_i, [c_p]),
    "pda_store_file_create": (c_p, [c_cp, c_ll]),
    "pda_store_destroy": (None, [c_p]),
    "pda_store_set": (c_i, [c_p, c_cp, c_cp, c_i]),
    "pda_store_get": (c_ll, [c_p, c_cp, c_p, c_ll, c_ll]),
    "pda_store_add": (c_i, [c_p, c_cp, c_ll, P_ll]),
    "pda_store_check": (c_i, [c_p, c_cp]),
    "pda_store_delete": (c_i, [c_p, c_cp]),
    "pda_store_num_keys": (c_ll, [c_p]),
    "pda_store_compare_set": (c_ll, [c_p, c_cp, c_cp, c_i, c_cp, c_i, c_p, c_ll]),
    "pda_store_append": (c_ll, [c_p, c_cp, c_cp, c_i]),
    "pda_shm_create": (c_p, [c_cp, c_i, c_i, c_ll, c_ll]),
    "pda_shm_destroy": (None, [c_p]),
    "pda_shm_all_reduce_async": (c_ll, [c_p, c_p, c_ll, c_i, c_i]),
    "pda_shm_broadcast_async": (c_ll, [c_p, c_p, c_ll, c_i]),
    "pda_shm_all_gather_async": (c_ll, [c_p, c_p, c_p, c_ll]),
    "pda_shm_barrier_async": (c_ll, [c_p]),
    "pda_shm_wait": (c_i, [c_p, c_ll]),
    "pda_shm_poll": (c_i, [c_p, c_ll]),
    "pda_shm_abort": (None, [c_p]),
    "pda_reducer_create": (c_p, [c_i, P_ll, c_ll, c_ll, c_i, c_i]),
    "pda_reducer_destroy": (None, [c_p]),
    "pda_reducer_num_buckets": (c_i, [c_p]),
    "pda_reducer_bucket_bytes": (c_ll, [c_p, c_i]),
    "pda_reducer_bucket_params": (c_i, [c_p, c_i, P_i, c_i]),
    "pda_reducer_bucket_num_params": (c_i, [c_p, c_i]),
    "pda_reducer_bucket_of": (c_i, [c_p, c_i]),
    "pda_reducer_offset_of": (c_ll, [c_p, c_i]),
    "pda_reducer_reset": (None, [c_p]),
    "pda_reducer_mark_ready": (c_i, [c_p, c_i, P_i, c_i]),
    "pda_reducer_flush_unready": (c_i, [c_p, P_i, c_i]),
    "pda_reducer_num_unready": (c_i, [c_p]),
    "pda_reducer_record_arrival": (None, [c_p, c_i]),
    "pda_reducer_arrival": (c_i, [c_p, P_i, c_i]),
    "pda_reducer_rebuild": (None, [c_p, P_i, c_i]),
}
_OPTIONAL = {}


def register(sigs: dict) -> None:
    """Let other modules (hvd engine) add their symbols before/after load."""
    _OPTIONAL.update(sigs)
    if _LIB is not None:
        _bind(_LIB, sigs)


def _bind(l, sigs) -> None:
    for name, (res, args) in sigs.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args


def available() -> bool:
    return _PATH.exists()


def lib():
    global _LIB
    if _LIB is None:
        if not _PATH.exists():
            from .. import _build
            _build.build_runtime()
        l = ctypes.CDLL(str(_PATH))
        _bind(l, _SIGS)
        _bind(l, _OPTIONAL)
        _LIB = l
    return _LIB
