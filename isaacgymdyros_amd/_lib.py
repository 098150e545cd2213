"""Loads libdyroswalk_hip.so.  There is no fallback: if the HIP library is missing or lacks a symbol of
include/dyros_walk.h, importing the product path raises."""
from __future__ import annotations

import ctypes as C
import os

from . import abi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdyroswalk_hip.so")
_cached = None


class DyrosWalkLibraryError(RuntimeError):
    pass


def load():
    global _cached
    if _cached is None:
        if not os.path.exists(LIB_PATH):
            raise DyrosWalkLibraryError(
                "%s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). This package has no CPU or PyTorch fallback." % LIB_PATH)
        # One HIP runtime per process: torch ships its own libamdhip64 and owns the device memory and streams this
        # library is handed, so torch's copy must be the one already mapped when libdyroswalk_hip.so resolves its
        # DT_NEEDED entry (same SONAME -> the loader reuses it).  Loading in the other order maps a second runtime
        # that sees no device.
        import torch
        tlib = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(tlib):
            C.CDLL(tlib, mode=C.RTLD_GLOBAL)
        lib = C.CDLL(LIB_PATH)
        try:
            api = abi.declare(lib, "dw_")
        except AttributeError as e:
            raise DyrosWalkLibraryError("libdyroswalk_hip.so does not export the C-ABI of include/dyros_walk.h: %s" % e)
        if api["abi_version"]() != abi.K["DW_ABI_VERSION"]:
            raise DyrosWalkLibraryError("libdyroswalk_hip.so ABI version mismatch; rebuild it")
        _cached = (lib, api)
    return _cached


def check(api, rc):
    if rc != 0:
        raise DyrosWalkLibraryError("dyroswalk: %s (code %d)" % (api["last_error"]().decode(), rc))
