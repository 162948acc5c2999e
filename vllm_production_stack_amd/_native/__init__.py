"""ctypes loader for the in-tree native libraries (built by ``_native/build.py``)."""

from __future__ import annotations

import ctypes
import threading
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_LOCK = threading.Lock()
_LIBS: dict[str, ctypes.CDLL] = {}


class NativeLibraryMissing(RuntimeError):
    pass


def lib_path(name: str) -> Path:
    return _HERE / f"lib{name}.so"


def have(name: str) -> bool:
    return lib_path(name).exists()


def load(name: str) -> ctypes.CDLL:
    """Load ``lib<name>.so`` from this directory; raises loudly when it has not been built."""
    with _LOCK:
        lib = _LIBS.get(name)
        if lib is None:
            p = lib_path(name)
            if not p.exists():
                raise NativeLibraryMissing(
                    f"{p} is missing: run `python -m vllm_production_stack_amd._native.build`"
                )
            lib = ctypes.CDLL(str(p))
            _LIBS[name] = lib
        return lib


def binary_path(name: str) -> Path:
    return _HERE / name
