"""Loader of the product library ``stan4bart_amd/csrc/libs4b.so`` (HIP, gfx950).

There is deliberately no fallback: if the library has not been built, or the process has no
MI355X-class device, creation fails loudly (the C layer raises "no HIP device available").
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libs4b.so")
_lib = None


def build_library(force: bool = False) -> str:
    """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC] + (["-B"] if force else [])
    subprocess.run(args, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB_PATH


def load_library() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("S4B_LIB_PATH", LIB_PATH)   # tuning builds of the same HIP sources (e.g. `make timing`)
        if path != LIB_PATH:
            _lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C stan4bart_amd/csrc` "
                "(or __graft_entry__.build()). The HIP path has no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    return _lib
