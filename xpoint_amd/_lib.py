"""ctypes binding of libxpoint_hip.so (the C ABI declared in include/xpoint_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxpoint_hip.so")

_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_l = ctypes.c_int64


class XPointHipError(RuntimeError):
    pass


# name -> argtypes; every function returns int (0 ok) except xp_last_error / xp_version.
_SIGNATURES = {
    "xp_device_info": [c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.c_char_p, c_i],
    "xp_selective_scan_fwd": [c_p] * 9 + [c_i] * 7 + [c_p],
}


def exported_symbols():
    """Symbols include/xpoint_hip.h declares (parsed from the header; used by the CPU load test)."""
    import re
    hdr = os.path.join(_HERE, "..", "include", "xpoint_hip.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(xp_[a-z0-9_]+)\s*\(", txt)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XPointHipError(
            f"{LIB_PATH} not found: build it with `python -m xpoint_amd.build` (hipcc, gfx950). "
            "xpoint_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.xp_last_error.restype = ctypes.c_char_p
    lib.xp_last_error.argtypes = []
    lib.xp_version.restype = c_i
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_i
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().xp_last_error().decode("utf-8", "replace")
        raise XPointHipError(f"{what} failed (rc={rc}): {msg}")


def call(name: str, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) float32/int32 torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "xpoint_amd ops need contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
