"""ctypes binding of libvc_hip.so, the C-ABI library of hand-written gfx950 kernels.

The library is loaded from this package directory (built in-tree by
``__graft_entry__.build()`` / ``make -C csrc``).  There is no fallback: if the shared
object is missing, or an entry point returns a non-zero status, a RuntimeError is
raised -- the product path never silently runs anything else.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VC_LIB_PATH") or os.path.join(_HERE, "libvc_hip.so")  # VC_LIB_PATH: an experimental build

VC_F32 = 0
VC_BF16 = 1

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libvc_hip.so not found at {LIB_PATH}; build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(there is no CPU or PyTorch fallback for the HIP path)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.vc_last_error.restype = ctypes.c_char_p
        _lib.vc_version.restype = ctypes.c_int
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().vc_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libvc_hip {what} failed (status {rc}): {msg}")


def dtype_code(dt):
    if dt == torch.float32:
        return VC_F32
    if dt == torch.bfloat16:
        return VC_BF16
    raise TypeError(f"libvc_hip supports float32 and bfloat16 compute types, not {dt}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    """Call entry point `name`; ints are passed as C int, floats as C float."""
    fn = getattr(lib(), name)
    cargs = []
    for a in args:
        if isinstance(a, bool):
            cargs.append(ctypes.c_int(int(a)))
        elif isinstance(a, int):
            cargs.append(ctypes.c_int(a))
        elif isinstance(a, float):
            cargs.append(ctypes.c_float(a))
        else:
            cargs.append(a)
    check(fn(*cargs), name)


def query(name, *args):
    """Call an entry point that returns a value (not a status code); ints are passed as C int."""
    fn = getattr(lib(), name)
    if name in ("vc_lstm_persist_xchg_bytes", "vc_lstm_persist_bwd_xchg_bytes", "vc_cbam_workspace_floats"):
        fn.restype = ctypes.c_longlong
    return fn(*[ctypes.c_int(int(a)) if isinstance(a, (bool, int)) else a for a in args])


def i64(v):
    return ctypes.c_int64(int(v))


def int_array(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])
