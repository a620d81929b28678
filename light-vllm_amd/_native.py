"""Loader of the native libraries.  No fallback: if the HIP library or the torch
binding is missing or fails to load, importing this module raises, so a product
path can never silently run on something else."""
import ctypes
import os

import torch  # must be first: its HIP runtime (same SONAME) is the one the kernels use

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
LIB_PATH = os.path.join(_LIBDIR, "liblvllm_hip.so")
EXT_PATH = os.path.join(_LIBDIR, "_C.so")

_state = {"lib": None, "ext": False}


class NativeLibraryError(ImportError):
    pass


def load_hip_library() -> ctypes.CDLL:
    """ctypes handle on the C-ABI (include/lvllm_hip.h)."""
    if _state["lib"] is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        lib.lvllm_last_error.restype = ctypes.c_char_p
        lib.lvllm_version.restype = ctypes.c_char_p
        _state["lib"] = lib
    return _state["lib"]


def load_torch_ops() -> None:
    """Registers torch.ops._C / _C_cache_ops / _C_cuda_utils (reference schemas)."""
    if not _state["ext"]:
        load_hip_library()
        if not os.path.exists(EXT_PATH):
            raise NativeLibraryError(f"{EXT_PATH} not found: run __graft_entry__.build()")
        torch.ops.load_library(EXT_PATH)
        _state["ext"] = True


def loaded_libraries():
    return {"hip": LIB_PATH if _state["lib"] is not None else None,
            "torch_ops": EXT_PATH if _state["ext"] else None}
