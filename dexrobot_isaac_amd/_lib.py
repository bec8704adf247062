"""Loader for libdexsim.so.  There is NO fallback: if the HIP library is missing or its ABI drifted the
product fails loudly (a CPU path would void every parity claim)."""
import ctypes
import os

from . import _abi

_LIB = None
# DEXSIM_LIB_PATH: diagnostics only (A/B of kernel variants, the phase-stamp build); it still has to be a libdexsim build
LIB_PATH = os.environ.get("DEXSIM_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdexsim.so")


class DexSimError(RuntimeError):
    pass


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DexSimError(
                f"{LIB_PATH} not found. Build it with `python -m dexrobot_isaac_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for sym in _abi.EXPORTED_SYMBOLS:
            if not hasattr(lib, sym):
                raise DexSimError(f"libdexsim.so does not export {sym}")
        _abi.declare_prototypes(lib)
        _abi.check_struct_sizes(lib)
        _LIB = lib
    return _LIB


def check(rc, what=""):
    if rc != 0:
        lib = load()
        msg = lib.dexsim_last_error().decode() or lib.dexsim_error_string(rc).decode()
        raise DexSimError(f"{what}: {msg} (code {rc})")
