"""Build libdexsim.so (HIP, gfx950 only) in-tree with hipcc.  No JIT cache: the .so sits next to the package
so that it travels with the repo snapshot and shows up as loaded native code."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdexsim.so")
SOURCES = ["dexsim.hip", "dexsim_device.h", "dexsim_physics.hip.inc", "dexsim_l2.hip.inc"]


DEBUG_SPIN_LIB = os.path.join(HERE, "libdexsim_dbgspin.so")


def _stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", "dexsim.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_debug_spin(force=False, bound=2000000):
    """Diagnostic twin of the library with every LDS token wait bounded (-DDEXSIM_DEBUG_SPIN=<polls>, see
    dexsim_physics.hip.inc): a sequencing bug shows up as a recorded word in the counters block instead of a hung GPU.
    Loaded only by tests/ (DEXSIM_LIB_PATH); the product never uses it."""
    return build_lib(force=force, defs=(f"-DDEXSIM_DEBUG_SPIN={int(bound)}",), out=DEBUG_SPIN_LIB)


def build_lib(force=False, verbose=False, defs=(), out=None):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Returns the path of the library."""
    lib = out or LIB
    if not force and not _stale(lib):
        return lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libdexsim.so cannot be built on this machine")
    # -fno-slp-vectorize: the auto-vectoriser's v_pk_* pairs cost more v_mov shuffles and wait states than they
    # save here (13.8k -> 13.4k instructions, scratch 112 -> 48 B in k_substep); packed FP32 is written by hand
    # (f2) where it pays.
    # -fno-hip-fp32-correctly-rounded-divide-sqrt: fp32 `/` and sqrtf as the <= 2.5 ulp sequences (v_rcp / v_rsq + one
    # Newton step) instead of the correctly rounded ones (v_div_scale / v_div_fmas / v_div_fixup, ~10 instructions each): the
    # step kernel has ~60 divisions per wave and sub-step and is VALU-issue bound on the SIMDs that carry two finger waves
    # (75.4 -> 69.9 us per control step).  Still fp32 arithmetic; every parity test (golden vectors of the reference at 2e-5 /
    # 2e-6, oracle at the fp64-derived tolerance) holds unchanged -- 2.5 ulp is 3e-7 relative.
    # -mllvm -amdgpu-sched-strategy=max-ilp: the step kernel is a few long dependency chains on waves that are (almost) alone on
    # their SIMD, so the scheduler should interleave independent chains rather than minimise register pressure: same VGPR
    # counts, still no spills / scratch, 61.3 -> 62.2 M env-steps/s (contact-rich regime 187 -> 189 us: within 1 %).
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
           "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fPIC", "-shared", "-std=c++17", *defs,
           "-o", lib, os.path.join(CSRC, "dexsim.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return lib


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
