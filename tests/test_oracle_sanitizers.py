"""SURVEY.md section 5 "race detection / sanitizers": the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer
(`make -C oracle asan`: oracle + a C driver in one sanitized executable; GPU ASan is not available on the pool).  The
driver runs reset, control steps with random actions, an explicit reset_idx, the general contact path and the stand-alone
stage entry points; any report makes the executable exit non-zero."""
import ctypes as C
import os
import subprocess

import pytest

from dexrobot_isaac_amd.config import build_sim_config, default_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("task,exe", [("BlindGrasping", "oracle_asan"), ("BaseTask", "oracle_asan"), ("BlindGrasping", "oracle_asan_f64")])
def test_oracle_clean_under_asan_ubsan(tmp_path, task, exe):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    cfg = default_cfg(task)
    cfg["env"]["numEnvs"] = 9
    cfg["env"]["episodeLength"] = 12                       # time-outs -> in-step resets inside the run
    sc, model = build_sim_config(cfg, dr={"mass": (0.05, 0.2), "friction": (0.5, 1.5), "seed": 7} if task == "BlindGrasping" else None)
    blob = tmp_path / "structs.bin"
    ms = model.to_struct()
    blob.write_bytes(bytes(C.string_at(C.byref(sc), C.sizeof(sc))) + bytes(C.string_at(C.byref(ms), C.sizeof(ms))))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", exe), str(blob), "40"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "asan driver ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr
