"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/dexsim.h declares, its
struct sizes match the Python mirrors, and its tables agree with the host-side configuration tables.
No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

from dexrobot_isaac_amd import _abi, _lib
from dexrobot_isaac_amd.build import build_lib
from dexrobot_isaac_amd.config import OBS_KEYS, REWARD_TERMS, build_sim_config, default_cfg
from dexrobot_isaac_amd.hand_model import BODY_NAMES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build_lib()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dexsim.h")).read()
    declared = set(re.findall(r"\b(dexsim_[a-z_]+)\s*\(", hdr))
    assert declared == set(_abi.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_struct_sizes_match(lib):
    _abi.check_struct_sizes(lib)


def test_tables_match_host_config(lib):
    off = 0
    for i, (name, dim) in enumerate(OBS_KEYS):
        n, o, d = C.c_char_p(), C.c_int(), C.c_int()
        assert lib.dexsim_obs_key_info(i, C.byref(n), C.byref(o), C.byref(d)) == 0
        assert (n.value.decode(), o.value, d.value) == (name, off, dim)
        off += dim
    assert off == _abi.OBS_ALL_DIM
    assert lib.dexsim_obs_key_info(len(OBS_KEYS), None, None, None) != 0
    for i, name in enumerate(REWARD_TERMS):
        n = C.c_char_p()
        assert lib.dexsim_reward_term_name(i, C.byref(n)) == 0 and n.value.decode() == name
    for i, name in enumerate(BODY_NAMES):
        n = C.c_char_p()
        assert lib.dexsim_body_name(i, C.byref(n)) == 0 and n.value.decode() == name


@pytest.mark.parametrize("task,nobs", [("BlindGrasping", 158), ("BaseTask", 224)])
def test_layout_and_obs_dims(lib, task, nobs):
    cfg = default_cfg(task)
    cfg["env"]["numEnvs"] = 100
    sc, _ = build_sim_config(cfg)
    assert sc.num_obs == nobs and sc.num_actions == 18
    fields = (_abi.DexSimField * 128)()
    nf, words = C.c_int(), C.c_size_t()
    assert lib.dexsim_arena_layout(C.byref(sc), fields, 128, C.byref(nf), C.byref(words)) == 0
    names = [fields[i].name.decode() for i in range(nf.value)]
    assert len(set(names)) == nf.value and {"q", "qd", "obs_all", "crow", "reset_flag"} <= set(names)
    assert words.value == sum(fields[i].rows for i in range(nf.value)) * 128   # stride padded to 64
    assert lib.dexsim_arena_layout(C.byref(sc), fields, 3, C.byref(nf), C.byref(words)) == _abi.__dict__.get("ERR_LAYOUT", 5)


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    sc, model = build_sim_config(default_cfg("BlindGrasping"))
    h = C.c_void_p()
    ms = model.to_struct()
    rc = lib.dexsim_create(C.byref(sc), C.byref(ms), 0, C.byref(h))
    assert rc == 4 and b"HIP device" in lib.dexsim_last_error()
    from dexrobot_isaac_amd.core import DexSimCore
    from dexrobot_isaac_amd._lib import DexSimError
    with pytest.raises(DexSimError):
        DexSimCore(sc, ms, "cpu")
    with pytest.raises(DexSimError):
        DexSimCore(sc, ms, "cuda:0")


def test_bad_arguments_are_rejected(lib):
    sc, model = build_sim_config(default_cfg("BaseTask"))
    ms = model.to_struct()
    h = C.c_void_p()
    assert lib.dexsim_create(None, C.byref(ms), 0, C.byref(h)) == 1
    ms.jtype[4] = 0
    assert lib.dexsim_create(C.byref(sc), C.byref(ms), 0, C.byref(h)) == 1
    assert b"prismatic" in lib.dexsim_last_error()
    assert lib.dexsim_step(None, None, None) == 1
