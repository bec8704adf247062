"""ctypes wrapper around the CPU oracle (oracle/dexsim_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force=False):
    """Compile the oracle with gcc (fp32 + fp64 variants)."""
    so = os.path.join(_BUILD, "liboracle.so")
    src = os.path.join(_HERE, "dexsim_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "dexsim.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return so


def _load(f64=False):
    build()
    lib = C.CDLL(os.path.join(_BUILD, "liboracle_f64.so" if f64 else "liboracle.so"))
    vp = C.c_void_p
    lib.orc_create.restype = vp
    lib.orc_create.argtypes = [vp, vp]
    for name, args in {
        "orc_destroy": [vp], "orc_set_threads": [vp, C.c_int], "orc_set_reset_samples": [vp, vp],
        "orc_init_state": [vp], "orc_process_actions": [vp, vp, C.c_int], "orc_physics_step": [vp],
        "orc_substep": [vp, C.c_int], "orc_publish": [vp], "orc_post_physics": [vp, C.c_int],
        "orc_step": [vp, vp], "orc_reset_idx": [vp, vp, C.c_int], "orc_reset": [vp],
        "orc_get_obs_buf": [vp, vp], "orc_get_stats": [vp, vp], "orc_set_rc_first_call": [vp, C.c_int],
        "orc_compute_observations": [vp], "orc_l2_step_no_reset": [vp], "orc_reset_flagged_no_physics": [vp],
        "orc_fk": [vp, vp, vp], "orc_mass_matrix": [vp, vp, vp, vp, vp],
    }.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = None
    lib.orc_get_field.argtypes = [vp, C.c_char_p, vp]
    lib.orc_set_field.argtypes = [vp, C.c_char_p, vp]
    lib.orc_field_rows.argtypes = [C.c_char_p]
    lib.orc_get_contacts.argtypes = [vp, C.c_int, vp]
    lib.orc_any_reset.argtypes = [vp]
    lib.orc_field_name.restype = C.c_char_p
    lib.orc_field_name.argtypes = [C.c_int]
    return lib


class Oracle:
    """One oracle instance = N envs stepped on the CPU."""

    def __init__(self, sim_cfg, model_struct, f64=False, threads=0):
        self.lib = _load(f64)
        self.cfg = sim_cfg
        self.n = sim_cfg.num_envs
        self.num_obs = sim_cfg.num_obs
        self.num_actions = sim_cfg.num_actions
        self.h = self.lib.orc_create(C.byref(sim_cfg), C.byref(model_struct))
        if threads:
            self.lib.orc_set_threads(self.h, threads)
        self.lib.orc_init_state(self.h)

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass

    # -- state exchange: SoA [rows][N] float64
    def fields(self):
        return [self.lib.orc_field_name(i).decode() for i in range(self.lib.orc_num_fields())]

    def get(self, name):
        rows = self.lib.orc_field_rows(name.encode())
        if rows < 0:
            raise KeyError(name)
        out = np.zeros((rows, self.n), dtype=np.float64)
        self.lib.orc_get_field(self.h, name.encode(), out.ctypes.data)
        return out

    def set(self, name, value):
        rows = self.lib.orc_field_rows(name.encode())
        if rows < 0:
            raise KeyError(name)
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float64), (rows, self.n)))
        self.lib.orc_set_field(self.h, name.encode(), v.ctypes.data)

    def contacts(self, env):
        out = np.zeros((64, 10), dtype=np.float64)     # >= DEXSIM_KMAX rows
        k = self.lib.orc_get_contacts(self.h, env, out.ctypes.data)
        return out[:k]

    def obs_buf(self):
        out = np.zeros((self.n, self.num_obs), dtype=np.float32)
        self.lib.orc_get_obs_buf(self.h, out.ctypes.data)
        return out

    def stats(self):
        out = np.zeros(64, dtype=np.float32)
        self.lib.orc_get_stats(self.h, out.ctypes.data)
        return out

    def set_reset_samples(self, samples):
        if samples is None:
            self.lib.orc_set_reset_samples(self.h, None)
        else:
            s = np.ascontiguousarray(samples, dtype=np.float32)
            assert s.shape == (self.n, 29)
            self.lib.orc_set_reset_samples(self.h, s.ctypes.data)

    # -- pipeline
    def _act(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        assert a.shape == (self.n, self.num_actions), a.shape
        return a

    def init_state(self):
        self.lib.orc_init_state(self.h)

    def process_actions(self, actions, zero_targets=False):
        a = self._act(actions)
        self.lib.orc_process_actions(self.h, a.ctypes.data, int(zero_targets))

    def physics_step(self):
        self.lib.orc_physics_step(self.h)

    def substep(self, last=False):
        self.lib.orc_substep(self.h, int(last))

    def publish(self):
        self.lib.orc_publish(self.h)

    def post_physics(self, obs_only=False):
        self.lib.orc_post_physics(self.h, int(obs_only))

    def step(self, actions):
        a = self._act(actions)
        self.lib.orc_step(self.h, a.ctypes.data)
        return self.obs_buf(), self.get("rew")[0].astype(np.float32), self.get("reset_flag")[0].astype(bool)

    def reset(self):
        self.lib.orc_reset(self.h)
        return self.obs_buf()

    def reset_idx(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        self.lib.orc_reset_idx(self.h, ids.ctypes.data, len(ids))

    def compute_observations(self):
        self.lib.orc_compute_observations(self.h)

    def l2_step_no_reset(self):
        self.lib.orc_l2_step_no_reset(self.h)

    def reset_flagged_no_physics(self):
        self.lib.orc_reset_flagged_no_physics(self.h)

    def set_rc_first_call(self, v):
        self.lib.orc_set_rc_first_call(self.h, int(v))

    def fk(self, q):
        out = np.zeros((26, 6))
        qq = np.ascontiguousarray(q, dtype=np.float64)
        self.lib.orc_fk(self.h, qq.ctypes.data, out.ctypes.data)
        return out

    def mass_matrix(self, q, qd):
        M = np.zeros((26, 26))
        b = np.zeros(26)
        qq = np.ascontiguousarray(q, dtype=np.float64)
        qv = np.ascontiguousarray(qd, dtype=np.float64)
        self.lib.orc_mass_matrix(self.h, qq.ctypes.data, qv.ctypes.data, M.ctypes.data, b.ctypes.data)
        return M, b
