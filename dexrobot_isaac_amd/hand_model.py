"""Authored stand-in for the DexHand021 right-hand floating model.

The reference loads `dexrobot_mujoco/.../dexhand021_right_simplified_floating.xml`
(reference dexhand_env/tasks/dexhand_base.py:171) from a git submodule that is absent offline
(SURVEY.md §8c), so link lengths, inertias and collision geometry below are THIS BUILD'S numbers.
What IS pinned by the reference and reproduced here:
  * DOF order and names: 6 base + FINGER_JOINT_NAMES (constants.py:14-37);
  * body names right_hand_base, r_f_link{f}_{4,tip,pad} (hand_initializer.py:120-134,499-501,
    cfg/task/BaseTask.yaml:84-89);
  * PD gains: base kp 10000 / kd 20, fingers kp 20 / kd 1 (docs/reference-physics-implementation.md:25-26);
  * the built-in 90 deg Y rotation of right_hand_base (observation_encoder.py:1496-1503);
  * indicative joint ranges (docs/reference-dof-control-api.md:143-156, examples/dexhand_test.py:192-236,
    cfg/task/BlindGrasping.yaml:94-100).
Every other number is marked GUESS.
"""
import math
import numpy as np

from . import _abi

BASE_JOINT_NAMES = ["ARTx", "ARTy", "ARTz", "ARRx", "ARRy", "ARRz"]
FINGER_JOINT_NAMES = [f"r_f_joint{f}_{j}" for f in range(1, 6) for j in range(1, 5)]
DOF_NAMES = BASE_JOINT_NAMES + FINGER_JOINT_NAMES
FINGERTIP_BODY_NAMES = [f"r_f_link{f}_tip" for f in range(1, 6)]
FINGERPAD_BODY_NAMES = [f"r_f_link{f}_pad" for f in range(1, 6)]
HAND_BASE_BODY_NAME = "right_hand_base"

BODY_NAMES = (["hand_mount", "ARTx_link", "ARTy_link", "ARTz_link", "ARRx_link", "ARRy_link",
               HAND_BASE_BODY_NAME]
              + [n for f in range(1, 6) for n in (f"r_f_link{f}_1", f"r_f_link{f}_2", f"r_f_link{f}_3",
                                                  f"r_f_link{f}_4", f"r_f_link{f}_pad", f"r_f_link{f}_tip")])
assert len(BODY_NAMES) == _abi.NUM_HAND_BODIES


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


def quat_xyzw_to_mat(q):
    x, y, z, w = [float(v) for v in q]
    n = math.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def mat_to_quat_xyzw(R):
    """Shepperd's method, canonicalised to w >= 0 (host-side only: both the oracle and the HIP kernels
    receive the resulting constants, so no conversion branch ever runs per env)."""
    R = np.asarray(R, dtype=np.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    q = np.array(q)
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _sym6(I):
    return [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]


def _rod_inertia(m, L, r):
    """solid cylinder along local x, about its COM"""
    ixx = 0.5 * m * r * r
    iyy = m * (3 * r * r + L * L) / 12.0
    return np.diag([ixx, iyy, iyy])


class HandModel:
    """Plain-numpy description; `to_struct()` flattens it into the C-ABI DexHandModel."""

    # built-in mount of right_hand_base: 90 deg about Y -> quaternion [0, sqrt(.5), 0, sqrt(.5)]
    R_MOUNT = rot_y(math.pi / 2)

    # GUESS geometry (metres), hand frame H: x along the fingers, y across the palm, z = palm normal
    FINGER_BASE = [(0.025, 0.030, 0.005),      # thumb
                   (0.095, 0.033, 0.0), (0.095, 0.011, 0.0), (0.095, -0.011, 0.0), (0.095, -0.033, 0.0)]
    MCP_OFFSET = 0.008
    L_PROX = [0.040, 0.040, 0.045, 0.040, 0.033]
    L_MID = [0.030, 0.025, 0.028, 0.025, 0.020]
    L_DIST = [0.025, 0.022, 0.024, 0.022, 0.020]
    R_PROX = [0.0090, 0.0080, 0.0080, 0.0080, 0.0075]
    R_MID = [0.0085, 0.0075, 0.0075, 0.0075, 0.0070]
    R_DIST = [0.0080, 0.0070, 0.0070, 0.0070, 0.0065]
    # GUESS masses (kg)
    M_PALM = 0.35
    M_LINK = [0.004, 0.014, 0.009, 0.007]
    PALM_BOX = (0.10, 0.08, 0.025)
    PALM_CAP_Y = (-0.026, 0.0, 0.026)
    PALM_CAP_R = 0.0125

    def __init__(self, initial_hand_pos=(0.0, 0.0, 0.5), initial_hand_rot=(0.0, 0.0, 0.0, 1.0)):
        NJ = _abi.NJ
        self.spawn_pos = np.array(initial_hand_pos, dtype=np.float64)
        self.spawn_rot = quat_xyzw_to_mat(initial_hand_rot)
        self.jtype = np.zeros(NJ, dtype=np.int32)
        self.jRoff = np.tile(np.eye(3), (NJ, 1, 1))
        self.jpoff = np.zeros((NJ, 3))
        self.jaxis = np.zeros((NJ, 3))
        self.mass = np.zeros(NJ)
        self.com = np.zeros((NJ, 3))
        self.inertia = np.zeros((NJ, 6))
        self.kp = np.zeros(NJ)
        self.kd = np.zeros(NJ)
        self.armature = np.zeros(NJ)
        self.lo = np.zeros(NJ)
        self.hi = np.zeros(NJ)
        Rm = self.R_MOUNT

        # ---- base chain: 3 prismatic (world x,y,z at spawn) + 3 revolute (x, y, z), offsets from spawn
        # (docs/DESIGN_DECISIONS.md:21-37: base DOFs are offsets from the spawn pose)
        for j in range(6):
            self.jtype[j] = 0 if j < 3 else 1
            self.jaxis[j, j % 3] = 1.0
            self.kp[j], self.kd[j] = 10000.0, 20.0
            self.lo[j], self.hi[j] = (-1.0, 1.0) if j < 3 else (-math.pi, math.pi)   # GUESS ranges
        # palm body rides on ARRz; its inertia is given in the hand frame and rotated into the joint frame
        bx, by, bz = self.PALM_BOX
        I_H = np.diag([self.M_PALM / 12 * (by * by + bz * bz), self.M_PALM / 12 * (bx * bx + bz * bz),
                       self.M_PALM / 12 * (bx * bx + by * by)])
        self.mass[5] = self.M_PALM
        self.com[5] = Rm @ np.array([0.05, 0.0, 0.0])
        self.inertia[5] = _sym6(Rm @ I_H @ Rm.T)

        # ---- fingers
        limits = {  # (lo, hi) per joint-in-finger; thumb differs
            "thumb": [(0.0, 1.7), (0.0, 1.0), (0.0, 1.2), (0.0, 1.2)],
            "finger": [(0.0, 0.3), (0.0, 1.4), (0.0, 1.3), (0.0, 1.3)],
        }
        for f in range(5):
            j0 = 6 + 4 * f
            Lp, Lm, Ld = self.L_PROX[f], self.L_MID[f], self.L_DIST[f]
            rp, rm_, rd = self.R_PROX[f], self.R_MID[f], self.R_DIST[f]
            if f == 0:
                Rf = rot_z(math.radians(60.0))          # thumb splayed 60 deg towards +y
                ax1 = Rf.T @ np.array([1.0, 0.0, 0.0])  # rotation about the hand's long axis
                lim = limits["thumb"]
            else:
                Rf = np.eye(3)
                ax1 = np.array([0.0, 0.0, 1.0 if f <= 2 else -1.0])  # spread away from the middle finger
                lim = [list(x) for x in limits["finger"]]
                if f == 4:
                    lim[0] = (0.0, 0.6)                 # pinky spreads 2x (constants.py:78-82)
            # joint _1: at the finger base, welded to the palm through the mount rotation
            self.jRoff[j0] = Rm @ Rf
            self.jpoff[j0] = Rm @ np.array(self.FINGER_BASE[f])
            self.jaxis[j0] = ax1
            # joints _2.._4: flexion about -y (positive angle curls towards the palm side, +z)
            offs = [self.MCP_OFFSET, Lp, Lm]
            for l in range(1, 4):
                self.jpoff[j0 + l] = [offs[l - 1], 0.0, 0.0]
                self.jaxis[j0 + l] = [0.0, -1.0, 0.0]
            lens = [self.MCP_OFFSET, Lp, Lm, Ld]
            rads = [rp, rp, rm_, rd]
            for l in range(4):
                j = j0 + l
                self.jtype[j] = 1
                self.kp[j], self.kd[j] = 20.0, 1.0
                self.lo[j], self.hi[j] = lim[l]
                m = self.M_LINK[l] * (1.2 if f == 0 else 1.0)
                self.mass[j] = m
                self.com[j] = [0.5 * lens[l], 0.0, 0.0]
                self.inertia[j] = _sym6(_rod_inertia(m, lens[l], rads[l]))

        # ---- sites: hand base, 5 tips, 5 pads
        self.site_parent = np.zeros(_abi.NSITE, dtype=np.int32)
        self.site_R = np.tile(np.eye(3), (_abi.NSITE, 1, 1))
        self.site_p = np.zeros((_abi.NSITE, 3))
        self.site_parent[0] = 5
        self.site_R[0] = Rm
        for f in range(5):
            jd = 9 + 4 * f
            self.site_parent[1 + f] = jd
            self.site_p[1 + f] = [self.L_DIST[f], 0.0, 0.0]
            self.site_parent[6 + f] = jd
            self.site_p[6 + f] = [0.55 * self.L_DIST[f], 0.0, self.R_DIST[f]]

        # ---- collision capsules
        self.cap_parent = np.zeros(_abi.NCAP, dtype=np.int32)
        self.cap_p0 = np.zeros((_abi.NCAP, 3))
        self.cap_p1 = np.zeros((_abi.NCAP, 3))
        self.cap_r = np.zeros(_abi.NCAP)
        self.cap_fslot = np.zeros(_abi.NCAP, dtype=np.int32)
        for i, y in enumerate(self.PALM_CAP_Y):
            self.cap_parent[i] = 5
            self.cap_p0[i] = Rm @ np.array([0.015, y, 0.0])
            self.cap_p1[i] = Rm @ np.array([0.085, y, 0.0])
            self.cap_r[i] = self.PALM_CAP_R
            self.cap_fslot[i] = _abi.FSLOT_PALM
        for f in range(5):
            lens = [self.L_PROX[f], self.L_MID[f], self.L_DIST[f]]
            rads = [self.R_PROX[f], self.R_MID[f], self.R_DIST[f]]
            for l in range(3):
                c = 3 + 3 * f + l
                self.cap_parent[c] = 6 + 4 * f + 1 + l
                self.cap_p1[c] = [lens[l], 0.0, 0.0]
                self.cap_r[c] = rads[l]
                self.cap_fslot[c] = 3 * f + l
        self.hand_friction = 1.0   # GUESS (MuJoCo default geom friction)

        # ---- published rigid bodies
        nb = _abi.NUM_HAND_BODIES
        self.body_parent = np.full(nb, -1, dtype=np.int32)
        self.body_R = np.tile(np.eye(3), (nb, 1, 1))
        self.body_p = np.zeros((nb, 3))
        self.body_fslot = np.full(nb, -1, dtype=np.int32)
        for b in range(1, 6):
            self.body_parent[b] = b - 1
        self.body_parent[6] = 5
        self.body_R[6] = Rm
        self.body_fslot[6] = _abi.FSLOT_PALM
        for f in range(5):
            b0 = 7 + 6 * f
            for l in range(4):
                self.body_parent[b0 + l] = 6 + 4 * f + l
                if l >= 1:
                    self.body_fslot[b0 + l] = 3 * f + (l - 1)
            self.body_parent[b0 + 4] = 9 + 4 * f
            self.body_p[b0 + 4] = self.site_p[6 + f]
            self.body_parent[b0 + 5] = 9 + 4 * f
            self.body_p[b0 + 5] = self.site_p[1 + f]

        self.body_names = list(BODY_NAMES)
        self.dof_names = list(DOF_NAMES)

    # -- indices the reference resolves through Isaac Gym name lookups (hand_initializer.py:439-588)
    @property
    def hand_local_rigid_body_index(self):
        return self.body_names.index(HAND_BASE_BODY_NAME)

    @property
    def fingertip_local_indices(self):
        return [self.body_names.index(n) for n in FINGERTIP_BODY_NAMES]

    @property
    def fingerpad_local_indices(self):
        return [self.body_names.index(n) for n in FINGERPAD_BODY_NAMES]

    def body_indices(self, names):
        return [self.body_names.index(n) for n in names]

    def dof_props(self):
        """(26, 6) [stiffness, damping, friction, armature, lower, upper] (tensor_manager.py:236,547-554)"""
        out = np.zeros((_abi.NJ, 6), dtype=np.float32)
        out[:, 0], out[:, 1], out[:, 3], out[:, 4], out[:, 5] = self.kp, self.kd, self.armature, self.lo, self.hi
        return out

    def to_struct(self):
        m = _abi.DexHandModel()

        def put(dst, src):
            flat = np.asarray(src, dtype=np.float64).reshape(-1)
            arr = np.ctypeslib.as_array(dst).reshape(-1)
            arr[:] = flat.astype(arr.dtype)

        put(m.spawn_pos, self.spawn_pos)
        put(m.spawn_quat, mat_to_quat_xyzw(self.spawn_rot))
        put(m.jtype, self.jtype)
        put(m.jqoff, [mat_to_quat_xyzw(R) for R in self.jRoff])
        put(m.jpoff, self.jpoff)
        put(m.jaxis, self.jaxis)
        put(m.mass, self.mass)
        put(m.com, self.com)
        put(m.inertia, self.inertia)
        for name in ("kp", "kd", "armature", "lo", "hi"):
            put(getattr(m, name), getattr(self, name))
        put(m.site_parent, self.site_parent)
        put(m.site_q, [mat_to_quat_xyzw(R) for R in self.site_R])
        put(m.site_p, self.site_p)
        put(m.cap_parent, self.cap_parent)
        put(m.cap_p0, self.cap_p0)
        put(m.cap_p1, self.cap_p1)
        put(m.cap_r, self.cap_r)
        put(m.cap_fslot, self.cap_fslot)
        m.hand_friction = float(self.hand_friction)
        put(m.body_parent, self.body_parent)
        put(m.body_q, [mat_to_quat_xyzw(R) for R in self.body_R])
        put(m.body_p, self.body_p)
        put(m.body_fslot, self.body_fslot)
        return m
