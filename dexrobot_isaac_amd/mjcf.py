"""MJCF <-> HandModel.

The reference loads `dexrobot_mujoco/.../dexhand021_right_simplified_floating.xml` through Isaac Gym
(reference hand_initializer.py:209-257); that file is an absent submodule, so the engine ships an authored model
(hand_model.py).  This module is the loader that replaces the authored numbers the moment the real file is available:

    model = load_mjcf("/path/to/dexhand021_right_simplified_floating.xml")
    env = make_env("BlindGrasping", 4096, "cuda:0", "cuda:0", 0, hand_model=model)

What it reads, following how Isaac Gym imports MJCF per the reference's notes
(docs/reference-physics-implementation.md:9-47): body tree (pos / quat, MuJoCo quats are wxyz), hinge / slide joints
(axis, pos, range, damping -> kd), `<position kp=...>` actuators -> kp, `<inertial>` (mass, pos, diaginertia or
fullinertia, quat), capsule / sphere / box geoms (-> capsules), and the bodies named `right_hand_base`,
`r_f_link{f}_{tip,pad}` as sites.  The kernels' topology is fixed (6 base joints, 5 fingers x 4 hinges), so a file with
a different tree is rejected with a clear message.  `export_mjcf` writes the authored model in the same dialect; the
round trip is the loader's test (tests/test_mjcf.py) because no reference MJCF exists to test against.
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

from . import _abi
from .hand_model import (BASE_JOINT_NAMES, FINGER_JOINT_NAMES, HAND_BASE_BODY_NAME, HandModel, mat_to_quat_xyzw,
                         quat_xyzw_to_mat)


def _f(text, n=None, default=None):
    if text is None:
        return default
    v = np.array([float(x) for x in text.split()], dtype=np.float64)
    if n is not None and v.size != n:
        raise ValueError(f"expected {n} numbers, got '{text}'")
    return v


def _wxyz_to_mat(q):
    w, x, y, z = q
    return quat_xyzw_to_mat([x, y, z, w])


def _mat_to_wxyz(R):
    x, y, z, w = mat_to_quat_xyzw(R)
    return np.array([w, x, y, z])


def _fmt(v):
    return " ".join(f"{float(x):.9g}" for x in np.asarray(v).reshape(-1))


# --------------------------------------------------------------------------------------------------- export
def export_mjcf(model: HandModel) -> str:
    """Serialise a HandModel as MJCF: one body per joint (joint anchored at the body origin), capsule geoms by
    `fromto`, tip / pad as child bodies, `<position>` actuators carrying kp and joint `damping` carrying kd."""
    root = ET.Element("mujoco", model="dexhand021_right_like_floating")
    ET.SubElement(root, "compiler", angle="radian")
    world = ET.SubElement(root, "worldbody")
    mount = ET.SubElement(world, "body", name="hand_mount", pos=_fmt(model.spawn_pos), quat=_fmt(_mat_to_wxyz(model.spawn_rot)))
    elems = {}
    parent_of = lambda j: -1 if j == 0 else (j - 1 if j < 6 else (5 if (j - 6) % 4 == 0 else j - 1))
    names = ["ARTx_link", "ARTy_link", "ARTz_link", "ARRx_link", "ARRy_link", "ARRz_link"] + \
            [f"r_f_link{f}_{l}" for f in range(1, 6) for l in range(1, 5)]
    for j in range(_abi.NJ):
        p = parent_of(j)
        par = mount if p < 0 else elems[p]
        b = ET.SubElement(par, "body", name=names[j], pos=_fmt(model.jpoff[j]), quat=_fmt(_mat_to_wxyz(model.jRoff[j])))
        elems[j] = b
        if model.mass[j] > 0:
            I = model.inertia[j]
            ET.SubElement(b, "inertial", pos=_fmt(model.com[j]), mass=_fmt([model.mass[j]]),
                          fullinertia=_fmt([I[0], I[1], I[2], I[3], I[4], I[5]]))
        ET.SubElement(b, "joint", name=model.dof_names[j], type="slide" if model.jtype[j] == 0 else "hinge",
                      axis=_fmt(model.jaxis[j]), pos="0 0 0", range=_fmt([model.lo[j], model.hi[j]]), limited="true",
                      damping=_fmt([model.kd[j]]), armature=_fmt([model.armature[j]]))
    hb = ET.SubElement(elems[5], "body", name=HAND_BASE_BODY_NAME, pos=_fmt(model.site_p[0]), quat=_fmt(_mat_to_wxyz(model.site_R[0])))
    Rm_inv = model.site_R[0].T
    for c in range(_abi.NCAP):
        j = int(model.cap_parent[c])
        if j == 5:   # palm capsules live on right_hand_base, expressed in its (mounted) frame
            ET.SubElement(hb, "geom", type="capsule", size=_fmt([model.cap_r[c]]),
                          fromto=_fmt(np.concatenate([Rm_inv @ (model.cap_p0[c] - model.site_p[0]), Rm_inv @ (model.cap_p1[c] - model.site_p[0])])),
                          friction=_fmt([model.hand_friction, 0.005, 0.0001]))
        else:
            ET.SubElement(elems[j], "geom", type="capsule", size=_fmt([model.cap_r[c]]),
                          fromto=_fmt(np.concatenate([model.cap_p0[c], model.cap_p1[c]])),
                          friction=_fmt([model.hand_friction, 0.005, 0.0001]))
    for f in range(5):
        jd = 9 + 4 * f
        ET.SubElement(elems[jd], "body", name=f"r_f_link{f + 1}_pad", pos=_fmt(model.site_p[6 + f]), quat=_fmt(_mat_to_wxyz(model.site_R[6 + f])))
        ET.SubElement(elems[jd], "body", name=f"r_f_link{f + 1}_tip", pos=_fmt(model.site_p[1 + f]), quat=_fmt(_mat_to_wxyz(model.site_R[1 + f])))
    act = ET.SubElement(root, "actuator")
    for j in range(_abi.NJ):
        ET.SubElement(act, "position", name=f"act_{model.dof_names[j]}", joint=model.dof_names[j], kp=_fmt([model.kp[j]]))
    ET.indent(root)
    return ET.tostring(root, encoding="unicode")


# --------------------------------------------------------------------------------------------------- load
class _Joint:
    def __init__(self, el, body_frame_R, body_frame_p, parent_joint):
        self.name = el.get("name")
        self.type = {"hinge": 1, "slide": 0}.get(el.get("type", "hinge"))
        if self.type is None:
            raise ValueError(f"joint '{self.name}': only hinge and slide joints are supported")
        self.axis = _f(el.get("axis"), 3, np.array([0.0, 0.0, 1.0]))
        self.axis = self.axis / np.linalg.norm(self.axis)
        self.pos = _f(el.get("pos"), 3, np.zeros(3))
        self.range = _f(el.get("range"), 2)
        self.damping = float(el.get("damping", 0.0))
        self.armature = float(el.get("armature", 0.0))
        self.parent_joint = parent_joint


def load_mjcf(source, default_joint_damping=None) -> HandModel:
    """Parse an MJCF file (path) or string into a HandModel.  Raises ValueError when the tree does not have the
    6 + 5x4 DexHand topology the kernels are compiled for."""
    text = open(source).read() if not source.lstrip().startswith("<") else source
    root = ET.fromstring(text)
    comp = root.find("compiler")
    if comp is not None and comp.get("angle", "degree") != "radian":
        raise ValueError("only <compiler angle='radian'> files are supported")
    dflt = root.find("default/joint")
    dflt_damp = float(dflt.get("damping")) if dflt is not None and dflt.get("damping") else (default_joint_damping or 0.0)
    world = root.find("worldbody")
    m = HandModel.__new__(HandModel)
    NJ = _abi.NJ
    for name, shape in (("jRoff", (NJ, 3, 3)), ("jpoff", (NJ, 3)), ("jaxis", (NJ, 3)), ("com", (NJ, 3)), ("inertia", (NJ, 6))):
        setattr(m, name, np.zeros(shape))
    for name in ("mass", "kp", "kd", "armature", "lo", "hi"):
        setattr(m, name, np.zeros(NJ))
    m.jtype = np.zeros(NJ, dtype=np.int32)
    m.site_parent = np.zeros(_abi.NSITE, dtype=np.int32)
    m.site_R = np.tile(np.eye(3), (_abi.NSITE, 1, 1))
    m.site_p = np.zeros((_abi.NSITE, 3))
    caps = []                                   # (joint index, p0, p1, r)
    order = BASE_JOINT_NAMES + FINGER_JOINT_NAMES
    jidx = {n: i for i, n in enumerate(order)}
    seen = {}
    friction = []

    def inertial(el, j, R_bj, p_bj):
        if el is None:
            return
        mass = float(el.get("mass"))
        pos = _f(el.get("pos"), 3, np.zeros(3))
        Rq = _wxyz_to_mat(_f(el.get("quat"), 4, np.array([1.0, 0, 0, 0])))
        if el.get("fullinertia"):
            a = _f(el.get("fullinertia"), 6)
            I = np.array([[a[0], a[3], a[4]], [a[3], a[1], a[5]], [a[4], a[5], a[2]]])
        else:
            I = np.diag(_f(el.get("diaginertia"), 3))
        I = Rq @ I @ Rq.T
        # accumulate into joint j (several MJCF bodies may be welded to one joint frame)
        c_new, I_new = R_bj @ pos + p_bj, R_bj @ I @ R_bj.T
        m0, c0 = m.mass[j], m.com[j]
        I0 = np.array([[m.inertia[j][0], m.inertia[j][3], m.inertia[j][4]], [m.inertia[j][3], m.inertia[j][1], m.inertia[j][5]],
                       [m.inertia[j][4], m.inertia[j][5], m.inertia[j][2]]])
        mt = m0 + mass
        c = (m0 * c0 + mass * c_new) / mt
        pa = lambda mm, d: mm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        It = I0 + pa(m0, c0 - c) + I_new + pa(mass, c_new - c)
        m.mass[j], m.com[j] = mt, c
        m.inertia[j] = [It[0, 0], It[1, 1], It[2, 2], It[0, 1], It[0, 2], It[1, 2]]

    def geoms(body, j, R_bj, p_bj):
        for g in body.findall("geom"):
            t = g.get("type", "sphere")
            if g.get("contype") == "0" and g.get("conaffinity") == "0":
                continue
            size = _f(g.get("size"))
            if g.get("friction"):
                friction.append(_f(g.get("friction"))[0])
            if t == "capsule" and g.get("fromto"):
                ft = _f(g.get("fromto"), 6)
                p0, p1, r = ft[:3], ft[3:], size[0]
            else:
                gp = _f(g.get("pos"), 3, np.zeros(3))
                gR = _wxyz_to_mat(_f(g.get("quat"), 4, np.array([1.0, 0, 0, 0])))
                if t == "sphere":
                    p0 = p1 = gp
                    r = size[0]
                elif t == "capsule":
                    p0, p1, r = gp - gR @ [0, 0, size[1]], gp + gR @ [0, 0, size[1]], size[0]
                elif t == "box":        # capsule along the longest half-extent, radius = mean of the other two
                    k = int(np.argmax(size))
                    r = float(np.mean(np.delete(size, k)))
                    ax = np.zeros(3)
                    ax[k] = max(size[k] - r, 0.0)
                    p0, p1 = gp - gR @ ax, gp + gR @ ax
                else:
                    continue            # meshes etc.: no primitive fit available
            caps.append((j, R_bj @ p0 + p_bj, R_bj @ p1 + p_bj, float(r)))

    def walk(body, parent_joint, R_pj, p_pj):
        """R_pj, p_pj: pose of this body's PARENT BODY frame in the parent joint's frame."""
        R_b = R_pj @ _wxyz_to_mat(_f(body.get("quat"), 4, np.array([1.0, 0, 0, 0])))
        p_b = p_pj + R_pj @ _f(body.get("pos"), 3, np.zeros(3))
        name = body.get("name", "")
        joints = body.findall("joint")
        cur, R_cur, p_cur = parent_joint, R_b, p_b          # body frame in the current joint frame
        for jel in joints:
            jt = _Joint(jel, R_cur, p_cur, cur)
            if jt.name not in jidx:
                raise ValueError(f"unexpected joint '{jt.name}': not one of the 26 DexHand DOFs")
            j = jidx[jt.name]
            expect = -1 if j == 0 else (j - 1 if j < 6 else (5 if (j - 6) % 4 == 0 else j - 1))
            if cur != expect:
                raise ValueError(f"joint '{jt.name}' hangs off joint {cur}, the kernels expect parent {expect}")
            m.jtype[j] = jt.type
            m.jRoff[j] = R_cur
            m.jpoff[j] = p_cur + R_cur @ jt.pos
            m.jaxis[j] = jt.axis
            m.kd[j] = jt.damping if jel.get("damping") is not None else dflt_damp
            m.armature[j] = jt.armature
            if jt.range is None:
                raise ValueError(f"joint '{jt.name}' has no range")
            m.lo[j], m.hi[j] = jt.range
            seen[jt.name] = j
            cur, R_cur, p_cur = j, np.eye(3), -jt.pos          # body frame = joint frame * Trans(-anchor)
        if cur >= 0:
            inertial(body.find("inertial"), cur, R_cur, p_cur)
            geoms(body, cur, R_cur, p_cur)
        if name == HAND_BASE_BODY_NAME:
            m.site_parent[0], m.site_R[0], m.site_p[0] = cur, R_cur, p_cur
        for f in range(5):
            if name == f"r_f_link{f + 1}_tip":
                m.site_parent[1 + f], m.site_R[1 + f], m.site_p[1 + f] = cur, R_cur, p_cur
            if name == f"r_f_link{f + 1}_pad":
                m.site_parent[6 + f], m.site_R[6 + f], m.site_p[6 + f] = cur, R_cur, p_cur
        for child in body.findall("body"):
            walk(child, cur, R_cur, p_cur)

    tops = world.findall("body")
    if len(tops) != 1:
        raise ValueError("expected exactly one top-level body (the floating-base mount)")
    top = tops[0]
    m.spawn_pos = _f(top.get("pos"), 3, np.zeros(3))
    m.spawn_rot = _wxyz_to_mat(_f(top.get("quat"), 4, np.array([1.0, 0, 0, 0])))
    # the mount body itself carries no joint in the exported dialect; a file whose top body holds the base joints works too
    holder = ET.Element("body")
    holder.extend(list(top))
    for jel in top.findall("joint"):
        pass
    walk(holder, -1, np.eye(3), np.zeros(3))
    missing = [n for n in order if n not in seen]
    if missing:
        raise ValueError(f"MJCF lacks DexHand joints: {missing}")
    for a in root.findall("actuator/position"):
        if a.get("joint") in jidx:
            m.kp[jidx[a.get("joint")]] = float(a.get("kp", 1.0))
    # capsules: 3 on the palm + (proximal, middle, distal) per finger, in the order the kernels expect
    m.cap_parent = np.zeros(_abi.NCAP, dtype=np.int32)
    m.cap_p0, m.cap_p1 = np.zeros((_abi.NCAP, 3)), np.zeros((_abi.NCAP, 3))
    m.cap_r, m.cap_fslot = np.zeros(_abi.NCAP), np.zeros(_abi.NCAP, dtype=np.int32)
    palm = [c for c in caps if c[0] == 5]
    if len(palm) < 1:
        raise ValueError("no collision geom on the palm")
    palm = (palm * 3)[:3]
    for i, (j, p0, p1, r) in enumerate(palm):
        m.cap_parent[i], m.cap_p0[i], m.cap_p1[i], m.cap_r[i], m.cap_fslot[i] = 5, p0, p1, r, _abi.FSLOT_PALM
    for f in range(5):
        for l in range(3):
            j = 6 + 4 * f + 1 + l
            mine = [c for c in caps if c[0] == j]
            if not mine:
                raise ValueError(f"no collision geom on {order[j]}'s link")
            _, p0, p1, r = max(mine, key=lambda c: np.linalg.norm(c[2] - c[1]) + c[3])
            c = 3 + 3 * f + l
            m.cap_parent[c], m.cap_p0[c], m.cap_p1[c], m.cap_r[c], m.cap_fslot[c] = j, p0, p1, r, 3 * f + l
    m.hand_friction = float(np.mean(friction)) if friction else 1.0
    # published bodies: same naming / ordering contract as the authored model
    ref = HandModel()
    m.body_parent, m.body_fslot = ref.body_parent.copy(), ref.body_fslot.copy()
    m.body_R, m.body_p = np.tile(np.eye(3), (_abi.NUM_HAND_BODIES, 1, 1)), np.zeros((_abi.NUM_HAND_BODIES, 3))
    m.body_R[6], m.body_p[6] = m.site_R[0], m.site_p[0]
    for f in range(5):
        b0 = 7 + 6 * f
        m.body_R[b0 + 4], m.body_p[b0 + 4] = m.site_R[6 + f], m.site_p[6 + f]
        m.body_R[b0 + 5], m.body_p[b0 + 5] = m.site_R[1 + f], m.site_p[1 + f]
    m.body_names, m.dof_names = list(ref.body_names), list(ref.dof_names)
    return m
