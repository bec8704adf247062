"""CPU check of the polynomial sin/cos the HIP kernels use for joint half-angles (dexsim_device.h: sincos_joint): the
function body is re-evaluated here in float32 with the constants parsed from the header, against float64 libm."""
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dexrobot_isaac_amd", "csrc", "dexsim_device.h")


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_sincos_joint_accuracy():
    src = open(HDR).read()
    body = src[src.index("void sincos_joint("):]
    body = body[:body.index("\n}\n")]
    nums = [np.float32(x) for x in re.findall(r"(-?\d+\.\d+(?:e-?\d+)?)f", body)]
    two_over_pi, pio2_hi, pio2_lo, s3, s2, s1, c3, c2, c1, half, one = nums[:11]
    assert abs(float(two_over_pi) - 2 / np.pi) < 1e-7 and abs(-float(pio2_hi) - float(pio2_lo) - np.pi / 2) < 1e-9
    f = np.float32
    h = np.linspace(-4.0, 4.0, 400001).astype(np.float32)          # half angles of |q| <= 8 rad
    kf = np.rint(h * two_over_pi).astype(np.float32)
    r = _fma(kf, np.full_like(h, pio2_hi), h)
    r = _fma(kf, np.full_like(h, pio2_lo), r)
    z = (r * r).astype(np.float32)
    ps = _fma(_fma(_fma(np.full_like(h, s3), z, np.full_like(h, s2)), z, np.full_like(h, s1)), (z * r).astype(np.float32), r)
    pc = _fma(_fma(_fma(np.full_like(h, c3), z, np.full_like(h, c2)), z, np.full_like(h, c1)), (z * z).astype(np.float32),
              _fma(np.full_like(h, half), z, np.full_like(h, one)))
    k = kf.astype(np.int64)
    swap = (k & 1) != 0
    s0, c0 = np.where(swap, pc, ps), np.where(swap, ps, pc)
    sn = np.where((k & 2) != 0, -s0, s0)
    cs = np.where(((k + 1) & 2) != 0, -c0, c0)
    assert np.abs(sn - np.sin(h.astype(np.float64))).max() < 2e-7
    assert np.abs(cs - np.cos(h.astype(np.float64))).max() < 2e-7
