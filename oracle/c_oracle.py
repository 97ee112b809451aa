"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/vxm_oracle.c (`make -C oracle`)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_grad_loss.restype = ctypes.c_double
        _lib.orc_ncc_loss.restype = ctypes.c_double
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def warp3d(src, flow, mode="bilinear"):
    src, ps = _f(src)
    flow, pf = _f(flow)
    B, C, D, H, W = src.shape
    out = np.empty((B, C, D, H, W), np.float32)
    lib().orc_warp3d(ps, pf, out.ctypes.data_as(ctypes.c_void_p), B, C, D, H, W, 1 if mode == "nearest" else 0)
    return out


def vecint3d(vec, nsteps):
    vec, pv = _f(vec)
    B, _, D, H, W = vec.shape
    out = np.empty_like(vec)
    tmp = np.empty_like(vec)
    lib().orc_vecint3d(pv, out.ctypes.data_as(ctypes.c_void_p), tmp.ctypes.data_as(ctypes.c_void_p), B, D, H, W, nsteps)
    return out


def resize3d(x, vel_resize):
    import math
    x, px = _f(x)
    B, C, D, H, W = x.shape
    factor = 1.0 / vel_resize
    if factor == 1:
        return x.copy()
    oD, oH, oW = (int(math.floor(s * factor)) for s in (D, H, W))
    out = np.empty((B, C, oD, oH, oW), np.float32)
    lib().orc_resize3d(px, out.ctypes.data_as(ctypes.c_void_p), B, C, D, H, W, oD, oH, oW, ctypes.c_float(factor))
    return out


def grad_loss(y, penalty="l1", loss_mult=None):
    y, py = _f(y)
    B, C, D, H, W = y.shape
    return lib().orc_grad_loss(py, B, C, D, H, W, 1 if penalty == "l2" else 0,
                               ctypes.c_double(1.0 if loss_mult is None else loss_mult))


def ncc_loss(I, J, win=9):
    I, pi = _f(I)
    J, pj = _f(J)
    B, C, D, H, W = I.shape
    assert C == 1
    return lib().orc_ncc_loss(pi, pj, B, D, H, W, win)


def conv3d_k3(x, w, b, slope=0.2):
    x, px = _f(x)
    w, pw = _f(w)
    b, pb = _f(b)
    B, Cin, D, H, W = x.shape
    Cout = w.shape[0]
    y = np.empty((B, Cout, D, H, W), np.float32)
    lib().orc_conv3d_k3(px, pw, pb, y.ctypes.data_as(ctypes.c_void_p), B, Cin, Cout, D, H, W, ctypes.c_float(slope))
    return y
