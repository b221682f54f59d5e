"""ctypes loader for the plain-C oracle (test infrastructure; see oracle/vsseg_oracle_c.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvsseg_oracle_c.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    if not os.path.exists(_SO):
        build()
    L = ctypes.CDLL(_SO)
    L.vsseg_c_swi_starts.restype = ctypes.c_int
    L.vsseg_c_gaussian_1d.restype = ctypes.c_int
    L.vsseg_c_gaussian_1d.argtypes = [ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
    L.vsseg_c_swi_starts.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double] + [ctypes.c_void_p] * 4 + [ctypes.c_int]
    return L


def swi_starts(size, roi, overlap):
    L = lib()
    padded, padb, iv = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    starts = (ctypes.c_int * 64)()
    n = L.vsseg_c_swi_starts(size, roi, float(overlap), ctypes.byref(padded), ctypes.byref(padb), ctypes.byref(iv), starts, 64)
    assert n > 0
    return padded.value, padb.value, iv.value, [starts[i] for i in range(n)]


def gaussian_1d(sigma):
    L = lib()
    buf = np.zeros(4096, np.float32)
    n = L.vsseg_c_gaussian_1d(float(sigma), buf.ctypes.data, 4096)
    assert n > 0
    return buf[:n].copy()


def _iarr(v):
    return (ctypes.c_int * 3)(*v)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def conv3d(x, w, b, stride, pad):
    L = lib()
    N, Ci, X, Y, Z = x.shape
    Co, _, *k = w.shape
    out = [(d + 2 * p - kk) // s + 1 for d, p, kk, s in zip((X, Y, Z), pad, k, stride)]
    y = np.zeros((N, Co, *out), np.float32)
    x, w, b = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)
    L.vsseg_c_conv3d(_p(x), _p(w), _p(b), _p(y), N, Ci, X, Y, Z, Co, _iarr(k), _iarr(stride), _iarr(pad))
    return y


def conv_transpose3d(x, w, b, stride, pad, opad):
    L = lib()
    N, Ci, X, Y, Z = x.shape
    _, Co, *k = w.shape
    out = [(d - 1) * s - 2 * p + kk + o for d, p, kk, s, o in zip((X, Y, Z), pad, k, stride, opad)]
    y = np.zeros((N, Co, *out), np.float32)
    x, w, b = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)
    L.vsseg_c_conv_transpose3d(_p(x), _p(w), _p(b), _p(y), N, Ci, X, Y, Z, Co, _iarr(k), _iarr(stride), _iarr(pad), _iarr(opad))
    return y
