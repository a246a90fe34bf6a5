"""Delaunay edges from the REFERENCE's vendored Triangle (oracle/_ref/libref_triangle.so, built by
oracle/Makefile from /root/reference/src/flame/external/triangle/triangle.cpp in place).
Fixture generation only; exists only where /root/reference does (or where the prebuilt .so travelled)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_triangle.so")


def available() -> bool:
    return os.path.exists(_PATH)


def delaunay_edges(pos: np.ndarray) -> np.ndarray:
    L = C.CDLL(_PATH)
    IP = C.POINTER(C.c_int)
    L.ref_delaunay.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(IP), C.POINTER(C.c_int), C.POINTER(IP)]
    L.ref_delaunay.restype = C.c_int
    L.ref_delaunay_free.argtypes = [IP]
    pts = np.ascontiguousarray(pos, dtype=np.float32)
    e = IP()
    E = L.ref_delaunay(pts.ctypes.data_as(C.POINTER(C.c_float)), pts.shape[0], C.byref(e), None, None)
    if E < 0:
        raise RuntimeError("ref_delaunay failed")
    out = np.ctypeslib.as_array(e, shape=(2 * E,)).astype(np.int32).reshape(E, 2).copy()
    L.ref_delaunay_free(e)
    return out


def delaunay_triangles(pos: np.ndarray) -> np.ndarray:
    """Triangles (T,3) of the reference's Triangle run ("zneQB"), in its order and winding."""
    L = C.CDLL(_PATH)
    IP = C.POINTER(C.c_int)
    L.ref_delaunay.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(IP), C.POINTER(C.c_int), C.POINTER(IP)]
    L.ref_delaunay.restype = C.c_int
    L.ref_delaunay_free.argtypes = [IP]
    pts = np.ascontiguousarray(pos, dtype=np.float32)
    e, t, nt = IP(), IP(), C.c_int(0)
    E = L.ref_delaunay(pts.ctypes.data_as(C.POINTER(C.c_float)), pts.shape[0], C.byref(e), C.byref(nt), C.byref(t))
    if E < 0:
        raise RuntimeError("ref_delaunay failed")
    out = np.ctypeslib.as_array(t, shape=(3 * nt.value,)).astype(np.int32).reshape(-1, 3).copy()
    L.ref_delaunay_free(e)
    L.ref_delaunay_free(t)
    return out
