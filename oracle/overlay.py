"""ctypes binding of oracle/_build/liboverlay_oracle.so (CPU restatement of the reference's mesh overlay,
include/RenderSMPLMesh.h:16-110 — TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of the benchmarks may import this module.
PARITY UNPINNED: see the header of overlay_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboverlay_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "overlay_oracle.c")
        if not os.path.exists(_LIB) or (os.path.exists(src) and os.path.getmtime(_LIB) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = C.CDLL(_LIB)
        _lib.overlay_oracle_drawlist.restype = C.c_int
        _lib.overlay_oracle_render.restype = C.c_int
        _lib.overlay_oracle_filter_table.restype = C.POINTER(C.c_uint8)
        _lib.overlay_oracle_slope_table.restype = C.POINTER(C.c_uint8)
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def drawlist(cloud, faces, fx, fy, cx, cy, backface_cull=True):
    """Draw order of one frame: (face, depth, pts[n,6], gray), far to near (RenderSMPLMesh.h:36-92)."""
    cloud = np.ascontiguousarray(cloud, dtype=np.float64).reshape(-1, 3)
    faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
    nf = faces.shape[0]
    face = np.zeros(max(nf, 1), np.int32)
    depth = np.zeros(max(nf, 1), np.float64)
    pts = np.zeros((max(nf, 1), 6), np.int32)
    gray = np.zeros(max(nf, 1), np.int32)
    n = lib().overlay_oracle_drawlist(_p(cloud, C.c_double), C.c_int(cloud.shape[0]), _p(faces, C.c_int32), C.c_int(nf),
                                      C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy),
                                      C.c_int(int(backface_cull)), _p(face, C.c_int32), _p(depth, C.c_double),
                                      _p(pts, C.c_int32), _p(gray, C.c_int32))
    return face[:n], depth[:n], pts[:n], gray[:n]


def fill_triangle(img, pts, gray):
    """cv::fillConvexPoly(img, pts, 3, Scalar(g, g, g), LINE_AA) in place on an HxWx3 uint8 image."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3 and img.strides[2] == 1 and img.strides[1] == 3
    p = np.ascontiguousarray(pts, dtype=np.int32).reshape(6)
    lib().overlay_oracle_fill_triangle(_p(img, C.c_uint8), C.c_int(img.shape[1]), C.c_int(img.shape[0]),
                                       C.c_size_t(img.strides[0]), _p(p, C.c_int32), C.c_int(int(gray)))
    return img


def render(cloud, faces, img, fx, fy, cx, cy, fill=True, backface_cull=True, wireframe=False):
    """renderSMPLMesh(cloud, faces, img, fx, fy, cx, cy, fill, backface_cull, wireframe), in place."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3 and img.strides[2] == 1 and img.strides[1] == 3
    cloud = np.ascontiguousarray(cloud, dtype=np.float64).reshape(-1, 3)
    faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
    lib().overlay_oracle_render(_p(cloud, C.c_double), C.c_int(cloud.shape[0]), _p(faces, C.c_int32),
                                C.c_int(faces.shape[0]), _p(img, C.c_uint8), C.c_int(img.shape[1]),
                                C.c_int(img.shape[0]), C.c_size_t(img.strides[0]), C.c_double(fx), C.c_double(fy),
                                C.c_double(cx), C.c_double(cy), C.c_int(int(fill)), C.c_int(int(backface_cull)),
                                C.c_int(int(wireframe)))
    return img


def tables():
    f = np.ctypeslib.as_array(lib().overlay_oracle_filter_table(), shape=(64,)).copy()
    s = np.ctypeslib.as_array(lib().overlay_oracle_slope_table(), shape=(32,)).copy()
    return f, s
