"""ctypes binding of oracle/_build/liboracle.so (CPU restatement — TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: see the header of bodyfit_oracle.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bodyfit_oracle.cpp")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.oracle_model_create.restype = C.c_void_p
        _lib.oracle_gmm_create.restype = C.c_void_p
        _lib.oracle_mean_pixel_error.restype = C.c_double
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _c32i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class OracleModel:
    def __init__(self, model, pose_blend_data: bool = True):
        self.m = model
        self._keep = [_c64(model.v_template), _c64(model.shapedirs),
                      _c64(model.posedirs) if pose_blend_data else None, _c64(model.j_regressor),
                      _c64(model.weights), _c32i(model.parent), _c32i(model.landmark_vid)]
        k = self._keep
        self.V, self.nJ, self.nS = model.n_verts, model.n_joints, model.n_shape
        self.P = model.posedirs.shape[2]
        self.h = C.c_void_p(lib().oracle_model_create(self.V, self.nJ, self.nS, self.P, _d(k[0]), _d(k[1]),
                                                      _d(k[2]), _d(k[3]), _d(k[4]), _i(k[5]),
                                                      len(model.landmark_vid), _i(k[6])))
        if getattr(model, "n_kp_regressors", 0):
            self._keep += [_c32i(model.kpreg_offset), _c32i(model.kpreg_vid), _c64(model.kpreg_weight)]
            lib().oracle_model_set_kp_regressors(self.h, model.n_kp_regressors, _i(self._keep[-3]), _i(self._keep[-2]),
                                                 _d(self._keep[-1]))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_model_destroy(self.h)
            self.h = None

    def derived(self):
        J0 = np.empty((self.nJ, 3)); S = np.empty((3 * self.nJ, self.nS)); off = np.empty((self.nJ, 3))
        lib().oracle_model_derived(self.h, _d(J0), _d(S), _d(off))
        return J0, S, off

    def kp_block(self, kp_id, uv, intr, R0, x, use_shape=True, pose_blend=True, mode=0, want_jac=True):
        x = _c64(x); intr = _c64(intr); R0 = _c64(R0)
        r = np.empty(2); J = np.empty((2, len(x))) if want_jac else None
        lib().oracle_kp_block(self.h, int(kp_id), C.c_double(uv[0]), C.c_double(uv[1]), _d(intr), _d(R0),
                              int(use_shape), int(pose_blend), _d(x), len(x), int(mode), _d(r), _d(J))
        return r, J

    def evaluate_batch(self, seq_or_tuple, params, beta, ncols, use_shape=True, pose_blend=True, mode=0,
                       nthreads=0, want_jac=True):
        """seq: object with kp_offset, kp_id, kp_uv, intr, R0.  beta: [nS] shared or [F,nS]."""
        s = seq_or_tuple
        F = len(s.kp_offset) - 1
        params = _c64(params); beta = _c64(beta)
        stride = 0 if beta.ndim == 1 else self.nS
        Ktot = int(s.kp_offset[F])
        r = np.empty(2 * Ktot); J = np.empty((2 * Ktot, ncols)) if want_jac else None
        ko, ki, ku, it, r0 = _c32i(s.kp_offset), _c32i(s.kp_id), _c64(s.kp_uv), _c64(s.intr), _c64(s.R0)
        lib().oracle_evaluate_batch(self.h, F, _i(ko), _i(ki), _d(ku), _d(it), _d(r0), int(ncols),
                                    int(use_shape), stride, int(pose_blend), _d(params), _d(beta), int(mode),
                                    int(nthreads), _d(r), _d(J))
        return r, J

    def forward(self, x, beta, R0, use_shape=True, pose_blend=True, nthreads=0, want_cloud=True):
        x = _c64(x); beta = _c64(beta); R0 = _c64(R0)
        joints = np.empty((self.nJ, 3)); cloud = np.empty((self.V, 3)) if want_cloud else None
        lib().oracle_forward(self.h, _d(x), _d(beta), _d(R0), int(use_shape), int(pose_blend), int(nthreads),
                             _d(joints), _d(cloud))
        return joints, cloud


    def forward_batch(self, params, beta, R0, use_shape=True, pose_blend=True, nthreads=0, want_cloud=True):
        params = _c64(params); beta = _c64(beta); R0 = _c64(R0)
        F = params.shape[0]
        stride = 0 if beta.ndim == 1 else self.nS
        joints = np.empty((F, self.nJ, 3)); cloud = np.empty((F, self.V, 3)) if want_cloud else None
        lib().oracle_forward_batch(self.h, F, _d(params), _d(beta), stride, _d(R0), int(use_shape), int(pose_blend),
                                   int(nthreads), _d(joints), _d(cloud))
        return joints, cloud


class OracleGmm:
    def __init__(self, weights, means, covs, resid_scale=np.sqrt(0.5)):
        self.K, self.D = means.shape
        w, m, c = _c64(weights), _c64(means), _c64(covs)
        self.h = C.c_void_p(lib().oracle_gmm_create(self.K, self.D, _d(w), _d(m), _d(c), C.c_double(resid_scale)))
        if not self.h:
            raise ValueError("covariance not SPD")

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_gmm_destroy(self.h)
            self.h = None

    def get(self):
        L = np.empty((self.K, self.D, self.D)); nlw = np.empty(self.K)
        lib().oracle_gmm_get(self.h, _d(L), _d(nlw))
        return L, nlw

    def residual(self, x):
        x = _c64(x); r = np.empty(self.D + 1)
        k = lib().oracle_gmm_residual(self.h, _d(x), _d(r))
        return r, k


def pose_prior(gmm, beta_pose, x, want_jac=True):
    x = _c64(x); D = len(x)
    nres = D + 1 if gmm is not None else D
    r = np.empty(nres); J = np.empty((nres, D)) if want_jac else None
    k = lib().oracle_pose_prior(gmm.h if gmm is not None else None, C.c_double(beta_pose), D, _d(x), _d(r), _d(J))
    return r, J, k


def priors_batch(gmm, beta_pose, params, beta_shape=0.0, beta=None, want_jac=True, nthreads=0):
    """Pose prior (+ per-frame shape prior) blocks of F frames, threads over blocks (oracle_priors_batch)."""
    params = _c64(params)
    F, D = params.shape[0], params.shape[1] - 7
    nres = D + 1 if gmm is not None else D
    r = np.zeros((F, nres)); J = np.zeros((F, nres, D)) if want_jac else None
    comp = np.zeros(F, np.int32)
    b = _c64(beta) if beta is not None else None
    nS = b.shape[1] if b is not None else 0
    rs = np.zeros((F, nS)) if b is not None else None
    lib().oracle_priors_batch(gmm.h if gmm is not None else None, C.c_double(beta_pose), D, F, _d(params),
                              C.c_double(beta_shape if b is not None else 0.0), nS, _d(b), int(want_jac), int(nthreads),
                              _d(r), _d(J), _i(comp), _d(rs))
    return r, J, comp, rs


def huber(delta, s):
    rho = np.empty(3)
    lib().oracle_huber(C.c_double(delta), C.c_double(s), _d(rho))
    return rho


def mean_pixel_error(jid, uv, joints, intr):
    jid = _c32i(jid); uv = _c64(uv); joints = _c64(joints); intr = _c64(intr)
    return lib().oracle_mean_pixel_error(len(jid), _i(jid), _d(uv), _d(joints), _d(intr))


def rodrigues(aa):
    aa = _c64(aa); R = np.empty((3, 3)); dR = np.empty((3, 3, 3))
    lib().oracle_rodrigues(_d(aa), _d(R), _d(dR))
    return R, dR


def max_threads():
    return lib().oracle_max_threads()
