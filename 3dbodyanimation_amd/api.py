"""ctypes binding of libbodyfit.so (include/bodyfit.h) and a thin host-side mirror of the reference types.

There is no CPU fallback: if the HIP library is missing, or no GPU is visible when a model is
created, the calls raise BodyfitError.  Names follow the reference: PixelKP (include/Sim3BA.h:9),
Sim3Params (:11-19), FramePoseParams (include/MultiFrameBA.h:9-14).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BODYFIT_LIB", os.path.join(_HERE, "libbodyfit.so"))
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "bodyfit.h")
N_FRAME_PARAMS = 76

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)


class BodyfitError(RuntimeError):
    pass


class _ModelDesc(C.Structure):
    _fields_ = [("n_verts", C.c_int), ("n_joints", C.c_int), ("n_shape", C.c_int), ("n_pose_feat", C.c_int),
                ("v_template", _dp), ("shapedirs", _dp), ("posedirs", _dp), ("j_regressor", _dp),
                ("weights", _dp), ("parent", _ip), ("n_landmarks", C.c_int), ("landmark_vid", _ip),
                ("n_kp_regressors", C.c_int), ("kpreg_offset", _ip), ("kpreg_vid", _ip), ("kpreg_weight", _dp)]


class _ProblemDesc(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("kp_offset", _ip), ("kp_id", _ip), ("kp_uv", _dp),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("R0", _dp), ("n_cols", C.c_int), ("use_shape", C.c_int), ("beta_per_frame", C.c_int),
                ("pose_blend", C.c_int), ("beta_pose", C.c_double), ("gmm", C.c_void_p),
                ("beta_shape", C.c_double), ("lambda_temporal", C.c_double), ("temporal_halo", C.c_int),
                ("huber_delta", C.c_double), ("want_mesh", C.c_int)]


class Layout(C.Structure):
    _fields_ = [("n_keypoints", C.c_int), ("n_cols", C.c_int), ("reproj_rows", C.c_int),
                ("prior_rows_per_frame", C.c_int), ("shape_rows", C.c_int), ("temporal_rows", C.c_int),
                ("total_rows", C.c_int)]


class FitOptions(C.Structure):
    _fields_ = [("max_iters", C.c_int), ("scale_lo", C.c_double), ("scale_hi", C.c_double), ("verbose", C.c_int),
                ("solver", C.c_int)]


class FitSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("termination", C.c_int), ("usable", C.c_int), ("n_successful", C.c_int),
                ("n_unsuccessful", C.c_int), ("n_sweeps", C.c_int), ("initial_cost", C.c_double),
                ("final_cost", C.c_double), ("n_sweeps_issued", C.c_int)]


class _OverlayDesc(C.Structure):
    _fields_ = [("device", C.c_int), ("n_vertices", C.c_int), ("n_faces", C.c_int), ("faces", C.POINTER(C.c_int32)),
                ("width", C.c_int), ("height", C.c_int), ("max_frames", C.c_int)]


_ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.c_int, C.c_int)
_ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, _dp, C.c_int)


class Comm(C.Structure):   # bodyfit_comm (include/bodyfit.h)
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("ctx", C.c_void_p), ("allreduce", _ALLREDUCE_CB),
                ("allgather", _ALLGATHER_CB)]


class Rccl:
    """RCCL communicator of a sharded solve (bodyfit_rccl_*): created by the library from a 128-byte id that rank 0 obtains
    and the application ships to the other ranks (any host channel: here whatever the caller uses, e.g. torch.distributed
    broadcast_object_list), or wrapped around an ncclComm_t the application already has."""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_ubyte * 128)()
        _check(load_library().bodyfit_rccl_unique_id(buf))
        return bytes(buf)

    @classmethod
    def create(cls, uid: bytes, rank: int, size: int, device: int = 0) -> "Rccl":
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        h = C.c_void_p()
        _check(load_library().bodyfit_rccl_create(buf, rank, size, device, C.byref(h)))
        return cls(h)

    @classmethod
    def wrap(cls, nccl_comm_ptr: int, rank: int, size: int) -> "Rccl":
        h = C.c_void_p()
        _check(load_library().bodyfit_rccl_wrap(C.c_void_p(nccl_comm_ptr), rank, size, C.byref(h)))
        return cls(h)

    def count(self) -> tuple[int, int]:
        """(ranks, this rank) as RCCL itself reports them (ncclCommCount, ncclCommUserRank)."""
        n, r = C.c_int(), C.c_int()
        _check(load_library().bodyfit_rccl_count(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def allreduce_shared(self, d_buf66_ptr: int, stream: int | None = None):
        """The evaluation path's one collective: ncclAllReduce(sum, f64) of the 66 doubles, in place, on `stream`."""
        _check(load_library().bodyfit_allreduce_shared_rccl(self.h, d_buf66_ptr, stream))

    def close(self):
        if self.h:
            load_library().bodyfit_rccl_destroy(self.h)
            self.h = None


class DeviceViews(C.Structure):
    _fields_ = [("residuals", C.c_void_p), ("jacobian", C.c_void_p), ("gmm_comp", C.c_void_p),
                ("cloud", C.c_void_p), ("joints", C.c_void_p), ("normal_eq", C.c_void_p),
                ("cloud_frame_stride", C.c_longlong)]


_lib = None


def launch_count() -> int:
    """kernels launched through the library by this process so far (bodyfit_launch_count)"""
    return int(load_library().bodyfit_launch_count())


def declared_symbols() -> list[str]:
    """Every function include/bodyfit.h declares (used by the ABI test)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bodyfit_[a-z0-9_]+)\s*\(", txt)))


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BodyfitError(f"{LIB_PATH} is missing: build it with `make -C 3dbodyanimation_amd/csrc` "
                           "(__graft_entry__.build()). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; loading the system one first
    # makes torch.cuda see no GPU.  Importing torch first lets libbodyfit.so bind to the runtime torch
    # already mapped (same soname), so streams / tensors / RCCL and our kernels share one context.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.bodyfit_last_error.restype = C.c_char_p
    lib.bodyfit_mean_pixel_error.restype = C.c_double
    lib.bodyfit_mean_pixel_error.argtypes = [C.c_int, _ip, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.bodyfit_model_create.argtypes = [C.POINTER(_ModelDesc), C.c_int, C.POINTER(C.c_void_p)]
    lib.bodyfit_model_destroy.argtypes = [C.c_void_p]
    lib.bodyfit_model_get_derived.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.bodyfit_gmm_create.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_int, C.POINTER(C.c_void_p)]
    lib.bodyfit_gmm_destroy.argtypes = [C.c_void_p]
    lib.bodyfit_gmm_get.argtypes = [C.c_void_p, _dp, _dp]
    lib.bodyfit_problem_create.argtypes = [C.c_void_p, C.POINTER(_ProblemDesc), C.POINTER(C.c_void_p)]
    lib.bodyfit_problem_destroy.argtypes = [C.c_void_p]
    lib.bodyfit_problem_layout.argtypes = [C.c_void_p, C.POINTER(Layout)]
    lib.bodyfit_problem_views.argtypes = [C.c_void_p, C.POINTER(DeviceViews)]
    lib.bodyfit_evaluate_batch.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _ip, C.c_int]
    lib.bodyfit_evaluate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.bodyfit_reduce_shared_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bodyfit_arm_shared_reduction.argtypes = [C.c_void_p, C.c_void_p]
    lib.bodyfit_profile_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, _dp]
    lib.bodyfit_solve.argtypes = [C.c_void_p, _dp, _dp, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(FitOptions),
                                  C.POINTER(FitSummary), C.c_int]
    lib.bodyfit_solve_sharded.argtypes = [C.c_void_p, _dp, _dp, C.POINTER(C.c_ubyte), C.POINTER(Comm), C.POINTER(FitOptions),
                                          C.POINTER(FitSummary)]
    lib.bodyfit_solve_sharded_rccl.argtypes = [C.c_void_p, _dp, _dp, C.POINTER(C.c_ubyte), C.c_void_p, C.POINTER(FitOptions),
                                               C.POINTER(FitSummary)]
    lib.bodyfit_rccl_unique_id.argtypes = [C.POINTER(C.c_ubyte)]
    lib.bodyfit_rccl_create.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.bodyfit_rccl_wrap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.bodyfit_rccl_destroy.argtypes = [C.c_void_p]
    lib.bodyfit_rccl_destroy.restype = None
    lib.bodyfit_allreduce_shared_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bodyfit_rccl_count.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.bodyfit_sweep_status.argtypes = [C.c_void_p, C.c_void_p]
    lib.bodyfit_sweep_timeouts.argtypes = [C.c_void_p]
    lib.bodyfit_sweep_timeouts.restype = C.c_long
    lib.bodyfit_set_exchange_timeout.argtypes = [C.c_void_p, C.c_double]
    lib.bodyfit_set_shard_proxy.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.bodyfit_launch_count.argtypes = []
    lib.bodyfit_launch_count.restype = C.c_long
    lib.bodyfit_last_exchange_count.argtypes = [C.c_void_p]
    lib.bodyfit_last_exchange_count.restype = C.c_long
    lib.bodyfit_forward.argtypes = [C.c_void_p, _dp, _dp, _dp, _fp]
    lib.bodyfit_writeback_batch.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _fp, _dp]
    lib.bodyfit_evaluate_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_dp), _dp, C.POINTER(_dp)]
    _u8p = C.POINTER(C.c_uint8)
    _i32p = C.POINTER(C.c_int32)
    lib.bodyfit_overlay_create.argtypes = [C.POINTER(_OverlayDesc), C.POINTER(C.c_void_p)]
    lib.bodyfit_overlay_destroy.argtypes = [C.c_void_p]
    lib.bodyfit_overlay_destroy.restype = None
    lib.bodyfit_overlay_render_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p,
                                                  C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                                  C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.bodyfit_overlay_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, _u8p, C.c_size_t,
                                           C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                           C.c_int]
    lib.bodyfit_overlay_drawlist.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), _i32p, _i32p, _i32p]
    lib.bodyfit_overlay_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise BodyfitError(f"bodyfit status {rc}: {load_library().bodyfit_last_error().decode()}")


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _c32i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def device_count() -> int:
    return load_library().bodyfit_device_count()


# ------------------------------------------------------------------------------------------------
# reference-side value types
# ------------------------------------------------------------------------------------------------
@dataclass
class PixelKP:  # include/Sim3BA.h:9
    jid: int
    u: float
    v: float


@dataclass
class Sim3Params:  # include/Sim3BA.h:11-19   data = [s, aa(3), t(3)]
    data: np.ndarray = field(default_factory=lambda: np.array([1.0, 0, 0, 0, 0, 0, 3.0]))

    @property
    def scale(self):
        return self.data[0]

    @property
    def aa_root(self):
        return self.data[1:4]

    @property
    def trans(self):
        return self.data[4:7]


@dataclass
class FramePoseParams:  # include/MultiFrameBA.h:9-14
    scale: float = 1.0
    rootAA: np.ndarray = field(default_factory=lambda: np.zeros(3))
    rootT: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 3.0]))
    jointAA: np.ndarray = field(default_factory=lambda: np.zeros((24, 3)))  # index 0 unused

    def pack(self) -> np.ndarray:
        return np.concatenate([[self.scale], self.rootAA, self.rootT, self.jointAA[1:].reshape(-1)])

    @staticmethod
    def unpack(x) -> "FramePoseParams":
        j = np.zeros((24, 3))
        j[1:] = np.asarray(x[7:]).reshape(-1, 3)
        return FramePoseParams(float(x[0]), np.array(x[1:4]), np.array(x[4:7]), j)


class Model:
    """Device-resident SMPL model (ark::AvatarModel stand-in)."""

    def __init__(self, m, device: int = 0, pose_blend_data: bool = True):
        lib = load_library()
        self._keep = [_c64(m.v_template), _c64(m.shapedirs), _c64(m.posedirs) if pose_blend_data else None,
                      _c64(m.j_regressor), _c64(m.weights), _c32i(m.parent), _c32i(m.landmark_vid)]
        k = self._keep
        self.n_verts, self.n_joints, self.n_shape = m.v_template.shape[0], len(m.parent), m.shapedirs.shape[2]
        self.n_landmarks = len(m.landmark_vid)
        self.n_kp_regressors = getattr(m, "n_kp_regressors", 0)
        if self.n_kp_regressors:
            self._keep += [_c32i(m.kpreg_offset), _c32i(m.kpreg_vid), _c64(m.kpreg_weight)]
            reg = (self.n_kp_regressors, _i(self._keep[-3]), _i(self._keep[-2]), _d(self._keep[-1]))
        else:
            reg = (0, None, None, None)
        desc = _ModelDesc(self.n_verts, self.n_joints, self.n_shape, m.posedirs.shape[2] if pose_blend_data else 0,
                          _d(k[0]), _d(k[1]), _d(k[2]), _d(k[3]), _d(k[4]), _i(k[5]), self.n_landmarks, _i(k[6]), *reg)
        h = C.c_void_p()
        _check(lib.bodyfit_model_create(C.byref(desc), device, C.byref(h)))
        self.h = h
        self.device = device

    def derived(self):
        J0 = np.empty((self.n_joints, 3)); S = np.empty((3 * self.n_joints, self.n_shape))
        off = np.empty((self.n_joints, 3))
        _check(load_library().bodyfit_model_get_derived(self.h, _d(J0), _d(S), _d(off)))
        return J0, S, off

    def close(self):
        if getattr(self, "h", None):
            load_library().bodyfit_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Gmm:
    """Device-resident max-mixture pose prior (ark::GaussianMixture stand-in)."""

    def __init__(self, weights, means, covs, resid_scale=np.sqrt(0.5), device: int = 0):
        self.K, self.D = means.shape
        w, mu, cv = _c64(weights), _c64(means), _c64(covs)
        h = C.c_void_p()
        _check(load_library().bodyfit_gmm_create(self.K, self.D, _d(w), _d(mu), _d(cv), resid_scale, device, C.byref(h)))
        self.h = h

    def get(self):
        L = np.empty((self.K, self.D, self.D)); nlw = np.empty(self.K)
        _check(load_library().bodyfit_gmm_get(self.h, _d(L), _d(nlw)))
        return L, nlw

    def close(self):
        if getattr(self, "h", None):
            load_library().bodyfit_gmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Problem:
    """The residual blocks of one solve (what the reference adds to its ceres::Problem)."""

    def __init__(self, model: Model, kp_offset, kp_id, kp_uv, intr, R0, n_cols=86, use_shape=True,
                 beta_per_frame=False, pose_blend=True, beta_pose=0.0, gmm: Gmm | None = None, beta_shape=0.0,
                 lambda_temporal=0.0, temporal_halo=False, huber_delta=3.0, want_mesh=False):
        lib = load_library()
        self.model = model
        self.gmm = gmm
        ko, ki, ku, r0 = _c32i(kp_offset), _c32i(kp_id), _c64(kp_uv), _c64(R0)
        self.n_frames = len(ko) - 1
        desc = _ProblemDesc(self.n_frames, _i(ko), _i(ki), _d(ku), float(intr[0]), float(intr[1]), float(intr[2]),
                            float(intr[3]), _d(r0), int(n_cols), int(use_shape), int(beta_per_frame),
                            int(pose_blend), float(beta_pose), gmm.h if gmm is not None else None,
                            float(beta_shape), float(lambda_temporal), int(temporal_halo), float(huber_delta),
                            int(want_mesh))
        h = C.c_void_p()
        _check(lib.bodyfit_problem_create(model.h, C.byref(desc), C.byref(h)))
        self.h = h
        self.layout = Layout()
        _check(lib.bodyfit_problem_layout(self.h, C.byref(self.layout)))
        self.n_cols = n_cols
        self.want_mesh = want_mesh
        self.n_param_rows = self.n_frames + (1 if temporal_halo else 0)

    @classmethod
    def from_sequence(cls, model: Model, seq, **kw):
        return cls(model, seq.kp_offset, seq.kp_id, seq.kp_uv, seq.intr, seq.R0, **kw)

    def evaluate(self, frame_params, beta=None, want_jacobian=True):
        L = self.layout
        x = _c64(frame_params); b = _c64(beta) if beta is not None else None
        assert x.size == self.n_param_rows * N_FRAME_PARAMS, "frame_params must be [F(+1), 76]"
        r = np.empty(L.total_rows); comp = np.zeros(self.n_frames, np.int32)
        J = np.empty((L.reproj_rows, L.n_cols)) if want_jacobian else None
        _check(load_library().bodyfit_evaluate_batch(self.h, _d(x), _d(b), _d(r), _d(J), _i(comp), int(want_jacobian)))
        return r, J, comp

    def evaluate_device(self, d_params_ptr: int, d_beta_ptr: int | None, want_jacobian=True, stream: int | None = None):
        _check(load_library().bodyfit_evaluate_device(self.h, d_params_ptr, d_beta_ptr, int(want_jacobian), stream))

    def reduce_shared_device(self, d_out_ptr: int | None = None, stream: int | None = None):
        _check(load_library().bodyfit_reduce_shared_device(self.h, d_out_ptr, stream))

    def sweep_status(self, stream: int | None = None):
        """Waits for `stream`; raises if an asynchronous one-launch sweep since the last check left its cloud incomplete
        (bodyfit_sweep_status: the problem then uses the two-launch sweep, so evaluating again gives the whole result)."""
        _check(load_library().bodyfit_sweep_status(self.h, stream))

    def sweep_timeouts(self) -> int:
        """one-launch sweeps of this problem found incomplete since it was created (bodyfit_sweep_timeouts); 0 in a healthy run"""
        return int(load_library().bodyfit_sweep_timeouts(self.h))

    def set_exchange_timeout(self, seconds: float):
        """bound on every exchange / status read of this problem's sharded solves (bodyfit_set_exchange_timeout; 0: none)"""
        _check(load_library().bodyfit_set_exchange_timeout(self.h, float(seconds)))

    def set_shard_proxy(self, n_ranks: int, rank: int = 0):
        """measurement aid (bodyfit_set_shard_proxy): sharded solves through a one-rank communicator run as `rank` of `n_ranks`
        identical shards; n_ranks <= 1 switches it off"""
        _check(load_library().bodyfit_set_shard_proxy(self.h, int(n_ranks), int(rank)))

    def arm_shared_reduction(self, d_out_ptr: int | None):
        """Following Jacobian sweeps deposit [cost | g_beta | H_bb] in d_out_ptr at their own tail when they can (one-launch
        sweep, shared beta, <= 256 frames + prior tiles); reduce_shared_device(d_out_ptr) then launches nothing.  None disarms."""
        _check(load_library().bodyfit_arm_shared_reduction(self.h, d_out_ptr))

    def profile_sweep(self, d_params_ptr, d_beta_ptr, want_jacobian=True, with_reduce=False, iters=50, stream=None):
        ms = np.zeros(5)
        _check(load_library().bodyfit_profile_sweep(self.h, d_params_ptr, d_beta_ptr, int(want_jacobian),
                                                    int(with_reduce), int(iters), stream, _d(ms)))
        # (the prior workgroups ride on one of the launches; sweep_roles != 0: the sweep was ONE launch)
        return dict(frame_resjac=ms[0], mesh_blend_lbs=ms[2], reduce_shared=ms[3], sweep_roles=ms[4])

    def views(self) -> DeviceViews:
        v = DeviceViews()
        _check(load_library().bodyfit_problem_views(self.h, C.byref(v)))
        return v

    def forward(self, frame_params, beta=None, want_cloud=True):
        x = _c64(frame_params); b = _c64(beta) if beta is not None else None
        joints = np.empty((self.n_frames, self.model.n_joints, 3))
        cloud = np.empty((self.n_frames, self.model.n_verts, 3), np.float32) if want_cloud else None
        _check(load_library().bodyfit_forward(self.h, _d(x), _d(b), _d(joints),
                                              cloud.ctypes.data_as(_fp) if cloud is not None else None))
        return joints, cloud

    def writeback(self, frame_params, beta=None, want_cloud=False):
        """The reference's post-solve write-back for every frame, on the device (bodyfit_writeback_batch):
        R0' = R(rootAA) R0, update() without the Sim3 scale, mean pixel error of the FK keypoints."""
        x = _c64(frame_params); b = _c64(beta) if beta is not None else None
        F = self.n_frames
        r0 = np.empty((F, 3, 3)); joints = np.empty((F, self.model.n_joints, 3)); px = np.empty(F)
        cloud = np.empty((F, self.model.n_verts, 3), np.float32) if want_cloud else None
        _check(load_library().bodyfit_writeback_batch(self.h, _d(x), _d(b), _d(r0), _d(joints),
                                                      cloud.ctypes.data_as(_fp) if cloud is not None else None, _d(px)))
        return dict(R0=r0, joints=joints, cloud=cloud, mean_px=px)

    def solve(self, frame_params, beta=None, constant=None, independent=False, max_iters=100, scale_bounds=(0.3, 3.0),
              verbose=False, solver=0):
        """Ceres-like LM over this problem (bodyfit_solve).  Returns fitted params, beta, [FitSummary]."""
        x = _c64(frame_params).copy()
        b = _c64(beta).copy() if beta is not None else None
        cst = None
        if constant is not None:
            cst = np.ascontiguousarray(constant, dtype=np.uint8)
        n_sum = self.n_frames if independent else 1
        sums = (FitSummary * n_sum)()
        opt = FitOptions(int(max_iters), float(scale_bounds[0]), float(scale_bounds[1]), int(verbose), int(solver))
        _check(load_library().bodyfit_solve(self.h, _d(x), _d(b), cst.ctypes.data_as(C.POINTER(C.c_ubyte)) if cst is not None else None,
                                            int(independent), C.byref(opt), sums, n_sum))
        return x, b, list(sums)

    def solve_sharded(self, frame_params, beta, comm: "Comm", constant=None, max_iters=100, scale_bounds=(-1e300, 1e300),
                      verbose=False):
        """This rank's shard of one window (bodyfit_solve_sharded).  frame_params: the shard's rows (+ the halo row when the
        problem has one).  Returns the shard's fitted rows, beta (the same on every rank) and the FitSummary."""
        x = _c64(frame_params).copy()
        b = _c64(beta).copy()
        assert x.size == self.n_param_rows * N_FRAME_PARAMS
        cst = np.ascontiguousarray(constant, dtype=np.uint8) if constant is not None else None
        summ = FitSummary()
        opt = FitOptions(int(max_iters), float(scale_bounds[0]), float(scale_bounds[1]), int(verbose), 3)
        _check(load_library().bodyfit_solve_sharded(self.h, _d(x), _d(b),
                                                    cst.ctypes.data_as(C.POINTER(C.c_ubyte)) if cst is not None else None,
                                                    C.byref(comm), C.byref(opt), C.byref(summ)))
        return x.reshape(-1)[:self.n_frames * N_FRAME_PARAMS].reshape(self.n_frames, N_FRAME_PARAMS).copy(), b, summ

    def solve_sharded_rccl(self, frame_params, beta, comm: "Rccl", constant=None, max_iters=100,
                           scale_bounds=(-1e300, 1e300), verbose=False):
        """This rank's shard of one window with the exchanges as RCCL all-gathers on the solve's device buffers and stream
        (bodyfit_solve_sharded_rccl)."""
        x = _c64(frame_params).copy()
        b = _c64(beta).copy()
        assert x.size == self.n_param_rows * N_FRAME_PARAMS
        cst = np.ascontiguousarray(constant, dtype=np.uint8) if constant is not None else None
        summ = FitSummary()
        opt = FitOptions(max_iters, scale_bounds[0], scale_bounds[1], int(verbose), 3)
        _check(load_library().bodyfit_solve_sharded_rccl(self.h, _d(x), _d(b),
                                                         cst.ctypes.data_as(C.POINTER(C.c_ubyte)) if cst is not None else None,
                                                         comm.h, C.byref(opt), C.byref(summ)))
        return x.reshape(-1)[:self.n_frames * N_FRAME_PARAMS].reshape(self.n_frames, N_FRAME_PARAMS).copy(), b, summ

    def last_exchange_count(self) -> int:
        """all-gathers issued by the last sharded solve of this problem"""
        return int(load_library().bodyfit_last_exchange_count(self.h))

    def cache_sweep(self, params, beta=None):
        """One sweep kept in the problem's host cache for evaluate_block (what bodyfit_ceres::SweepCallback does): no caller
        buffers, so only the structurally non-zero Jacobian column blocks cross PCIe."""
        x = _c64(params)
        b = _c64(beta) if beta is not None else None
        _check(load_library().bodyfit_evaluate_batch(self.h, _d(x), _d(b) if b is not None else None, None, None, None, 1))

    def evaluate_block(self, kind: int, index: int, blocks: list[np.ndarray], n_res: int, want=None):
        """ceres::CostFunction::Evaluate on one block.  `want[b]` False -> jacobians[b] = NULL."""
        blocks = [_c64(b) for b in blocks]
        nb = len(blocks)
        params = (_dp * nb)(*[_d(b) for b in blocks])
        r = np.empty(n_res)
        jacs = [np.full((n_res, len(b)), np.nan) for b in blocks]
        if want is None:
            want = [True] * nb
        jp = (_dp * nb)(*[(_d(j) if w else None) for j, w in zip(jacs, want)])
        _check(load_library().bodyfit_evaluate_block(self.h, kind, index, params, _d(r), jp))
        return r, jacs

    def close(self):
        if getattr(self, "h", None):
            load_library().bodyfit_problem_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Overlay:
    """Mesh overlay on the device: smpl::render::renderSMPLMesh (include/RenderSMPLMesh.h:16-110) for a batch of
    frames.  `faces` are the model's triangles (AvatarModel::mesh, src/main_single_frame.cpp:185-188)."""

    def __init__(self, faces, n_vertices: int, width: int, height: int, max_frames: int = 1, device: int = 0):
        self.faces = _c32i(faces).reshape(-1, 3)
        self.n_vertices, self.width, self.height, self.max_frames = int(n_vertices), int(width), int(height), int(max_frames)
        d = _OverlayDesc(device, self.n_vertices, self.faces.shape[0], self.faces.ctypes.data_as(C.POINTER(C.c_int32)),
                         self.width, self.height, self.max_frames)
        h = C.c_void_p()
        _check(load_library().bodyfit_overlay_create(C.byref(d), C.byref(h)))
        self.h = h

    def render(self, cloud, images, intr, fill=True, backface_cull=True, wireframe=False):
        """cloud [F][V][3] float32 or float64 (camera coordinates), images [F][H][W][3] uint8, modified in place."""
        cloud = np.asarray(cloud)
        if cloud.dtype != np.float32:
            cloud = np.ascontiguousarray(cloud, dtype=np.float64)
        cloud = np.ascontiguousarray(cloud).reshape(-1, self.n_vertices, 3)
        F = cloud.shape[0]
        if not (isinstance(images, np.ndarray) and images.dtype == np.uint8 and images.flags.c_contiguous
                and images.size == F * self.height * self.width * 3):
            raise BodyfitError("Overlay.render: images must be a C-contiguous uint8 array [F][H][W][3]")
        _check(load_library().bodyfit_overlay_render(
            self.h, cloud.ctypes.data_as(C.c_void_p), int(cloud.dtype == np.float64), self.n_vertices * 3, F,
            images.ctypes.data_as(C.POINTER(C.c_uint8)), self.width * 3, self.width * 3 * self.height,
            float(intr[0]), float(intr[1]), float(intr[2]), float(intr[3]), int(fill), int(backface_cull), int(wireframe)))
        return images

    def render_device(self, d_cloud_ptr: int, cloud_is_f64: bool, cloud_frame_stride: int, n_frames: int, d_images_ptr: int,
                      intr, row_stride=None, frame_stride=None, fill=True, backface_cull=True, stream=None):
        rs = self.width * 3 if row_stride is None else int(row_stride)
        fs = rs * self.height if frame_stride is None else int(frame_stride)
        _check(load_library().bodyfit_overlay_render_device(
            self.h, d_cloud_ptr, int(cloud_is_f64), int(cloud_frame_stride), int(n_frames), d_images_ptr, rs, fs,
            float(intr[0]), float(intr[1]), float(intr[2]), float(intr[3]), int(fill), int(backface_cull), 0, stream))

    def drawlist(self, frame: int = 0):
        nf = self.faces.shape[0]
        face = np.zeros(nf, np.int32); pts = np.zeros((nf, 6), np.int32); gray = np.zeros(nf, np.int32)
        n = C.c_int(0)
        p32 = C.POINTER(C.c_int32)
        _check(load_library().bodyfit_overlay_drawlist(self.h, int(frame), C.byref(n), face.ctypes.data_as(p32),
                                                       pts.ctypes.data_as(p32), gray.ctypes.data_as(p32)))
        return face[:n.value], pts[:n.value], gray[:n.value]

    def last_timing(self):
        ms = (C.c_float * 4)()
        _check(load_library().bodyfit_overlay_last_timing(self.h, ms))
        return dict(faces=ms[0], order=ms[1], binning=ms[2], tiles=ms[3])

    def close(self):
        if getattr(self, "h", None):
            load_library().bodyfit_overlay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def renderSMPLMesh(cloud, faces, img, fx, fy, cx, cy, fill=True, backface_cull=True, wireframe=False, device=0):
    """include/RenderSMPLMesh.h:16-24, one frame: cloud [V][3] (or the reference's 3xV, column-major), img HxWx3 uint8
    drawn in place."""
    cloud = np.asarray(cloud)
    if cloud.ndim == 2 and cloud.shape[0] == 3 and cloud.shape[1] != 3:
        cloud = cloud.T
    ov = Overlay(faces, cloud.shape[0], img.shape[1], img.shape[0], 1, device)
    try:
        ov.render(cloud[None], img.reshape(1, *img.shape), (fx, fy, cx, cy), fill, backface_cull, wireframe)
    finally:
        ov.close()
    return img


def mean_pixel_error(jid, uv, joints, intr) -> float:
    jid = _c32i(jid); uv = _c64(uv); joints = _c64(joints)
    return load_library().bodyfit_mean_pixel_error(len(jid), _i(jid), _d(uv), _d(joints), float(intr[0]), float(intr[1]),
                                                   float(intr[2]), float(intr[3]))
