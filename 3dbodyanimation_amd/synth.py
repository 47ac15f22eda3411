"""Seeded synthetic SMPL-shaped model and keypoint sequences (no licensed SMPL data).

Sizes and conventions follow SURVEY.md §8(d): V=6890, 10 shape keys, 207 pose-blend columns,
24 joints, the SMPL kintree, BODY_25 keypoints = 14 FK joints + 11 vertex landmarks, camera rule
of /root/reference/src/main_single_frame.cpp:171-176 (f = 0.9 max(W,H), c = (W/2, H/2)), initial
state of src/main_multi_frame.cpp:88-101 (R0 = Ry(pi) diag(1,-1,1), t = (0,0,3), s = 1).

Everything here is plain numpy f64 and is product-side host code (bench.py and the tests use it to
make inputs); it is not the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

N_VERTS = 6890
N_JOINTS = 24
N_SHAPE = 10
N_POSE_FEAT = 207
N_FRAME_PARAMS = 76  # [s, rootAA(3), rootT(3), jointAA[1..23](3 each)]  include/Sim3BA.h:36-40

SMPL_PARENT = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], dtype=np.int32
)

# Approximate T-pose joint table (metres, y up, x to the body's left), hand-written for this project.
_REST_JOINTS = np.array(
    [
        [0.000, 0.000, 0.000],  # 0 pelvis
        [0.070, -0.090, 0.000],  # 1 l_hip
        [-0.070, -0.090, 0.000],  # 2 r_hip
        [0.000, 0.110, -0.020],  # 3 spine1
        [0.100, -0.470, 0.005],  # 4 l_knee
        [-0.100, -0.470, 0.005],  # 5 r_knee
        [0.000, 0.250, 0.000],  # 6 spine2
        [0.090, -0.870, -0.030],  # 7 l_ankle
        [-0.090, -0.870, -0.030],  # 8 r_ankle
        [0.000, 0.300, 0.020],  # 9 spine3
        [0.110, -0.930, 0.090],  # 10 l_foot
        [-0.110, -0.930, 0.090],  # 11 r_foot
        [0.000, 0.520, -0.020],  # 12 neck
        [0.080, 0.430, -0.010],  # 13 l_collar
        [-0.080, 0.430, -0.010],  # 14 r_collar
        [0.000, 0.600, 0.030],  # 15 head
        [0.170, 0.450, -0.020],  # 16 l_shoulder
        [-0.170, 0.450, -0.020],  # 17 r_shoulder
        [0.430, 0.440, -0.030],  # 18 l_elbow
        [-0.430, 0.440, -0.030],  # 19 r_elbow
        [0.680, 0.440, -0.020],  # 20 l_wrist
        [-0.680, 0.440, -0.020],  # 21 r_wrist
        [0.770, 0.430, -0.020],  # 22 l_hand
        [-0.770, 0.430, -0.020],  # 23 r_hand
    ]
)

# BODY_25 order -> keypoint id (id < 24: FK joint, id >= 24: landmark id - 24).  SURVEY.md §8(d).
BODY25_IDS = np.array(
    [24 + 0, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7,
     24 + 1, 24 + 2, 24 + 3, 24 + 4, 24 + 5, 24 + 6, 24 + 7, 24 + 8, 24 + 9, 24 + 10],
    dtype=np.int32,
)
# MediaPipe-mapped joints of the reference loader (include/Utils.h:22-23) incl. the two trailing
# zeros of the 17-slot array (quirk Q1: pelvis twice).
USE_SMPL_REFERENCE = np.array([1, 2, 4, 5, 7, 8, 10, 11, 15, 16, 17, 18, 19, 20, 21, 0, 0], dtype=np.int32)

# anatomical targets for the 11 landmark vertices (relative to the rest skeleton)
_LANDMARK_POINTS = np.array(
    [
        [0.000, 0.640, 0.130],  # nose
        [-0.035, 0.670, 0.105],  # r eye
        [0.035, 0.670, 0.105],  # l eye
        [-0.085, 0.650, 0.020],  # r ear
        [0.085, 0.650, 0.020],  # l ear
        [0.100, -0.960, 0.190],  # l big toe
        [0.150, -0.960, 0.160],  # l small toe
        [0.090, -0.950, -0.080],  # l heel
        [-0.100, -0.960, 0.190],  # r big toe
        [-0.150, -0.960, 0.160],  # r small toe
        [-0.090, -0.950, -0.080],  # r heel
    ]
)


@dataclass
class SynthModel:
    v_template: np.ndarray  # [V,3]
    shapedirs: np.ndarray  # [V,3,nS]
    posedirs: np.ndarray  # [V,3,P]
    j_regressor: np.ndarray  # [nJ,V]
    weights: np.ndarray  # [V,nJ]
    parent: np.ndarray  # [nJ] int32, root = -1
    landmark_vid: np.ndarray  # [nL] int32
    # derived (f64): initialJointPos, jointShapeReg  (avatar: jointShapeRegBase / jointShapeReg)
    J0: np.ndarray = field(default=None)
    S: np.ndarray = field(default=None)
    faces: np.ndarray = field(default=None)  # [nF,3] int32 triangles (AvatarModel::mesh); optional
    # sparse keypoint regressors over the POSED vertices (CSR; keypoint id n_joints + n_landmarks + r): OpenPose-style extra
    # keypoints that are a weighted mean of a few surface vertices instead of one vertex
    kpreg_offset: np.ndarray = field(default=None)  # [nR + 1] int32
    kpreg_vid: np.ndarray = field(default=None)     # [nnz] int32
    kpreg_weight: np.ndarray = field(default=None)  # [nnz] float64

    @property
    def n_kp_regressors(self):
        return 0 if self.kpreg_offset is None else len(self.kpreg_offset) - 1

    @property
    def n_verts(self):
        return self.v_template.shape[0]

    @property
    def n_joints(self):
        return self.parent.shape[0]

    @property
    def n_shape(self):
        return self.shapedirs.shape[2]

    def finalize(self):
        self.J0 = self.j_regressor @ self.v_template
        self.S = np.einsum("jv,vak->jak", self.j_regressor, self.shapedirs).reshape(3 * self.n_joints, -1)
        return self


def add_kp_regressors(model: "SynthModel", n_rows: int = 3, support: int = 12, seed: int = 0) -> "SynthModel":
    """Attach `n_rows` sparse regressor rows: each is a convex combination of the `support` vertices nearest to a landmark
    vertex (so a row spans several skinning joints), weights drawn once from a seeded generator."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(seed)
    tree = cKDTree(model.v_template)
    off, vid, w = [0], [], []
    for r in range(n_rows):
        centre = model.v_template[model.landmark_vid[(3 * r + 1) % len(model.landmark_vid)]]
        _, nn = tree.query(centre, k=support)
        c = rng.uniform(0.2, 1.0, support)
        vid += [int(i) for i in nn]
        w += list(c / c.sum())
        off.append(len(vid))
    model.kpreg_offset = np.asarray(off, np.int32)
    model.kpreg_vid = np.asarray(vid, np.int32)
    model.kpreg_weight = np.asarray(w, np.float64)
    return model


N_FACES = 13776  # SMPL: 2 V - 4


def make_faces(model: "SynthModel", n_faces: int = N_FACES, seed: int = 0) -> np.ndarray:
    """Synthetic connectivity for the overlay (the capsule point cloud has none): two triangles per vertex with its
    nearest neighbours, so the triangles have the size of the local vertex spacing and a random orientation (about half
    of them face the camera, as on a closed surface)."""
    from scipy.spatial import cKDTree

    v = model.v_template
    k = 2 * ((n_faces + len(v) - 1) // len(v)) + 1
    _, nn = cKDTree(v).query(v, k=min(k, len(v)))
    tris = []
    for c in range(1, nn.shape[1] - 1, 2):
        tris.append(np.stack([nn[:, 0], nn[:, c], nn[:, c + 1]], 1))
    f = np.concatenate(tris, 0)
    f = f[np.random.default_rng(seed).permutation(len(f))[:n_faces]]
    return np.ascontiguousarray(f, dtype=np.int32)


def _seg_dist(p, a, b):
    ab = b - a
    t = np.clip(((p - a) @ ab) / max(ab @ ab, 1e-12), 0.0, 1.0)
    return np.linalg.norm(p - (a + t[:, None] * ab), axis=1)


def make_model(seed: int = 0, n_verts: int = N_VERTS, n_landmarks: int = 11) -> SynthModel:
    rng = np.random.default_rng(seed)
    nJ = N_JOINTS
    J = _REST_JOINTS
    # vertices on capsules around bones (parent -> joint), count proportional to length * radius
    radius = np.full(nJ, 0.05)
    radius[[1, 2, 4, 5]] = 0.07
    radius[[3, 6, 9]] = 0.12
    radius[[12]] = 0.05
    radius[[15]] = 0.09
    radius[[18, 19, 20, 21, 22, 23]] = 0.035
    radius[[10, 11]] = 0.04
    ends = []
    for j in range(1, nJ):
        ends.append((J[SMPL_PARENT[j]], J[j], radius[j]))
    ends.append((J[15], J[15] + np.array([0, 0.12, 0.0]), 0.095))  # skull
    ends.append((J[10], J[10] + np.array([0, -0.03, 0.10]), 0.035))  # l toes
    ends.append((J[11], J[11] + np.array([0, -0.03, 0.10]), 0.035))  # r toes
    wts = np.array([max(np.linalg.norm(b - a), 0.05) * r for a, b, r in ends])
    counts = np.floor(wts / wts.sum() * n_verts).astype(int)
    counts[0] += n_verts - counts.sum()
    pts = []
    for (a, b, r), n in zip(ends, counts):
        t = rng.uniform(0, 1, n)
        d = rng.normal(size=(n, 3))
        ax = (b - a) / max(np.linalg.norm(b - a), 1e-9)
        d -= (d @ ax)[:, None] * ax
        d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-12
        pts.append(a + t[:, None] * (b - a) + r * d * rng.uniform(0.85, 1.0, n)[:, None])
    v = np.concatenate(pts, 0)
    v = v[rng.permutation(n_verts)]
    n_lm = min(n_landmarks, len(_LANDMARK_POINTS))
    landmark_vid = np.array(
        [int(np.argmin(np.linalg.norm(v - p, axis=1))) for p in _LANDMARK_POINTS[:n_lm]], dtype=np.int32
    )
    # skinning weights: softmax(-d^2/sigma^2) of the distance to each joint's bone, top-4, renormalised
    dist = np.empty((n_verts, nJ))
    for j in range(nJ):
        ch = np.where(SMPL_PARENT == j)[0]
        b = J[ch].mean(0) if len(ch) else J[j] + (J[j] - J[SMPL_PARENT[j]]) * 0.5
        dist[:, j] = _seg_dist(v, J[j], b)
    logit = -(dist**2) / (0.06**2)
    logit -= logit.max(1, keepdims=True)
    w = np.exp(logit)
    idx = np.argsort(-w, axis=1)[:, :4]
    W = np.zeros_like(w)
    np.put_along_axis(W, idx, np.take_along_axis(w, idx, 1), 1)
    W /= W.sum(1, keepdims=True)
    # joint regressor: 64 nearest vertices, non-negative, rows sum to 1
    Jreg = np.zeros((nJ, n_verts))
    k_sup = min(64, n_verts)
    for j in range(nJ):
        near = np.argsort(np.linalg.norm(v - J[j], axis=1))[:k_sup]
        c = rng.uniform(0.2, 1.0, k_sup)
        Jreg[j, near] = c / c.sum()
    # shape directions: smooth (affine in position) + small noise; pose directions: small noise
    shapedirs = np.empty((n_verts, 3, N_SHAPE))
    centre = v.mean(0)
    for k in range(N_SHAPE):
        B = rng.normal(scale=0.03, size=(3, 3))
        shapedirs[:, :, k] = (v - centre) @ B.T + rng.normal(scale=0.002, size=(n_verts, 3))
    posedirs = rng.normal(scale=0.002, size=(n_verts, 3, N_POSE_FEAT))
    return SynthModel(v, shapedirs, posedirs, Jreg, W, SMPL_PARENT.copy(), landmark_vid).finalize()


# ------------------------------------------------------------------------------------------------
# numpy forward (used to synthesise observations; independent of both the HIP path and the oracle)
# ------------------------------------------------------------------------------------------------
def rodrigues(aa: np.ndarray) -> np.ndarray:
    th2 = float(aa @ aa)
    K = np.array([[0, -aa[2], aa[1]], [aa[2], 0, -aa[0]], [-aa[1], aa[0], 0]])
    if th2 > np.finfo(np.float64).eps:
        th = np.sqrt(th2)
        k = K / th
        return np.eye(3) + np.sin(th) * k + (1 - np.cos(th)) * (k @ k)
    return np.eye(3) + K


R0_DEFAULT = np.diag([-1.0, -1.0, -1.0])  # Ry(pi) diag(1,-1,1)   src/main_single_frame.cpp:210-212


def forward_numpy(model: SynthModel, x: np.ndarray, beta: np.ndarray, R0: np.ndarray, pose_blend=True,
                  vids=None, use_shape=True):
    """Camera-frame joints [nJ,3] and vertices [len(vids),3] for one frame (x = 76 packed params)."""
    nJ = model.n_joints
    b = beta if use_shape else np.zeros(model.n_shape)
    s, raa, t = x[0], x[1:4], x[4:7]
    Rl = [np.eye(3)] + [rodrigues(x[7 + 3 * (j - 1): 10 + 3 * (j - 1)]) for j in range(1, nJ)]
    Jb = model.J0 + (model.S @ b).reshape(nJ, 3)
    Jc = Jb - Jb[0]
    A = [np.eye(3)] * nJ
    Pj = np.zeros((nJ, 3))
    for j in range(1, nJ):
        p = model.parent[j]
        A[j] = A[p] @ Rl[j]
        Pj[j] = Pj[p] + A[p] @ (Jc[j] - Jc[p])
    M = s * rodrigues(raa) @ R0
    joints = Pj @ M.T + t
    if vids is None:
        vids = np.arange(model.n_verts)
    vids = np.asarray(vids, dtype=np.int64)
    vp = model.v_template[vids] - Jb[0] + model.shapedirs[vids] @ b
    if pose_blend:
        feat = np.concatenate([(Rl[j] - np.eye(3)).reshape(-1) for j in range(1, nJ)])
        vp = vp + model.posedirs[vids] @ feat
    out = np.zeros((len(vids), 3))
    Wv = model.weights[vids]
    for j in range(nJ):
        wj = Wv[:, j]
        nz = wj != 0
        if nz.any():
            out[nz] += wj[nz, None] * ((vp[nz] - Jc[j]) @ A[j].T + Pj[j])
    return joints, out @ M.T + t


def project(X, intr):
    fx, fy, cx, cy = intr
    return np.stack([fx * X[:, 0] / X[:, 2] + cx, fy * X[:, 1] / X[:, 2] + cy], 1)


def camera_intrinsics(W=1920, H=1080):
    f = 0.9 * max(W, H)  # src/main_single_frame.cpp:171-176
    return np.array([f, f, 0.5 * W, 0.5 * H])


@dataclass
class SynthSequence:
    intr: np.ndarray  # [4] fx fy cx cy
    R0: np.ndarray  # [F,9]
    kp_offset: np.ndarray  # [F+1] int32
    kp_id: np.ndarray  # [Ktot] int32
    kp_uv: np.ndarray  # [Ktot,2]
    gt_params: np.ndarray  # [F,76]
    gt_beta: np.ndarray  # [nS]
    init_params: np.ndarray  # [F,76]

    @property
    def n_frames(self):
        return self.R0.shape[0]


def _smooth(a, taps=5):
    if a.shape[0] < 2:
        return a
    k = np.ones(taps) / taps
    pad = taps // 2
    ap = np.pad(a, ((pad, pad),) + ((0, 0),) * (a.ndim - 1), mode="edge")
    return np.apply_along_axis(lambda c: np.convolve(c, k, mode="valid"), 0, ap)


def make_sequence(model: SynthModel, n_frames: int, seed: int = 0, kp_ids=BODY25_IDS, noise_px=1.0,
                  beta_fixed=False, pose_sigma=0.25, W=1920, H=1080, ragged=False) -> SynthSequence:
    rng = np.random.default_rng(seed + 1000)
    nJ = model.n_joints
    F = n_frames
    theta = rng.normal(scale=pose_sigma, size=(F, nJ - 1, 3))
    theta[:, 21:, :] = 0.0  # joints 22, 23 (hands) stay at zero
    theta = _smooth(theta)
    raa = _smooth(rng.normal(scale=0.1, size=(F, 3)))
    t = np.array([0, 0, 3.0]) + np.cumsum(rng.normal(scale=0.02, size=(F, 3)), 0)
    beta = np.zeros(model.n_shape) if beta_fixed else rng.normal(size=model.n_shape)
    gt = np.zeros((F, N_FRAME_PARAMS))
    gt[:, 0] = 1.0
    gt[:, 1:4] = raa
    gt[:, 4:7] = t
    gt[:, 7:] = theta.reshape(F, -1)
    init = np.zeros_like(gt)
    init[:, 0] = 1.0
    init[:, 6] = 3.0
    intr = camera_intrinsics(W, H)
    R0 = np.tile(R0_DEFAULT.reshape(1, 9), (F, 1))
    kp_ids = np.asarray(kp_ids, dtype=np.int32)
    nL = len(model.landmark_vid)
    is_lm = (kp_ids >= nJ) & (kp_ids < nJ + nL)
    is_reg = kp_ids >= nJ + nL                       # sparse regressor rows over the posed vertices
    lm = kp_ids[is_lm] - nJ
    reg_rows = kp_ids[is_reg] - nJ - nL
    reg_vid = (np.concatenate([model.kpreg_vid[model.kpreg_offset[r]:model.kpreg_offset[r + 1]] for r in reg_rows])
               if len(reg_rows) else np.zeros(0, np.int32))
    offs, ids, uvs = [0], [], []
    for f in range(F):
        vids = np.concatenate([model.landmark_vid[lm], reg_vid]).astype(np.int64) if (len(lm) or len(reg_vid)) else []
        joints, lmk = forward_numpy(model, gt[f], beta, R0_DEFAULT, True, vids)
        pts = np.empty((len(kp_ids), 3))
        pts[kp_ids < nJ] = joints[kp_ids[kp_ids < nJ]]
        if len(lm):
            pts[is_lm] = lmk[:len(lm)]
        if len(reg_rows):
            o, rp = len(lm), []
            for r in reg_rows:
                e0, e1 = model.kpreg_offset[r], model.kpreg_offset[r + 1]
                rp.append(model.kpreg_weight[e0:e1] @ lmk[o:o + e1 - e0])
                o += e1 - e0
            pts[is_reg] = np.array(rp)
        uv = project(pts, intr) + rng.normal(scale=noise_px, size=(len(kp_ids), 2))
        keep = np.ones(len(kp_ids), bool)
        if ragged:  # drop a few keypoints per frame (visibility < 0.5 in the reference loader)
            keep = rng.uniform(size=len(kp_ids)) > 0.2
            if f % 7 == 3:
                keep[:] = False  # an empty frame ('[]' JSONs exist in the shipped data)
        ids.append(kp_ids[keep])
        uvs.append(uv[keep])
        offs.append(offs[-1] + int(keep.sum()))
    return SynthSequence(intr, R0, np.array(offs, np.int32), np.concatenate(ids).astype(np.int32),
                         np.concatenate(uvs, 0), gt, beta, init)


def make_gmm(seed: int = 0, n_comp: int = 8, dim: int = 69):
    """Synthetic SPD mixture with covariance eigenvalues log-uniform in [1e-3, 1.4] (the shipped
    data/avatar-model/pose_prior.txt has eigenvalues in [1.0e-3, 1.37])."""
    rng = np.random.default_rng(seed + 77)
    w = rng.uniform(0.5, 1.5, n_comp)
    w /= w.sum()
    means = rng.normal(scale=0.2, size=(n_comp, dim))
    covs = np.empty((n_comp, dim, dim))
    for k in range(n_comp):
        Q, _ = np.linalg.qr(rng.normal(size=(dim, dim)))
        ev = np.exp(rng.uniform(np.log(1e-3), np.log(1.4), dim))
        covs[k] = (Q * ev) @ Q.T
        covs[k] = 0.5 * (covs[k] + covs[k].T)
    return w, means, covs
