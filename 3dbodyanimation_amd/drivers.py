"""Driver-level compatibility: the file formats and the staging logic of the reference's
two executables, on top of the HIP evaluator and bodyfit_solve.  No images, no rendering.

  load_mp_json            MediaPipe-33 JSON -> PixelKP list          include/Utils.h:18-23,61-99 (incl. quirk Q1)
  load_pose_prior_txt     K D / weights / means / covariances         scripts/convert_gmm_to_avatar.py:14-29
  load_smpl_npz           SMPL v1.0.0 keys, root parent -> -1          scripts/npz_fixer.py:4-17
  intrinsics              f = 0.9 max(W,H), c = (W/2, H/2)             src/main_single_frame.cpp:171-176
  run_single              3dba_single: every frame fitted on its own  src/main_single_frame.cpp:192-270
  run_multi               3dba_multi: anchors, then sliding windows    src/main_multi_frame.cpp:85-217
The reference's quirks that change results are kept: Q3 frozen joints in the pose-only
single-frame fit, Q5 pixel error without the Sim3 scale, Q7 stage-1 poses are not written back, Q8 the root
orientation r[0] is left-multiplied after every solve (so it compounds in window overlaps), Q9 the stage-2
"beta lock" is an L2 prior of weight 1e5 toward zero on the first window frame's private copy of beta.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass

import numpy as np

from . import api, synth

MP_MAP = [-1, 23, 24, -1, 25, 26, -1, 27, 28, -1, 31, 32, -1, -1, -1, 0, 11, 12, 13, 14, 15, 16, -1, -1]
USE_SMPL = [1, 2, 4, 5, 7, 8, 10, 11, 15, 16, 17, 18, 19, 20, 21, 0, 0]  # 17 slots, two trailing zeros (Q1)


def load_mp_json(path: str, W: int, H: int) -> list[tuple[int, float, float]]:
    """include/Utils.h:61-99: visibility >= 0.5, pixel = normalised * (W, H), pelvis/chest as mid-points."""
    with open(path) as f:
        j = json.load(f)
    if not isinstance(j, list) or len(j) < 33:
        return []

    def num(o, k, d=None):
        v = o.get(k) if isinstance(o, dict) else None
        return v if isinstance(v, (int, float)) and not isinstance(v, bool) else d

    def mid(a, b):
        xa, ya, xb, yb = num(j[a], "x"), num(j[a], "y"), num(j[b], "x"), num(j[b], "y")
        if None in (xa, ya, xb, yb):
            return None
        return 0.5 * (xa + xb), 0.5 * (ya + yb), min(num(j[a], "visibility", 1.0), num(j[b], "visibility", 1.0))

    pel, ch = mid(23, 24), mid(11, 12)
    out = []
    for sid in USE_SMPL:
        if sid == 0:
            if pel is None:
                continue
            x, y, vis = pel
        elif sid == 6:
            if ch is None:
                continue
            x, y, vis = ch
        else:
            mp = MP_MAP[sid]
            if mp < 0:
                continue
            x, y = num(j[mp], "x"), num(j[mp], "y")
            if x is None or y is None:
                continue
            vis = num(j[mp], "visibility", 1.0)
        if vis < 0.5:
            continue
        out.append((sid, x * W, y * H))
    return out


@dataclass
class KeypointSequence:
    kp_offset: np.ndarray
    kp_id: np.ndarray
    kp_uv: np.ndarray
    names: list

    @property
    def n_frames(self):
        return len(self.kp_offset) - 1

    def frame(self, f):
        k0, k1 = self.kp_offset[f], self.kp_offset[f + 1]
        return self.kp_id[k0:k1], self.kp_uv[k0:k1]


def load_keypoint_folder(folder: str, W: int, H: int) -> KeypointSequence:
    names = sorted(n for n in os.listdir(folder) if n.lower().endswith(".json"))   # list_sorted, include/Utils.h:33-42
    offs, ids, uvs = [0], [], []
    for n in names:
        for sid, u, v in load_mp_json(os.path.join(folder, n), W, H):
            ids.append(sid); uvs.append((u, v))
        offs.append(len(ids))
    return KeypointSequence(np.array(offs, np.int32), np.array(ids, np.int32), np.array(uvs, float).reshape(-1, 2), names)


def load_pose_prior_txt(path: str):
    with open(path) as f:
        K, D = map(int, f.readline().split())
        w = np.array(f.readline().split(), float)
        mu = np.array([f.readline().split() for _ in range(K)], float)
        cov = np.array([f.readline().split() for _ in range(K)], float).reshape(K, D, D)
    return w, mu, cov


def load_smpl_npz(path: str, landmark_vid=()) -> synth.SynthModel:
    z = np.load(path, allow_pickle=True)
    jr = z["J_regressor"]
    jr = np.asarray(jr.item().todense() if jr.dtype == object else jr, dtype=np.float64)
    kt = np.asarray(z["kintree_table"]).astype(np.int64)
    parent = kt[0].copy()
    parent[parent == kt[1]] = -1                     # npz_fixer: the root's parent becomes -1
    parent[parent > 10 ** 6] = -1                    # the stock files store 2^32 - 1 for the root
    m = synth.SynthModel(np.asarray(z["v_template"], float), np.asarray(z["shapedirs"], float)[:, :, :10],
                         np.asarray(z["posedirs"], float), jr, np.asarray(z["weights"], float),
                         parent.astype(np.int32), np.asarray(landmark_vid, np.int32))
    if "f" in z.files:
        m.faces = np.ascontiguousarray(z["f"], dtype=np.int32)   # AvatarModel::mesh (src/main_single_frame.cpp:185-188)
    return m.finalize()


def intrinsics(W: int, H: int) -> np.ndarray:
    return synth.camera_intrinsics(W, H)


def _updated(gpu_model, kps, intr, r0, t, joint_aa, beta):
    """Avatar::update() + mean_pixel_error for frames whose write-back already happened (r[0] = r0, p = t, each with its
    own beta copy), on the device: bodyfit_writeback_batch with a zero root angle-axis."""
    F = r0.shape[0]
    off, kid, uv = kps
    x = np.zeros((F, 76)); x[:, 0] = 1.0; x[:, 4:7] = t; x[:, 7:] = joint_aa
    p = api.Problem(gpu_model, off, kid, uv, intr, r0.reshape(F, 9), n_cols=86, use_shape=True,
                    beta_per_frame=(np.ndim(beta) == 2))
    wb = p.writeback(x, beta)
    return wb["joints"], wb["mean_px"]


def render_overlays(gpu_model, faces, kps, intr, r0, t, joint_aa, beta, size, frames_bgr=None, out_dir=None, names=None):
    """smpl::render::renderSMPLMesh on every frame's updated avatar (src/main_single_frame.cpp:273-277,
    src/main_multi_frame.cpp:205-229), batched: Avatar::update() on the device with the vertices left resident
    (bodyfit_writeback_batch), then bodyfit_overlay_render_device into device images.  `size` = (W, H); frames_bgr
    [F][H][W][3] uint8 are the video frames to draw on (black when None).  Returns the overlays; with out_dir they are
    also written as binary PPM (PNG encoding is outside the path)."""
    import torch

    F = r0.shape[0]
    W, H = int(size[0]), int(size[1])
    off, kid, uv = kps
    x = np.zeros((F, 76)); x[:, 0] = 1.0; x[:, 4:7] = t; x[:, 7:] = joint_aa
    p = api.Problem(gpu_model, off, kid, uv, intr, r0.reshape(F, 9), n_cols=86, use_shape=True,
                    beta_per_frame=(np.ndim(beta) == 2), want_mesh=True)
    p.writeback(x, beta, want_cloud=True)
    v = p.views()
    imgs = (torch.zeros((F, H, W, 3), dtype=torch.uint8, device="cuda") if frames_bgr is None
            else torch.from_numpy(np.ascontiguousarray(frames_bgr, dtype=np.uint8)).cuda())
    ov = api.Overlay(faces, gpu_model.n_verts, W, H, max_frames=F)
    ov.render_device(v.cloud, False, v.cloud_frame_stride, F, imgs.data_ptr(), intr)
    torch.cuda.synchronize()
    out = imgs.cpu().numpy()
    ov.close(); p.close()
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        for k in range(F):
            with open(os.path.join(out_dir, (names[k] if names else f"frame_{k}_render") + ".ppm"), "wb") as f:
                f.write(b"P6\n%d %d\n255\n" % (W, H))
                f.write(out[k, :, :, ::-1].tobytes())
    return out


def _subsequence(seq, ids):
    offs, kid, uv = [0], [], []
    for f in ids:
        a, b = seq.frame(f)
        kid.append(a); uv.append(b); offs.append(offs[-1] + len(a))
    return (np.array(offs, np.int32), np.concatenate(kid) if kid else np.zeros(0, np.int32),
            np.concatenate(uv) if uv else np.zeros((0, 2)))


def run_single(gpu_model, seq: KeypointSequence, intr, max_iters=100, beta_pose=20.0, beta_shape=30.0, opt_shape=False,
               gmm: api.Gmm | None = None, out_dir: str | None = None, faces=None, image_size=None, frames_bgr=None):
    """3dba_single: all frames in one batched solve, each with its own LM state (the frames never interact).
    Frames without keypoints are skipped as in src/main_single_frame.cpp:200-203.  In the pose-only mode the
    reference's shape block only carries its own prior and stays at zero (include/Sim3BA.h:630-637), so it is
    not built here."""
    keep = [int(f) for f in np.where(np.diff(seq.kp_offset) > 0)[0]]
    n = len(keep)
    if n == 0:
        return dict(frames=[], params=np.zeros((0, 76)), beta=None, r0=np.zeros((0, 3, 3)), log=[], summaries=[])
    off, kid, uv = _subsequence(seq, keep)
    R0 = np.tile(synth.R0_DEFAULT.reshape(1, 9), (n, 1))
    x0 = np.zeros((n, 76)); x0[:, 0] = 1.0; x0[:, 6] = 3.0
    shape_block = bool(opt_shape and beta_shape > 0.0)
    t0 = time.perf_counter()
    prob = api.Problem(gpu_model, off, kid, uv, intr, R0, n_cols=86 if shape_block else 76, use_shape=shape_block,
                       beta_per_frame=shape_block, beta_pose=beta_pose, gmm=gmm,
                       beta_shape=beta_shape if shape_block else 0.0)
    const = None
    if not opt_shape:                                 # OptimizePoseReprojection freezes 10, 11, 22, 23 (Q3)
        const = np.zeros(76, np.uint8)
        for j in (10, 11, 22, 23):
            const[7 + 3 * (j - 1):10 + 3 * (j - 1)] = 1
    x, beta, summ = prob.solve(x0, np.zeros((n, 10)) if shape_block else None, constant=const, independent=True,
                               max_iters=max_iters)
    ms = (time.perf_counter() - t0) * 1e3 / n
    wb = prob.writeback(x, beta)       # R0' = R(rootAA) R0, update() without the scale (Q5), mean pixel error: on the device
    r0_new = wb["R0"]
    rows = [(f, float(wb["mean_px"][k]), ms) for k, f in enumerate(keep)]
    _write_log(out_dir, rows)
    overlays = None
    if faces is not None and image_size is not None:               # frame_<i>_render (src/main_single_frame.cpp:273-277)
        bsh = beta if beta is not None else np.zeros(10)
        overlays = render_overlays(gpu_model, faces, (off, kid, uv), intr, r0_new, x[:, 4:7], x[:, 7:], bsh, image_size,
                                   None if frames_bgr is None else np.asarray(frames_bgr)[keep], out_dir,
                                   [f"frame_{f}_render" for f in keep])
    return dict(frames=keep, params=x, beta=beta, r0=r0_new, log=rows, summaries=summ, overlays=overlays)


def run_multi(gpu_model, seq: KeypointSequence, intr, max_iters_s1=1000, skip=10, wsize=20, overlap=5, beta_pose=5.0,
              beta_shape=25.0, lambda_t=3.0, out_dir: str | None = None, stage2_iters=60, faces=None, image_size=None,
              frames_bgr=None, trace: list | None = None):
    """3dba_multi: stage 1 on the anchors (shared beta), stage 2 on sliding windows with the beta 'lock'.
    trace (a list): receives a snapshot of the whole state (poses, r0, t, joint_aa, w) after every stage, for stage-by-stage
    comparisons (tests/staged_oracle.py)."""
    F = seq.n_frames
    r0 = np.tile(synth.R0_DEFAULT.reshape(1, 3, 3), (F, 1, 1))      # avatars[i]->r[0]
    t = np.tile(np.array([0.0, 0.0, 3.0]), (F, 1))                  # avatars[i]->p
    jaa = np.zeros((F, 69))                                         # avatars[i]->r[1..] as angle-axis
    w = np.zeros((F, 10))                                           # avatars[i]->w (private copies)
    poses = np.zeros((F, 76)); poses[:, 0] = 1.0; poses[:, 6] = 3.0  # FramePoseParams
    rows = []
    stage2 = []      # FitSummary of every stage-2 window

    def solve(ids, x_init, w_block, bshape, iters):
        off, kid, uv = _subsequence(seq, ids)
        prob = api.Problem(gpu_model, off, kid, uv, intr, r0[ids].reshape(len(ids), 9), n_cols=86,
                           use_shape=bshape > 0.0, beta_pose=beta_pose, beta_shape=bshape,
                           lambda_temporal=lambda_t if len(ids) > 1 else 0.0)
        x, b, s = prob.solve(x_init, w_block, independent=False, max_iters=iters, scale_bounds=(-1e300, 1e300))
        for k, f in enumerate(ids):                                 # write-back, include/MultiFrameBA.h:154-173
            r0[f] = synth.rodrigues(x[k, 1:4]) @ r0[f]              # compounds on later solves (Q8)
            t[f] = x[k, 4:7]
            jaa[f] = x[k, 7:]
        return x, b, s

    # ---- stage 1: anchors ---------------------------------------------------------------------------------
    anchors = list(range(0, F, skip))
    t0 = time.perf_counter()
    _, b1, s1 = solve(anchors, poses[anchors].copy(), w[anchors[0]].copy(), beta_shape, max_iters_s1)   # Q7: copy
    w[anchors[0]] = b1
    ms_anchor = (time.perf_counter() - t0) * 1e3
    _, px = _updated(gpu_model, _subsequence(seq, anchors), intr, r0[anchors], t[anchors], jaa[anchors], w[anchors])
    for k, f in enumerate(anchors):                                 # each avatar's own w (:141-147)
        rows.append((f, float(px[k]), ms_anchor / len(anchors)))
    w[:] = w[0]                                                     # share the shape among all avatars (:154)

    def snap(ids):
        if trace is not None:
            trace.append(dict(ids=list(ids), poses=poses.copy(), r0=r0.copy(), t=t.copy(), joint_aa=jaa.copy(), w=w.copy()))

    snap(anchors)
    # ---- stage 2: sliding windows -----------------------------------------------------------------------------
    stride = wsize - overlap
    for s in range(0, F, stride):
        e = min(s + wsize, F)
        ids = list(range(s, e))
        t0 = time.perf_counter()
        x, bw, s2 = solve(ids, poses[ids].copy(), w[s].copy(), 1e5, stage2_iters)     # beta lock (Q9)
        stage2.append(s2[0])
        w[s] = bw
        poses[ids] = x
        snap(ids)
        ms_win = (time.perf_counter() - t0) * 1e3
        _, px = _updated(gpu_model, _subsequence(seq, ids), intr, r0[ids], t[ids], jaa[ids], w[ids])
        for k, f in enumerate(ids):
            rows.append((f, float(px[k]), ms_win / (e - s)))
    _write_log(out_dir, rows)
    overlays = None
    if faces is not None and image_size is not None:
        # the reference renders a frame once no later window touches it (src/main_multi_frame.cpp:205-229): that is its
        # final state, so all frames are drawn here in one batch
        overlays = render_overlays(gpu_model, faces, _subsequence(seq, list(range(F))), intr, r0, t, jaa, w, image_size,
                                   frames_bgr, out_dir, [f"frame_{f}_multi" for f in range(F)])
    return dict(poses=poses, r0=r0, t=t, joint_aa=jaa, w=w, log=rows, stage1=s1[0], stage2=stage2, overlays=overlays)


def _write_log(out_dir, rows):
    if out_dir is None:
        return
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "log.csv")
    new = not os.path.exists(path)                                   # append mode, header if missing (:257-270)
    with open(path, "a") as f:
        if new:
            f.write("frame,mean_pixel_error_px,time_ms\n")
        for fr, px, ms in rows:
            f.write(f"{fr},{px},{ms}\n")
