"""GPU parity: the HIP path through the C ABI against the CPU restatement (oracle/) on identical seeded
inputs, and against the committed golden fixtures.  f64 residual/Jacobian tolerance 1e-9 abs (scaled by
the Jacobian magnitude ~6e2 -> 1e-12 relative); f32 mesh tolerance stated per test."""
import os

import numpy as np
import pytest

from conftest import random_params

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _oracle_full(oracle_mod, omodel, seq, x, beta, n_cols, use_shape, pose_blend, beta_pose, ogmm, beta_shape,
                 lam, halo=False):
    """The whole batched residual vector in the ABI's row layout, from the oracle."""
    F = len(seq.kp_offset) - 1
    r, J = omodel.evaluate_batch(seq, x[:F], beta, n_cols, use_shape, pose_blend, mode=0)
    parts = [r]
    comps = np.zeros(F, int)
    if beta_pose > 0:
        for f in range(F):
            rp, _, k = oracle_mod.pose_prior(ogmm, beta_pose, x[f, 7:])
            parts.append(rp); comps[f] = k
    if beta_shape > 0 and n_cols > 76:
        parts.append(beta_shape * np.asarray(beta).reshape(-1))
    if lam > 0:
        n_pairs = F - 1 + (1 if halo else 0)
        for f in range(n_pairs):
            a, b = x[f], x[f + 1]
            parts.append(lam * np.concatenate([a[4:7] - b[4:7], a[1:4] - b[1:4], a[7:] - b[7:]]))
    return np.concatenate(parts), J, comps


@pytest.mark.parametrize("cfg", ["pose_only_76", "shape_shared", "shape_per_frame_gmm", "shape_unused_Q12"])
def test_residual_and_jacobian_match_oracle(api, synth, model, gpu_model, oracle_mod, omodel, cfg):
    F = 37
    seq = synth.make_sequence(model, F, seed=5, ragged=True)
    rng = np.random.default_rng(21)
    x = random_params(rng, F)
    x[3, 7:] = 0.0           # all-zero pose: first-order Rodrigues branch
    x[4, 7:10] = 1e-9        # tiny but non-zero angle inside the first-order branch
    x[5, 7:10] = 3e-8        # just above the branch threshold
    kw = dict(pose_blend=True, huber_delta=3.0)
    ogmm = None
    if cfg == "pose_only_76":
        beta, n_cols, use_shape = None, 76, False
        kw.update(n_cols=76, use_shape=False, beta_pose=20.0)
        bp, bs, lam = 20.0, 0.0, 0.0
    elif cfg == "shape_shared":
        beta, n_cols, use_shape = rng.normal(size=10), 86, True
        kw.update(n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)
        bp, bs, lam = 5.0, 25.0, 3.0
    elif cfg == "shape_per_frame_gmm":
        beta, n_cols, use_shape = rng.normal(size=(F, 10)), 86, True
        w, mu, cov = synth.make_gmm(0)
        ogmm = oracle_mod.OracleGmm(w, mu, cov)
        kw.update(n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0, gmm=api.Gmm(w, mu, cov),
                  beta_shape=30.0)
        bp, bs, lam = 20.0, 30.0, 0.0
    else:
        beta, n_cols, use_shape = rng.normal(size=10), 86, False
        kw.update(n_cols=86, use_shape=False)
        bp, bs, lam = 0.0, 0.0, 0.0
    prob = api.Problem.from_sequence(gpu_model, seq, **kw)
    r, J, comp = prob.evaluate(x, beta, True)
    ro, Jo, co = _oracle_full(oracle_mod, omodel, seq, x, beta if beta is not None else np.zeros(10), n_cols,
                              use_shape, True, bp, ogmm, bs, lam)
    assert r.shape == ro.shape
    assert np.abs(r - ro).max() < 1e-9
    assert np.abs(J - Jo).max() < 1e-9 * max(1.0, np.abs(Jo).max())
    if ogmm is not None:
        assert np.array_equal(comp, co)
    # residual-only sweep returns the same residuals
    r2, J2, _ = prob.evaluate(x, beta, False)
    assert J2 is None and np.array_equal(r, r2)


def test_golden_fixture(api, gpu_model):
    g = np.load(os.path.join(GOLD, "oracle_golden.npz"))
    prob = api.Problem(gpu_model, g["kp_offset"], g["kp_id"], g["kp_uv"], g["intr"], g["R0"], n_cols=86,
                       use_shape=True, want_mesh=True)
    r, J, _ = prob.evaluate(g["params"], g["beta"], True)
    assert np.abs(r - g["r"]).max() < 1e-9 and np.abs(J - g["J"]).max() < 1e-9 * np.abs(g["J"]).max()
    joints, cloud = prob.forward(g["params"], g["beta"])
    assert np.abs(joints[0] - g["joints0"]).max() < 1e-11
    assert np.abs(cloud[0][g["cloud_vids"]] - g["cloud0_sample"]).max() < 5e-6  # f32 mesh, metres


def test_reference_pose_prior_data(api, gpu_model, synth, model, oracle_mod):
    """The reference's own pose_prior.txt (8 x 69) through the HIP GMM sweep vs the oracle."""
    g = np.load(os.path.join(GOLD, "pose_prior_reference.npz"))
    gm = api.Gmm(g["weights"], g["means"], g["covs"])
    og = oracle_mod.OracleGmm(g["weights"], g["means"], g["covs"])
    L, nlw = gm.get(); Lo, nlwo = og.get()
    assert np.abs(L - Lo).max() < 1e-9 * np.abs(Lo).max() and np.abs(nlw - nlwo).max() < 1e-9
    F = 16
    seq = synth.make_sequence(model, F, seed=9)
    rng = np.random.default_rng(5)
    x = random_params(rng, F, pose_sigma=0.4)
    x[0, 7:] = 0.0
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=76, use_shape=False, beta_pose=20.0, gmm=gm)
    r, _, comp = prob.evaluate(x, None, False)
    L_ = prob.layout
    rp = r[L_.reproj_rows:].reshape(F, 70)
    for f in range(F):
        ro, _, k = oracle_mod.pose_prior(og, 20.0, x[f, 7:])
        assert comp[f] == k and np.abs(rp[f] - ro).max() < 1e-8
    assert comp[0] == int(g["comp_at_zero"]) and np.abs(rp[0] - 20.0 * g["resid_at_zero"]).max() < 1e-8


def test_derived_tables_match(gpu_model, omodel):
    J0, S, off = gpu_model.derived()
    J0o, So, offo = omodel.derived()
    assert np.abs(J0 - J0o).max() < 1e-13 and np.abs(S - So).max() < 1e-13 and np.abs(off - offo).max() < 1e-13


@pytest.mark.parametrize("F", [1, 33, 256, 300, 1024])   # 300 / 1024: several 32-frame units per wave
def test_mesh_forward_matches_oracle(api, synth, model, gpu_model, omodel, F):
    seq = synth.make_sequence(model, F, seed=3)
    rng = np.random.default_rng(31)
    x = random_params(rng, F)
    x[0, 7:] = 0.0
    beta = rng.normal(size=(F, 10))
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_per_frame=True, want_mesh=True)
    joints, cloud = prob.forward(x, beta)
    check = sorted(set([0, F // 2, F - 1, min(F - 1, 257), min(F - 1, 288), (3 * F) // 4]))
    worst = 0.0
    for f in check:
        jo, co = omodel.forward(x[f], beta[f], seq.R0[f])
        assert np.abs(joints[f] - jo).max() < 1e-11
        worst = max(worst, np.abs(cloud[f] - co).max())
    # f32 skinning of metre-scale coordinates (ulp 2.4e-7 at 3 m) + bf16x2-split pose blend (<= 2^-16 relative)
    assert worst < 5e-6, worst
    # size-independent properties at full size: zero pose/shape gives the template; rigid motion commutes
    x0 = np.zeros((F, 76)); x0[:, 0] = 1.0
    eyeR0 = np.tile(np.eye(3).reshape(1, 9), (F, 1))
    prob0 = api.Problem(gpu_model, seq.kp_offset, seq.kp_id, seq.kp_uv, seq.intr, eyeR0, n_cols=86, use_shape=True,
                        beta_per_frame=True, want_mesh=True)
    _, c0 = prob0.forward(x0, np.zeros((F, 10)))
    J0 = gpu_model.derived()[0]
    assert np.abs(c0[F - 1] - (model.v_template - J0[0])).max() < 1e-6
    assert np.abs(c0[0] - c0[F - 1]).max() == 0.0


def test_mesh_without_pose_blend_and_landmark_consistency(api, synth, model, gpu_model, omodel):
    F = 8
    seq = synth.make_sequence(model, F, seed=4)
    rng = np.random.default_rng(41)
    x = random_params(rng, F); beta = rng.normal(size=10)
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, pose_blend=False, want_mesh=True)
    _, cloud = prob.forward(x, beta)
    _, co = omodel.forward(x[2], beta, seq.R0[2], pose_blend=False)
    assert np.abs(cloud[2] - co).max() < 5e-6
    # f64 landmark keypoints of the residual kernel agree with the f32 mesh vertices they name
    prob2 = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, pose_blend=True, want_mesh=True)
    r, _, _ = prob2.evaluate(x, beta, False)
    _, cloud2 = prob2.forward(x, beta)
    for f in [0, 5]:
        for k in range(seq.kp_offset[f], seq.kp_offset[f + 1]):
            if seq.kp_id[k] >= 24:
                X = cloud2[f][model.landmark_vid[seq.kp_id[k] - 24]].astype(np.float64)
                uv = np.array([seq.intr[0] * X[0] / X[2] + seq.intr[2], seq.intr[1] * X[1] / X[2] + seq.intr[3]])
                assert np.abs((uv - seq.kp_uv[k]) - r[2 * k:2 * k + 2]).max() < 5e-3  # pixels


def test_evaluate_block_is_ceres_evaluate(api, synth, model, gpu_model, oracle_mod, omodel):
    F = 5
    seq = synth.make_sequence(model, F, seed=6)
    rng = np.random.default_rng(51)
    x = random_params(rng, F); beta = rng.normal(size=10)
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0,
                                     lambda_temporal=3.0)
    prob.evaluate(x, beta, True)  # EvaluationCallback-style: one sweep, then per-block reads
    k = int(seq.kp_offset[2]) + 3
    f = 2
    blocks = [x[f, 0:1], x[f, 1:4], x[f, 4:7]] + [x[f, 7 + 3 * j:10 + 3 * j] for j in range(23)] + [beta]
    want = [True] * 27
    want[5] = False  # a constant block: jacobians[b] == NULL (include/Sim3BA.h:608-611)
    r, jacs = prob.evaluate_block(0, k, blocks, 2, want)
    ro, Jo = omodel.kp_block(seq.kp_id[k], seq.kp_uv[k], seq.intr, seq.R0[f], np.concatenate([x[f], beta]))
    assert np.abs(r - ro).max() < 1e-9
    Jcat = np.concatenate([j for j in jacs], axis=1)
    cols = np.ones(86, bool); cols[7 + 3 * 2:10 + 3 * 2] = False
    assert np.abs(Jcat[:, cols] - Jo[:, cols]).max() < 1e-9 * np.abs(Jo).max()
    assert np.isnan(jacs[5]).all()  # untouched
    # a block evaluated at parameters that differ from the last sweep triggers a fresh device sweep
    blocks2 = [b.copy() for b in blocks]; blocks2[4] = blocks2[4] + 0.05
    x2 = np.concatenate(blocks2[:26])
    r2, jacs2 = prob.evaluate_block(0, k, blocks2, 2)
    ro2, Jo2 = omodel.kp_block(seq.kp_id[k], seq.kp_uv[k], seq.intr, seq.R0[f], np.concatenate([x2, beta]))
    assert np.abs(r2 - ro2).max() < 1e-9 and np.abs(np.concatenate(jacs2, 1) - Jo2).max() < 1e-9 * np.abs(Jo2).max()
    # pose prior (L2), shape prior, temporal
    rp, jp = prob.evaluate_block(1, 1, [x[1, 7 + 3 * j:10 + 3 * j] for j in range(23)], 69)
    assert np.allclose(rp, 5.0 * x[1, 7:]) and np.allclose(jp[4][12:15], 5.0 * np.eye(3)) and jp[4][:12].max() == 0
    rs, js = prob.evaluate_block(2, 0, [beta], 10)
    assert np.allclose(rs, 25.0 * beta) and np.allclose(js[0], 25.0 * np.eye(10))
    rt, jt = prob.evaluate_block(3, 0, [x[0, 4:7], x[1, 4:7]], 3)
    assert np.allclose(rt, 3.0 * (x[0, 4:7] - x[1, 4:7])) and np.allclose(jt[1], -3.0 * np.eye(3))


def test_shared_reduction_matches_numpy(api, synth, model, gpu_model, oracle_mod):
    import torch
    F = 40
    seq = synth.make_sequence(model, F, seed=8, noise_px=4.0)
    rng = np.random.default_rng(61)
    x = seq.gt_params + rng.normal(scale=0.02, size=seq.gt_params.shape)
    beta = seq.gt_beta + 0.1
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0,
                                     lambda_temporal=3.0)
    r, J, _ = prob.evaluate(x, beta, True)
    out = torch.zeros(66, dtype=torch.float64, device="cuda")
    prob.reduce_shared_device(out.data_ptr(), None)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    K = prob.layout.n_keypoints
    rk = r[:2 * K].reshape(K, 2); Jb = J[:, 76:].reshape(K, 2, 10)
    s = (rk ** 2).sum(1)
    rho = np.array([oracle_mod.huber(3.0, v) for v in s])
    assert (s > 9).any() and (s <= 9).any()  # both Huber regions are exercised
    cost = 0.5 * rho[:, 0].sum() + 0.5 * (r[2 * K:] ** 2).sum()
    g = np.einsum("k,kri,kr->i", rho[:, 1], Jb, rk) + 25.0 * (25.0 * beta)
    H = np.einsum("k,kri,krj->ij", rho[:, 1], Jb, Jb) + 25.0 ** 2 * np.eye(10)
    assert abs(got[0] - cost) < 1e-9 * cost
    assert np.abs(got[1:11] - g).max() < 1e-9 * np.abs(g).max()
    assert np.abs(got[11:] - H[np.triu_indices(10)]).max() < 1e-9 * np.abs(H).max()


def test_invalid_arguments_fail_loudly(api, synth, model, gpu_model):
    seq = synth.make_sequence(model, 2, seed=1)
    with pytest.raises(api.BodyfitError):
        api.Problem.from_sequence(gpu_model, seq, n_cols=80)
    bad = seq.kp_id.copy(); bad[0] = 99
    with pytest.raises(api.BodyfitError):
        api.Problem(gpu_model, seq.kp_offset, bad, seq.kp_uv, seq.intr, seq.R0)
    with pytest.raises(api.BodyfitError):
        api.Problem.from_sequence(gpu_model, seq, n_cols=76, use_shape=True)


def test_reference_keypoint_files(api, synth, model, gpu_model, oracle_mod, omodel):
    """The reference's own data/keypoints/video1/*.json (reduced to PixelKP lists by a restated load_mp_json,
    tests/golden/make_golden.py): 38 frames, 5 of them empty, 12-16 keypoints, the pelvis twice (quirk Q1),
    480x270 images -> fx = fy = 432.  Ragged + empty frames through the HIP path vs the oracle, then the
    3dba_single fit of the non-empty frames vs the dense LM."""
    g = np.load(os.path.join(GOLD, "video1_keypoints.npz"))
    W, H = int(g["W"]), int(g["H"])
    f_ = 0.9 * max(W, H)
    class S: pass
    seq = S(); seq.kp_offset = g["kp_offset"]; seq.kp_id = g["kp_id"]; seq.kp_uv = g["kp_uv"]
    F = len(seq.kp_offset) - 1
    seq.intr = np.array([f_, f_, 0.5 * W, 0.5 * H]); seq.R0 = np.tile(synth.R0_DEFAULT.reshape(1, 9), (F, 1))
    assert F == 38 and (np.diff(seq.kp_offset) == 0).sum() == 5
    counts = np.bincount(seq.kp_id[seq.kp_offset[5]:seq.kp_offset[6]], minlength=24)
    assert counts[0] == 2  # Q1: the pelvis keypoint is emitted twice
    rng = np.random.default_rng(77)
    x = random_params(rng, F, pose_sigma=0.2)
    prob = api.Problem(gpu_model, seq.kp_offset, seq.kp_id, seq.kp_uv, seq.intr, seq.R0, n_cols=76, use_shape=False,
                       beta_pose=20.0)
    r, J, _ = prob.evaluate(x, None, True)
    ro, Jo = omodel.evaluate_batch(seq, x, np.zeros(10), 76, False, True, mode=0)
    K2 = prob.layout.reproj_rows
    assert np.abs(r[:K2] - ro).max() < 1e-9 and np.abs(J - Jo).max() < 1e-9 * np.abs(Jo).max()
    # fit (pose + Sim3, L2 prior 20, joints 10/11/22/23 constant) from the reference's initial state
    from oracle import lm_dense
    x0 = np.zeros((F, 76)); x0[:, 0] = 1.0; x0[:, 6] = 3.0
    const = np.zeros(76, np.uint8)
    for j in (10, 11, 22, 23):
        const[7 + 3 * (j - 1):10 + 3 * (j - 1)] = 1
    xf, _, summ = prob.solve(x0, None, constant=const, independent=True, max_iters=100)
    assert all(s.usable for s in summ)
    for f in [4, 20]:   # two non-empty frames against the dense LM
        k0, k1 = seq.kp_offset[f], seq.kp_offset[f + 1]
        assert k1 > k0
        s1 = S(); s1.kp_offset = np.array([0, k1 - k0], np.int32); s1.kp_id = seq.kp_id[k0:k1]
        s1.kp_uv = seq.kp_uv[k0:k1]; s1.intr = seq.intr; s1.R0 = seq.R0[f:f + 1]
        xo, _, info = lm_dense.solve(omodel, s1, x0[f:f + 1], None, n_cols=76, use_shape=False, beta_pose=20.0,
                                     max_iters=100, constant=const)
        d_rot = np.abs(np.delete(xf[f], [0, 4, 5, 6]) - np.delete(xo[0], [0, 4, 5, 6])).max()
        d_t = np.abs(xf[f, 4:7] / xf[f, 0] - xo[0, 4:7] / xo[0, 0]).max()
        assert max(d_rot, d_t) < 1e-4
        assert abs(summ[f].final_cost - info["final_cost"]) < 1e-5 * max(info["final_cost"], 1e-9)
    # an empty frame ('[]' JSON) keeps its parameters up to the prior pull and reports a usable solution
    e = int(np.where(np.diff(seq.kp_offset) == 0)[0][0])
    assert summ[e].usable


def test_many_keypoints_per_frame_and_no_posedirs(api, synth, model, gpu_model, oracle_mod, omodel):
    """(1) More than 32 keypoints in a frame exercises the keypoint-chunk loop of k_frame_resjac (and sends the
    solve to the host loop: the device LM stages at most 32 keypoints).  (2) A model uploaded without posedirs
    (upstream avatar applies no pose-corrective blendshapes) equals pose_blend = 0."""
    F = 3
    base = synth.make_sequence(model, F, seed=12)
    rng = np.random.default_rng(5)
    ids, uvs, offs = [], [], [0]
    for f in range(F):
        k0, k1 = base.kp_offset[f], base.kp_offset[f + 1]
        rep = 3 if f == 1 else 1                      # frame 1 gets 75 keypoints (duplicates are legal: quirk Q1)
        ids.append(np.tile(base.kp_id[k0:k1], rep))
        uvs.append(np.tile(base.kp_uv[k0:k1], (rep, 1)) + rng.normal(scale=0.5, size=(rep * (k1 - k0), 2)))
        offs.append(offs[-1] + rep * (k1 - k0))
    class S: pass
    seq = S(); seq.kp_offset = np.array(offs, np.int32); seq.kp_id = np.concatenate(ids).astype(np.int32)
    seq.kp_uv = np.concatenate(uvs); seq.intr = base.intr; seq.R0 = base.R0
    x = random_params(rng, F); beta = rng.normal(size=10)
    prob = api.Problem(gpu_model, seq.kp_offset, seq.kp_id, seq.kp_uv, seq.intr, seq.R0, n_cols=86, use_shape=True)
    r, J, _ = prob.evaluate(x, beta, True)
    ro, Jo = omodel.evaluate_batch(seq, x, beta, 86, True, True, mode=0)
    assert np.abs(r - ro).max() < 1e-9 and np.abs(J - Jo).max() < 1e-9 * np.abs(Jo).max()
    pf = api.Problem(gpu_model, seq.kp_offset, seq.kp_id, seq.kp_uv, seq.intr, seq.R0, n_cols=86, use_shape=True,
                     beta_per_frame=True, beta_pose=5.0, beta_shape=10.0)
    with pytest.raises(api.BodyfitError):
        pf.solve(base.init_params, np.zeros((F, 10)), independent=True, max_iters=3, solver=2)   # device loop refuses
    xs, bs, summ = pf.solve(base.init_params, np.zeros((F, 10)), independent=True, max_iters=5)  # auto -> host loop
    assert all(s.usable for s in summ) and summ[1].final_cost < summ[1].initial_cost
    # (2) model without pose-corrective blendshapes
    gm0 = api.Model(model, device=0, pose_blend_data=False)
    p0 = api.Problem(gm0, base.kp_offset, base.kp_id, base.kp_uv, base.intr, base.R0, n_cols=86, use_shape=True,
                     want_mesh=True)
    r0, J0, _ = p0.evaluate(x, beta, True)
    r1, J1 = omodel.evaluate_batch(base, x, beta, 86, True, False, mode=0)
    assert np.abs(r0 - r1).max() < 1e-9 and np.abs(J0 - J1).max() < 1e-9 * np.abs(J1).max()
    _, cloud = p0.forward(x, beta)
    _, co = omodel.forward(x[1], beta, base.R0[1], pose_blend=False)
    assert np.abs(cloud[1] - co).max() < 5e-6


@pytest.mark.gpu
def test_writeback_batch_matches_host_composition(api, synth, model, gpu_model, omodel, oracle_mod):
    """bodyfit_writeback_batch against the same steps done by the checker: R0' = R(rootAA) R0, forward with
    s = 1 / zero root angle-axis through the oracle, the ORACLE's mean_pixel_error (include/Utils.h:102-115 restated in
    oracle/bodyfit_oracle.cpp) over the FK keypoints; frames without keypoints -> 0.  The product's own host function
    bodyfit_mean_pixel_error is checked against the oracle on the same inputs as well."""
    F = 9
    seq = synth.make_sequence(model, F, seed=21, ragged=True)
    rng = np.random.default_rng(5)
    x = random_params(rng, F, pose_sigma=0.3)
    beta = rng.normal(size=10) * 0.5
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, want_mesh=True)
    wb = prob.writeback(x, beta, want_cloud=True)
    for f in range(F):
        R0n = synth.rodrigues(x[f, 1:4]) @ seq.R0[f].reshape(3, 3)
        assert np.abs(wb["R0"][f] - R0n).max() < 1e-12
        xu = x[f].copy(); xu[0] = 1.0; xu[1:4] = 0.0
        jo, co = omodel.forward(xu, beta, R0n)
        assert np.abs(wb["joints"][f] - jo).max() < 1e-9
        assert np.abs(wb["cloud"][f] - co).max() < 5e-6
        k0, k1 = seq.kp_offset[f], seq.kp_offset[f + 1]
        ids, uv = seq.kp_id[k0:k1], seq.kp_uv[k0:k1]
        fk = ids < 24
        want = oracle_mod.mean_pixel_error(ids[fk], uv[fk], jo, seq.intr) if fk.any() else 0.0
        assert abs(wb["mean_px"][f] - want) < 1e-8 * max(1.0, want)
        if fk.any():   # the C-ABI host function (mean_pixel_error binding) against the checker too
            assert abs(api.mean_pixel_error(ids[fk], uv[fk], jo, seq.intr) - want) < 1e-10 * max(1.0, want)


@pytest.mark.gpu
def test_large_batch_equals_small_batches(api, synth, model, gpu_model):
    """Size-independent property at a full-size batch (1024 frames, two workgroups per CU): every frame's residuals and
    Jacobian are bit-identical to the same frame evaluated in a 40-frame problem."""
    F = 1024
    seq = synth.make_sequence(model, F, seed=5)
    rng = np.random.default_rng(9)
    x = random_params(rng, F, pose_sigma=0.2)
    beta = rng.normal(size=(F, 10)) * 0.5
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0,
                                     beta_shape=30.0)
    r, J, _ = prob.evaluate(x, beta, True)
    K2 = prob.layout.reproj_rows
    for f0 in (0, 492, F - 40):
        sl = slice(f0, f0 + 40)
        k0, k1 = seq.kp_offset[f0], seq.kp_offset[f0 + 40]
        sub = api.Problem(gpu_model, seq.kp_offset[f0:f0 + 41] - k0, seq.kp_id[k0:k1], seq.kp_uv[k0:k1], seq.intr, seq.R0[sl],
                          n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0, beta_shape=30.0)
        rs, Js, _ = sub.evaluate(x[sl], beta[sl], True)
        assert np.array_equal(rs[:2 * (k1 - k0)], r[2 * k0:2 * k1])
        assert np.array_equal(Js, J[2 * k0:2 * k1])
        npr = prob.layout.prior_rows_per_frame
        assert np.array_equal(rs[2 * (k1 - k0):2 * (k1 - k0) + 40 * npr], r[K2 + f0 * npr:K2 + (f0 + 40) * npr])


@pytest.mark.gpu
def test_small_model_variants(api, synth, oracle_mod):
    """Generality of the kernels beyond the SMPL sizes the bench uses: 6 shape directions, a 2048-vertex mesh (64 full
    tiles), no landmarks, a 3-component GMM; residuals, Jacobian and mesh against the oracle."""
    import dataclasses
    base = synth.make_model(3, n_verts=2048, n_landmarks=0)
    m = dataclasses.replace(base, shapedirs=base.shapedirs[:, :, :6].copy()).finalize()
    assert m.n_shape == 6 and len(m.landmark_vid) == 0
    gm = api.Model(m)
    om = oracle_mod.OracleModel(m)
    F = 20
    seq = synth.make_sequence(m, F, seed=8, kp_ids=[i for i in synth.BODY25_IDS if i < 24])
    rng = np.random.default_rng(12)
    x = random_params(rng, F, pose_sigma=0.25)
    beta = rng.normal(size=6) * 0.7
    w, mu, cov = synth.make_gmm(5, n_comp=3)
    gmm = api.Gmm(w, mu, cov)
    prob = api.Problem.from_sequence(gm, seq, n_cols=76 + 6, use_shape=True, beta_pose=3.0, gmm=gmm, beta_shape=2.0,
                                     lambda_temporal=1.5, want_mesh=True)
    r, J, comp = prob.evaluate(x, beta, True)
    ro, Jo = om.evaluate_batch(seq, x, beta, 82, True, True, mode=0)
    K2 = prob.layout.reproj_rows
    assert np.abs(r[:K2] - ro).max() < 1e-9 and np.abs(J - Jo).max() < 1e-9 * max(1.0, np.abs(Jo).max())
    og = oracle_mod.OracleGmm(w, mu, cov)
    for f in (0, F - 1):
        rp, kc = og.residual(x[f, 7:])
        assert kc == comp[f]
        assert np.abs(r[K2 + f * 70:K2 + (f + 1) * 70] - 3.0 * rp).max() < 1e-9
    joints, cloud = prob.forward(x, beta)
    for f in (0, 7, F - 1):
        jo, co = om.forward(x[f], beta, seq.R0[f])
        assert np.abs(joints[f] - jo).max() < 1e-11
        assert np.abs(cloud[f] - co).max() < 5e-6


@pytest.mark.gpu
def test_sweeps_are_deterministic(api, synth, model, gpu_model):
    """Race / ordering check: 40 evaluations of the same inputs give bit-identical residuals, Jacobian, mesh and
    shared-beta reduction (fixed-order sums everywhere, no atomics on the data path)."""
    import torch
    F = 96
    seq = synth.make_sequence(model, F, seed=13)
    rng = np.random.default_rng(3)
    x = random_params(rng, F, pose_sigma=0.3)
    beta = rng.normal(size=10) * 0.5
    w, mu, cov = synth.make_gmm(0)
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_pose=4.0, gmm=api.Gmm(w, mu, cov),
                                     beta_shape=20.0, lambda_temporal=2.0, want_mesh=True)
    dx = torch.from_numpy(x).cuda(); db = torch.from_numpy(beta).cuda()
    out = torch.zeros(66, dtype=torch.float64, device="cuda")
    def snapshot():
        prob.evaluate_device(dx.data_ptr(), db.data_ptr(), True, None)
        prob.reduce_shared_device(out.data_ptr(), None)
        torch.cuda.synchronize()
        r, J, comp = prob.evaluate(x, beta, True)          # same point through the host-pointer path (copies r, J back)
        joints, cloud = prob.forward(x, beta)
        return r.copy(), J.copy(), comp.copy(), cloud.copy(), out.cpu().numpy().copy()

    ref = snapshot()
    for _ in range(40):
        cur = snapshot()
        for a, b in zip(ref, cur):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_model_on_a_second_device(api, synth, model, oracle_mod):
    """A model created on a device other than the first one used: every kernel's dynamic-LDS grant is per device
    (csrc/device_once.h).  Needs two GPUs (skipped on a one-GPU box; the bookkeeping itself is tested on the CPU in test_abi)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU")
    gm1 = api.Model(model, device=1)
    seq = synth.make_sequence(model, 5, seed=77)
    w, mu, cov = synth.make_gmm(0)
    prob = api.Problem.from_sequence(gm1, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0,
                                     gmm=api.Gmm(w, mu, cov, device=1), beta_shape=30.0, want_mesh=True)
    x = seq.gt_params + 0.01
    b = np.tile(seq.gt_beta, (5, 1))
    r, J, _ = prob.evaluate(x, b, True)
    om = oracle_mod.OracleModel(model)
    ro, Jo = om.evaluate_batch(seq, x, b[0], 86, True, True)
    K = len(ro)
    assert np.abs(r[:K] - ro).max() < 1e-9
    xs, bs, ss = prob.solve(seq.init_params, np.zeros((5, 10)), independent=True, max_iters=20)   # the LM kernels' grants too
    assert all(s.final_cost <= s.initial_cost for s in ss)


@pytest.mark.gpu
def test_packed_cache_serves_every_block_of_the_dense_panel(api, synth, model, gpu_model):
    """bodyfit_evaluate_batch without a caller's Jacobian buffer ships only the column blocks a probe sweep found non-zero
    (k_pack_jacobian); bodyfit_evaluate_block must hand out, for EVERY keypoint (FK joints of every chain depth, vertex
    landmarks) and every parameter block, exactly the numbers of the dense panel, zeros included."""
    F = 6
    seq = synth.make_sequence(model, F, seed=16)
    assert (seq.kp_id >= 24).any() and (seq.kp_id < 24).any()
    rng = np.random.default_rng(5)
    x = random_params(rng, F); beta = rng.normal(size=10)
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0,
                                     lambda_temporal=3.0)
    r, J, _ = prob.evaluate(x, beta, True)              # the dense panel
    prob.cache_sweep(x, beta)                           # the packed cache
    zero_blocks = 0
    for f in range(F):
        blocks = [x[f, 0:1], x[f, 1:4], x[f, 4:7]] + [x[f, 7 + 3 * j:10 + 3 * j] for j in range(23)] + [beta]
        for k in range(int(seq.kp_offset[f]), int(seq.kp_offset[f + 1])):
            rb, jacs = prob.evaluate_block(0, k, blocks, 2)
            assert np.array_equal(rb, r[2 * k:2 * k + 2])
            Jcat = np.concatenate(jacs, axis=1)
            assert np.array_equal(Jcat, J[2 * k:2 * k + 2]), (f, k, int(seq.kp_id[k]))
            zero_blocks += sum(1 for j in jacs if not j.any())
    assert zero_blocks > 0   # (the packing has something to leave out: non-ancestor joints of the FK keypoints)


def test_a_problem_built_from_pooled_blocks_equals_one_built_from_fresh_memory(api, synth, model, gpu_model, oracle_mod, omodel):
    """Device blocks of a destroyed problem are kept for the next problem of the same shape (BlockPool, bodyfit_api.hip): the
    staged drivers create and destroy two problems per stage.  Stale contents are then the NORMAL case for every buffer the
    creation does not clear (Jacobian, partials, cloud, write-back).  Problem A (one sequence, one set of prior weights) is
    evaluated, solved and destroyed; problem B — same sizes, other keypoints, other weights — takes A's blocks, and must agree bit
    for bit with B2, built while B is alive (the pool has nothing of those sizes left: fresh hipMalloc), and with the oracle."""
    F = 24
    seq_a = synth.make_sequence(model, F, seed=11)
    seq_b = synth.make_sequence(model, F, seed=12)
    kw_a = dict(n_cols=86, use_shape=True, beta_pose=9.0, beta_shape=40.0, lambda_temporal=7.0, want_mesh=True)
    kw_b = dict(n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0, want_mesh=True)
    rng = np.random.default_rng(5)
    xa, xb = random_params(rng, F), random_params(rng, F)
    ba, bb = rng.normal(size=10), rng.normal(size=10)
    pa = api.Problem.from_sequence(gpu_model, seq_a, **kw_a)
    pa.evaluate(xa, ba, True); pa.forward(xa, ba); pa.writeback(xa, ba, want_cloud=True)
    pa.solve(seq_a.init_params, np.zeros(10), independent=False, max_iters=3, scale_bounds=(-1e300, 1e300), solver=3)
    pa.close()
    pb = api.Problem.from_sequence(gpu_model, seq_b, **kw_b)          # takes A's blocks
    pb2 = api.Problem.from_sequence(gpu_model, seq_b, **kw_b)         # fresh memory
    rb, Jb, _ = pb.evaluate(xb, bb, True)
    r2, J2, _ = pb2.evaluate(xb, bb, True)
    assert np.array_equal(rb, r2) and np.array_equal(Jb, J2)
    jb, cb = pb.forward(xb, bb); j2, c2 = pb2.forward(xb, bb)
    assert np.array_equal(jb, j2) and np.array_equal(cb, c2)
    wb, w2 = pb.writeback(xb, bb, want_cloud=True), pb2.writeback(xb, bb, want_cloud=True)
    for key in ("R0", "joints", "cloud", "mean_px"):
        assert np.array_equal(wb[key], w2[key]), key
    sb = pb.solve(seq_b.init_params, np.zeros(10), independent=False, max_iters=6, scale_bounds=(-1e300, 1e300), solver=3)
    s2 = pb2.solve(seq_b.init_params, np.zeros(10), independent=False, max_iters=6, scale_bounds=(-1e300, 1e300), solver=3)
    assert np.array_equal(sb[0], s2[0]) and np.array_equal(sb[1], s2[1]) and sb[2][0].final_cost == s2[2][0].final_cost
    ro, Jo, _ = _oracle_full(oracle_mod, omodel, seq_b, xb, bb, 86, True, True, 5.0, None, 25.0, 3.0)
    assert np.abs(rb - ro).max() < 1e-9 and np.abs(Jb - Jo).max() <= 1e-9 * max(1.0, np.abs(Jo).max())
