"""GPU: fitted parameters of the HIP evaluator under the product's LM (bodyfit_solve) against the oracle
evaluator under the independent dense numpy LM, on identical keypoint inputs.  north_star tolerance:
parameters within 1e-4 (rad / m / unitless)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def gauge_free_diff(x, xo):
    """The reference's camera model X = s R q + t projects identically for (s, t) -> (c s, c t): the Sim3
    scale is an exact null direction of every reprojection residual (include/Sim3BA.h:216-225), pinned only
    by LM damping and rounding.  Parameters are therefore compared modulo that gauge: rotations and joint
    angles directly, the translation as t / s; the scale itself gets its own (looser) check."""
    x = np.atleast_2d(x); xo = np.atleast_2d(xo)
    d_rot = np.abs(np.delete(x, [0, 4, 5, 6], axis=1) - np.delete(xo, [0, 4, 5, 6], axis=1)).max()
    d_t = np.abs(x[:, 4:7] / x[:, :1] - xo[:, 4:7] / xo[:, :1]).max()
    in_bounds = bool(np.all((x[:, 0] >= 0.3 - 1e-12) & (x[:, 0] <= 3.0 + 1e-12)))
    return max(d_rot, d_t), in_bounds


def _lm(oracle_mod):
    from oracle import lm_dense
    return lm_dense


def test_c1_c2_single_frame_pose_only(api, synth, model, gpu_model, oracle_mod, omodel):
    """3dba_single without --opt-shape: ReprojCost blocks (76 columns), L2 pose prior, joints 10/11/22/23
    held constant (include/Sim3BA.h:608-611), 25 keypoints, reference initial state."""
    seq = synth.make_sequence(model, 1, seed=0, beta_fixed=True)
    const = np.zeros(76, np.uint8)
    for j in (10, 11, 22, 23):
        const[7 + 3 * (j - 1):10 + 3 * (j - 1)] = 1
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=76, use_shape=False, beta_pose=20.0)
    x, _, summ = prob.solve(seq.init_params, None, constant=const, independent=True, max_iters=100)
    xo, _, info = _lm(oracle_mod).solve(omodel, seq, seq.init_params, None, n_cols=76, use_shape=False, beta_pose=20.0,
                                        max_iters=100, constant=const)
    assert summ[0].usable and summ[0].termination == 0 and info["termination"] == 0
    assert abs(summ[0].final_cost - info["final_cost"]) < 1e-6 * info["final_cost"]
    d, ok_s = gauge_free_diff(x, xo)
    assert d < TOL and ok_s
    assert np.all(x[0, 7 + 27:7 + 33] == 0) and np.all(x[0, 7 + 63:] == 0)   # constant blocks untouched
    assert summ[0].final_cost < 0.05 * summ[0].initial_cost


def test_c3_batched_independent_frames_shape_and_gmm(api, synth, model, gpu_model, oracle_mod, omodel):
    """--opt-shape --use-gmm, frames fitted independently in one batched solve (own LM state per frame)."""
    F = 6
    seq = synth.make_sequence(model, F, seed=1)
    w, mu, cov = synth.make_gmm(0)
    gmm = api.Gmm(w, mu, cov); ogmm = oracle_mod.OracleGmm(w, mu, cov)
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0,
                                     gmm=gmm, beta_shape=30.0)
    x, b, summ = prob.solve(seq.init_params, np.zeros((F, 10)), independent=True, max_iters=60)
    lm = _lm(oracle_mod)
    for f in [0, 3, 5]:
        class S: pass
        s = S(); k0, k1 = seq.kp_offset[f], seq.kp_offset[f + 1]
        s.kp_offset = np.array([0, k1 - k0], np.int32); s.kp_id = seq.kp_id[k0:k1]; s.kp_uv = seq.kp_uv[k0:k1]
        s.intr = seq.intr; s.R0 = seq.R0[f:f + 1]
        xo, bo, info = lm.solve(omodel, s, seq.init_params[f:f + 1], np.zeros(10), n_cols=86, use_shape=True,
                                beta_pose=20.0, ogmm=ogmm, beta_shape=30.0, max_iters=60)
        assert summ[f].termination == 0 and info["termination"] == 0
        assert abs(summ[f].final_cost - info["final_cost"]) < 1e-5 * info["final_cost"]
        d, ok_s = gauge_free_diff(x[f], xo[0])
        assert d < TOL and ok_s and np.abs(b[f] - bo).max() < TOL


@pytest.mark.parametrize("F,shape", [(1, False), (40, True), (256, True)])
def test_speculative_iteration_equals_four_launch_iteration(api, synth, model, gpu_model, monkeypatch, F, shape):
    """The batched device LM judges a candidate in the prologue of the next k_lm_step and takes the Jacobian from the
    candidate sweep (two launches per iteration); BODYFIT_LM_PLAIN=1 keeps step / residual sweep / accept / Jacobian sweep.
    Same arithmetic in another launch order: same decisions, same iteration counts, same iterates."""
    seq = synth.make_sequence(model, F, seed=4, beta_fixed=not shape)
    w, mu, cov = synth.make_gmm(0)

    def fit(plain):
        monkeypatch.setenv("BODYFIT_LM_PLAIN", "1" if plain else "0")
        if shape:
            prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0,
                                             gmm=api.Gmm(w, mu, cov), beta_shape=30.0)
            return prob.solve(seq.init_params, np.zeros((F, 10)), independent=True, max_iters=100)
        prob = api.Problem.from_sequence(gpu_model, seq, n_cols=76, use_shape=False, beta_pose=20.0)
        return prob.solve(seq.init_params, None, independent=True, max_iters=100)

    xs, bs, ss = fit(False)
    xp, bp, sp = fit(True)
    # same arithmetic in the same order (the cost reduction of the prologue sums like k_lm_accept): identical decisions
    for f in range(F):
        a, b = ss[f], sp[f]
        assert (a.iterations, a.n_successful, a.n_unsuccessful, a.termination) == \
               (b.iterations, b.n_successful, b.n_unsuccessful, b.termination), f
        assert a.final_cost == b.final_cost
    assert np.array_equal(xs, xp)
    if shape:
        assert np.array_equal(bs, bp)
    assert ss[0].n_sweeps < sp[0].n_sweeps   # one sweep per iteration instead of two


def test_c4_multi_frame_window_shared_beta(api, synth, model, gpu_model, oracle_mod, omodel):
    """OptimizeMultiFrame on one 20-frame window: shared beta, L2 pose prior 5, shape prior 25, temporal 3."""
    F = 20
    seq = synth.make_sequence(model, F, seed=2)
    kw = dict(n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0)
    prob = api.Problem.from_sequence(gpu_model, seq, lambda_temporal=3.0, **kw)
    x, b, summ = prob.solve(seq.init_params, np.zeros(10), independent=False, max_iters=40)
    xo, bo, info = _lm(oracle_mod).solve(omodel, seq, seq.init_params, np.zeros(10), lam=3.0, max_iters=40, **kw)
    assert abs(summ[0].final_cost - info["final_cost"]) < 1e-5 * info["final_cost"]
    d, _ = gauge_free_diff(x, xo)   # the multi-frame problem sets no bounds on the scale
    assert d < TOL and np.abs(b - bo).max() < TOL
    # the fit explains the observations: mean reprojection error near the 1 px noise floor
    r, _, _ = prob.evaluate(x, b, False)
    K = prob.layout.n_keypoints
    assert np.sqrt((r[:2 * K].reshape(K, 2) ** 2).sum(1)).mean() < 2.5


def test_noise_free_recovery_regression(api, synth, model, gpu_model):
    """End-to-end regression: noise-free observations from known parameters are explained to
    sub-pixel residuals."""
    seq = synth.make_sequence(model, 4, seed=3, noise_px=0.0, pose_sigma=0.15)
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=1e-3,
                                     beta_shape=1e-3)
    x, b, summ = prob.solve(seq.init_params, np.zeros((4, 10)), independent=True, max_iters=200)
    r, _, _ = prob.evaluate(x, b, False)
    K = prob.layout.n_keypoints
    assert np.abs(r[:2 * K]).max() < 0.3
    assert all(s.usable for s in summ)


@pytest.mark.gpu
def test_window_solve_device_normals_match_host_normals(api, synth, model, gpu_model, monkeypatch):
    """The window LM with the reprojection normal-equation panels built on the device (k_frame_normal) against the
    same solve forming them on the host from the copied-back Jacobian (BODYFIT_HOST_NORMALS): same iterates up to
    summation order."""
    F = 12
    seq = synth.make_sequence(model, F, seed=31)
    kw = dict(n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)
    prob = api.Problem.from_sequence(gpu_model, seq, **kw)
    x1, b1, s1 = prob.solve(seq.init_params, np.zeros(10), independent=False, max_iters=25, scale_bounds=(-1e300, 1e300))
    monkeypatch.setenv("BODYFIT_HOST_NORMALS", "1")
    prob2 = api.Problem.from_sequence(gpu_model, seq, **kw)
    x2, b2, s2 = prob2.solve(seq.init_params, np.zeros(10), independent=False, max_iters=25, scale_bounds=(-1e300, 1e300))
    assert s1[0].iterations == s2[0].iterations
    assert abs(s1[0].final_cost - s2[0].final_cost) < 1e-9 * s2[0].final_cost
    assert np.abs(x1[:, 1:] - x2[:, 1:]).max() < 1e-6 and np.abs(b1 - b2).max() < 1e-6


def test_c3_at_size_256_frames_sampled_against_dense_lm(api, synth, model, gpu_model, oracle_mod, omodel):
    """BASELINE configs[2] at its real size: 256 independent frames, --opt-shape, GMM prior, ONE batched device solve (pooled
    LM state, the batch runs as long as its slowest frame).  Eight frames, the slowest-converging one among them, are refitted
    by the dense numpy LM over the oracle evaluator and compared (include/Sim3BA.h:348-511)."""
    F = 256
    seq = synth.make_sequence(model, F, seed=1)
    w, mu, cov = synth.make_gmm(0)
    gmm = api.Gmm(w, mu, cov); ogmm = oracle_mod.OracleGmm(w, mu, cov)
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0,
                                     gmm=gmm, beta_shape=30.0)
    x, b, summ = prob.solve(seq.init_params, np.zeros((F, 10)), independent=True, max_iters=100)
    its = np.array([q.iterations for q in summ])
    assert sum(q.termination == 0 for q in summ) >= 250 and all(q.usable for q in summ)
    slow = int(np.argmax(its))
    sample = sorted(set([slow, 0, 37, 64, 101, 150, 203, 255]))
    lm = _lm(oracle_mod)
    for f in sample:
        class S: pass
        s = S(); k0, k1 = seq.kp_offset[f], seq.kp_offset[f + 1]
        s.kp_offset = np.array([0, k1 - k0], np.int32); s.kp_id = seq.kp_id[k0:k1]; s.kp_uv = seq.kp_uv[k0:k1]
        s.intr = seq.intr; s.R0 = seq.R0[f:f + 1]
        xo, bo, info = lm.solve(omodel, s, seq.init_params[f:f + 1], np.zeros(10), n_cols=86, use_shape=True,
                                beta_pose=20.0, ogmm=ogmm, beta_shape=30.0, max_iters=100)
        if summ[f].termination != 0 or info["termination"] != 0:
            continue   # (a frame that ran into the iteration limit on either side is not a parity case)
        assert abs(summ[f].final_cost - info["final_cost"]) < 1e-5 * info["final_cost"], f
        d, ok_s = gauge_free_diff(x[f], xo[0])
        assert d < TOL and ok_s and np.abs(b[f] - bo).max() < TOL, (f, d)
    assert summ[slow].termination == 0 or its[slow] == 100


def test_c5_stage1_103_anchors_against_oracle_evaluator(api, synth, model, gpu_model, oracle_mod, omodel):
    """BASELINE configs[4], stage 1 at its real size: 1024 frames, anchor_skip 10 -> 103 anchors, one shared beta
    (src/main_multi_frame.cpp:109-134).  The fit runs on the device (window LM); at the fitted point the oracle evaluator
    reproduces the residuals and Jacobian rows of sampled anchors, and the cost has dropped to the noise floor."""
    seq = synth.make_sequence(model, 1024, seed=3)
    ids = list(range(0, 1024, 10))
    class S: pass
    s = S(); offs = [0]; kid = []; uv = []
    for f in ids:
        k0, k1 = seq.kp_offset[f], seq.kp_offset[f + 1]
        kid.append(seq.kp_id[k0:k1]); uv.append(seq.kp_uv[k0:k1]); offs.append(offs[-1] + k1 - k0)
    s.kp_offset = np.array(offs, np.int32); s.kp_id = np.concatenate(kid); s.kp_uv = np.concatenate(uv)
    s.intr = seq.intr; s.R0 = seq.R0[ids]
    prob = api.Problem.from_sequence(gpu_model, s, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)
    x, b, summ = prob.solve(seq.init_params[ids], np.zeros(10), independent=False, max_iters=200, scale_bounds=(-1e300, 1e300))
    assert summ[0].usable and summ[0].final_cost < 0.02 * summ[0].initial_cost
    r, J, _ = prob.evaluate(x, b, True)
    ro, Jo = omodel.evaluate_batch(s, x, b, 86, True, True, mode=0)
    K2 = prob.layout.reproj_rows
    assert np.abs(r[:K2] - ro).max() < 1e-9
    for a in (0, 17, 51, 102):
        k0, k1 = 2 * s.kp_offset[a], 2 * s.kp_offset[a + 1]
        assert np.abs(J[k0:k1] - Jo[k0:k1]).max() <= 1e-9 * max(1.0, np.abs(Jo[k0:k1]).max())
    K = prob.layout.n_keypoints
    assert np.sqrt((r[:K2].reshape(K, 2) ** 2).sum(1)).mean() < 2.5


def test_c5_stage1_103_anchors_against_the_checkers_lm(api, synth, model, gpu_model, oracle_mod, omodel):
    """BASELINE configs[4], stage 1 at its real size (103 anchors, 7,838 unknowns, src/main_multi_frame.cpp:109-134; solver
    configuration include/MultiFrameBA.h:144-151): the FIT of the device window LM against the checker's LM — the oracle
    evaluator under oracle/lm_dense.py in its scipy.sparse form (same rows and rules as the dense form: tests/test_oracle.py) —
    iterate for iterate over 30 iterations from the reference's initial state."""
    seq = synth.make_sequence(model, 1024, seed=3)
    ids = list(range(0, 1024, 10))
    class S: pass
    s = S(); offs = [0]; kid = []; uv = []
    for f in ids:
        k0, k1 = seq.kp_offset[f], seq.kp_offset[f + 1]
        kid.append(seq.kp_id[k0:k1]); uv.append(seq.kp_uv[k0:k1]); offs.append(offs[-1] + k1 - k0)
    s.kp_offset = np.array(offs, np.int32); s.kp_id = np.concatenate(kid); s.kp_uv = np.concatenate(uv)
    s.intr = seq.intr; s.R0 = seq.R0[ids]
    assert len(ids) == 103
    prob = api.Problem.from_sequence(gpu_model, s, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)
    for iters in (8, 30):
        x, b, summ = prob.solve(seq.init_params[ids], np.zeros(10), independent=False, max_iters=iters, scale_bounds=(-1e300, 1e300),
                                solver=3)
        xo, bo, info = _lm(oracle_mod).solve(omodel, s, seq.init_params[ids], np.zeros(10), n_cols=86, use_shape=True, beta_pose=5.0,
                                            beta_shape=25.0, lam=3.0, max_iters=iters, scale_bounds=(-1e300, 1e300), sparse=True)
        assert (summ[0].iterations, summ[0].n_successful) == (info["iterations"], info["n_ok"])
        assert abs(summ[0].final_cost - info["final_cost"]) < 1e-8 * info["final_cost"]
        assert np.abs(x - xo).max() < 1e-6 and np.abs(b - bo).max() < 1e-6


@pytest.mark.parametrize("F", [545, 1024])
def test_one_long_window_against_the_checkers_lm(api, synth, model, gpu_model, oracle_mod, omodel, F):
    """BASELINE configs[4] as ONE window (what `bench.py --workload c5 --fit` and the north star's strong-scaling figure are
    quoted on; include/MultiFrameBA.h:33-177 with every frame of the sequence, solver configuration :144-151): the device
    window LM (block cyclic reduction over 10 / 11 levels + beta Schur) against the checker's LM — oracle evaluator under
    oracle/lm_dense.py in scipy.sparse form — iterate for iterate from the reference's initial state
    (src/main_multi_frame.cpp:96-100) over 8 iterations, which contain accepted steps and a rejected one."""
    seq = synth.make_sequence(model, F, seed=3)
    prob = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0,
                                     lambda_temporal=3.0)
    kw = dict(max_iters=8, scale_bounds=(-1e300, 1e300))
    x, b, summ = prob.solve(seq.init_params, np.zeros(10), independent=False, **kw)
    xo, bo, info = _lm(oracle_mod).solve(omodel, seq, seq.init_params, np.zeros(10), n_cols=86, use_shape=True, beta_pose=5.0,
                                        beta_shape=25.0, lam=3.0, sparse=True, **kw)
    assert info["n_bad"] >= 1 and info["n_ok"] >= 5
    assert (summ[0].iterations, summ[0].n_successful) == (info["iterations"], info["n_ok"])
    assert abs(summ[0].initial_cost - info["initial_cost"]) < 1e-9 * info["initial_cost"]
    assert abs(summ[0].final_cost - info["final_cost"]) < 1e-8 * info["final_cost"]
    assert np.abs(x - xo).max() < 1e-6 and np.abs(b - bo).max() < 1e-6
