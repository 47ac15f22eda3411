"""The device-resident window LM (k_window_lm.hip: block cyclic reduction over the frames, beta as a Schur complement,
Ceres' trust-region logic in small kernels) against the host loop of host_solver.cpp (sequential block-tridiagonal
Cholesky) on the same problems, and against the dense numpy LM over the oracle evaluator.  Both product solvers restate
the same algorithm (include/MultiFrameBA.h:144-151 hands it to ceres::Solve), so the iterates agree up to rounding."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0, lambda_temporal=3.0)


def _both(api, gpu_model, seq, F, max_iters, constant=None, bounds=(-1e300, 1e300), **kw):
    args = dict(KW); args.update(kw)
    ph = api.Problem.from_sequence(gpu_model, seq, **args)
    pd = api.Problem.from_sequence(gpu_model, seq, **args)
    xh, bh, sh = ph.solve(seq.init_params, np.zeros(10), constant=constant, independent=False, max_iters=max_iters,
                          scale_bounds=bounds, solver=1)
    xd, bd, sd = pd.solve(seq.init_params, np.zeros(10), constant=constant, independent=False, max_iters=max_iters,
                          scale_bounds=bounds, solver=3)
    return (xh, bh, sh[0]), (xd, bd, sd[0])


@pytest.mark.parametrize("F", [1, 2, 3, 5, 8, 13, 20, 33])
def test_device_window_lm_matches_host_loop(api, synth, model, gpu_model, F):
    seq = synth.make_sequence(model, F, seed=40 + F)
    (xh, bh, sh), (xd, bd, sd) = _both(api, gpu_model, seq, F, 12)
    # same iterates: same number of accepted / rejected steps, same costs, same parameters (scale excepted: the gauge)
    assert (sd.iterations, sd.n_successful, sd.n_unsuccessful) == (sh.iterations, sh.n_successful, sh.n_unsuccessful)
    assert abs(sd.initial_cost - sh.initial_cost) <= 1e-12 * sh.initial_cost
    assert abs(sd.final_cost - sh.final_cost) <= 1e-9 * sh.final_cost
    assert np.abs(xd[:, 1:] - xh[:, 1:]).max() < 1e-7 and np.abs(bd - bh).max() < 1e-7


def test_device_window_lm_to_convergence_and_dense_lm(api, synth, model, gpu_model, oracle_mod, omodel):
    from oracle import lm_dense
    F = 20
    seq = synth.make_sequence(model, F, seed=2)
    (xh, bh, sh), (xd, bd, sd) = _both(api, gpu_model, seq, F, 40)
    assert abs(sd.final_cost - sh.final_cost) <= 1e-8 * sh.final_cost
    kw = dict(n_cols=86, use_shape=True, beta_pose=5.0, beta_shape=25.0)
    xo, bo, info = lm_dense.solve(omodel, seq, seq.init_params, np.zeros(10), lam=3.0, max_iters=40,
                                  scale_bounds=(-1e300, 1e300), **kw)
    assert abs(sd.final_cost - info["final_cost"]) < 1e-5 * info["final_cost"]
    d_rot = np.abs(np.delete(xd, [0, 4, 5, 6], axis=1) - np.delete(xo, [0, 4, 5, 6], axis=1)).max()
    d_t = np.abs(xd[:, 4:7] / xd[:, :1] - xo[:, 4:7] / xo[:, :1]).max()
    assert max(d_rot, d_t) < 1e-4 and np.abs(bd - bo).max() < 1e-4


@pytest.mark.parametrize("case", ["constants", "bounds", "beta_lock", "all"])
def test_device_window_lm_constants_bounds_and_beta_lock(api, synth, model, gpu_model, case):
    """Constant parameter blocks, active scale bounds and the stage-2 configuration (beta locked by a 1e5 prior)."""
    F = 10
    seq = synth.make_sequence(model, F, seed=77)
    const = None
    if case in ("constants", "all"):
        const = np.zeros(76, np.uint8)
        for j in (10, 11, 22, 23):
            const[7 + 3 * (j - 1):10 + 3 * (j - 1)] = 1
    bounds = (0.95, 1.02) if case in ("bounds", "all") else (-1e300, 1e300)
    kw = dict(beta_shape=1e5) if case in ("beta_lock", "all") else {}
    (xh, bh, sh), (xd, bd, sd) = _both(api, gpu_model, seq, F, 15, constant=const, bounds=bounds, **kw)
    assert (sd.iterations, sd.n_successful) == (sh.iterations, sh.n_successful)
    # (the 1e5 lock puts 1e10 on the beta block's diagonal: the two elimination orders then differ by more rounding)
    tol_c, tol_x = (1e-6, 1e-5) if case in ("beta_lock", "all") else (1e-9, 1e-7)
    assert abs(sd.final_cost - sh.final_cost) <= tol_c * sh.final_cost, (sd.final_cost, sh.final_cost)
    assert np.abs(xd[:, 1:] - xh[:, 1:]).max() < tol_x and np.abs(bd - bh).max() < tol_x
    if const is not None:
        assert np.all(xd[:, 7 + 27:7 + 33] == 0) and np.all(xd[:, 7 + 63:] == 0)
    if case in ("bounds", "all"):
        assert xd[:, 0].min() >= 0.95 - 1e-12 and xd[:, 0].max() <= 1.02 + 1e-12
        assert np.abs(xd[:, 0] - xh[:, 0]).max() < 1e-6


def test_device_window_lm_103_anchors(api, synth, model, gpu_model):
    """C5 stage 1 at its real size (1024 frames, skip 10 -> 103 anchors, src/main_multi_frame.cpp:109-134)."""
    seq = synth.make_sequence(model, 1024, seed=3)
    ids = list(range(0, 1024, 10))
    class S: pass
    s = S(); offs = [0]; kid = []; uv = []
    for f in ids:
        k0, k1 = seq.kp_offset[f], seq.kp_offset[f + 1]
        kid.append(seq.kp_id[k0:k1]); uv.append(seq.kp_uv[k0:k1]); offs.append(offs[-1] + k1 - k0)
    s.kp_offset = np.array(offs, np.int32); s.kp_id = np.concatenate(kid); s.kp_uv = np.concatenate(uv)
    s.intr = seq.intr; s.R0 = seq.R0[ids]; s.init_params = seq.init_params[ids]
    (xh, bh, sh), (xd, bd, sd) = _both(api, gpu_model, s, len(ids), 10)
    assert (sd.iterations, sd.n_successful) == (sh.iterations, sh.n_successful)
    assert abs(sd.final_cost - sh.final_cost) <= 1e-9 * sh.final_cost
    assert np.abs(xd[:, 1:] - xh[:, 1:]).max() < 1e-7 and np.abs(bd - bh).max() < 1e-7


def test_diagonal_block_factorisation_equals_round_4s_bit_for_bit(tmp_path):
    """dense_inl.h diag_factor16_acc — the 16 x 16 diagonal-block Cholesky inside k_cr_factor, k_lm_step and k_cr_back's block
    inverses, hand-scheduled one-instruction asm statements — in its round-5 form (the rank-1 updates of registers none of
    whose entries is read again are left out: 4.5 instead of 8 per pivot) against round 4's (diag_factor16_acc_r4, kept as the
    reference): every word of L, of the appended rows x L^-T and of 1 / L_mm that a caller reads, on 64 random SPD blocks at
    four sizes of identity padding, and both against a host Cholesky (tools/ubench/diag16.hip, which also times them)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "diag16")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(root, "3dbodyanimation_amd", "csrc"),
                           "-o", exe, os.path.join(root, "tools", "ubench", "diag16.hip")], stderr=subprocess.DEVNULL)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout
    lines = [l for l in res.stdout.splitlines() if l.startswith("nvalid")]
    assert len(lines) == 4 and all("round 5's: 0;" in l for l in lines), res.stdout
    for l in lines:
        assert float(l.split("max")[1]) < 1e-14
