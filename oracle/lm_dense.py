"""Dense numpy Levenberg-Marquardt over the oracle evaluator (TEST INFRASTRUCTURE ONLY).

An independent restatement of the Ceres 1.14 trust-region LM the reference configures
(include/Sim3BA.h:472-479, include/MultiFrameBA.h:144-151): full dense Jacobian of every
residual block, Triggs-corrected Huber rows, Jacobi scaling, dense Cholesky of the damped normal equations.
The product's solver (3dbodyanimation_amd/csrc/host_solver.cpp) implements the same published algorithm
with a structured linear solve and the HIP evaluator; the parity tests compare the fitted parameters.
PARITY UNPINNED w.r.t. a real Ceres build (Ceres is not available in this environment).

sparse=True keeps the Jacobian and the normal equations in scipy.sparse form (the same rows, the same LM, a sparse LU of
the damped system instead of a dense Cholesky), so that the checker reaches the reference's stage-1 size (103 anchors,
7,838 unknowns: a dense Jacobian of 19,917 x 7,838 would be 1.25 GB and 2.4 TFLOP per normal-equation build).
"""
from __future__ import annotations

import numpy as np

from . import oracle as O

NP_ = 76


def _rows(om, seq, x, beta, n_cols, use_shape, pose_blend, beta_pose, ogmm, beta_shape, lam, huber, want_jac, jac_mode=0):
    """All residual rows (robustified) and the dense Jacobian of one problem with a shared beta."""
    F = x.shape[0]
    nb = n_cols - NP_
    n = F * NP_ + nb
    r, J = om.evaluate_batch(seq, x, beta if nb else np.zeros(10), n_cols, use_shape, pose_blend,
                             mode=jac_mode if want_jac else 0, want_jac=want_jac)
    K = len(r) // 2
    s = r.reshape(K, 2)
    sq = (s ** 2).sum(1)
    rho = np.array([O.huber(huber, v) for v in sq]) if K else np.zeros((0, 3))
    cost = 0.5 * rho[:, 0].sum() if K else 0.0
    sw = np.sqrt(rho[:, 1]) if K else np.zeros(0)
    res = [np.repeat(sw, 2) * r]
    jac = []
    if want_jac:
        Jd = np.zeros((2 * K, n))
        for f in range(F):
            k0, k1 = seq.kp_offset[f], seq.kp_offset[f + 1]
            Jd[2 * k0:2 * k1, f * NP_:(f + 1) * NP_] = J[2 * k0:2 * k1, :NP_]
            if nb:
                Jd[2 * k0:2 * k1, F * NP_:] = J[2 * k0:2 * k1, NP_:]
        jac.append(np.repeat(sw, 2)[:, None] * Jd)
    if beta_pose > 0:
        for f in range(F):
            rp, Jp, _ = O.pose_prior(ogmm, beta_pose, x[f, 7:], want_jac)
            res.append(rp)
            cost += 0.5 * rp @ rp
            if want_jac:
                Jf = np.zeros((len(rp), n)); Jf[:, f * NP_ + 7:(f + 1) * NP_] = Jp
                jac.append(Jf)
    if beta_shape > 0 and nb:
        rs = beta_shape * beta
        res.append(rs); cost += 0.5 * rs @ rs
        if want_jac:
            Jf = np.zeros((nb, n)); Jf[:, F * NP_:] = beta_shape * np.eye(nb)
            jac.append(Jf)
    if lam > 0:
        src = np.concatenate([np.arange(4, 7), np.arange(1, 4), np.arange(7, NP_)])
        for f in range(F - 1):
            rt = lam * (x[f, src] - x[f + 1, src])
            res.append(rt); cost += 0.5 * rt @ rt
            if want_jac:
                Jf = np.zeros((75, n))
                Jf[np.arange(75), f * NP_ + src] = lam
                Jf[np.arange(75), (f + 1) * NP_ + src] = -lam
                jac.append(Jf)
    return cost, np.concatenate(res), (np.concatenate(jac, 0) if want_jac else None)


def _rows_sparse(om, seq, x, beta, n_cols, use_shape, pose_blend, beta_pose, ogmm, beta_shape, lam, huber, want_jac, jac_mode=0):
    """_rows with the Jacobian as a scipy.sparse CSR matrix (identical rows, identical values)."""
    import scipy.sparse as sp
    F = x.shape[0]
    nb = n_cols - NP_
    n = F * NP_ + nb
    r, J = om.evaluate_batch(seq, x, beta if nb else np.zeros(10), n_cols, use_shape, pose_blend,
                             mode=jac_mode if want_jac else 0, want_jac=want_jac)
    K = len(r) // 2
    sq = (r.reshape(K, 2) ** 2).sum(1)
    rho = np.array([O.huber(huber, v) for v in sq]) if K else np.zeros((0, 3))
    cost = 0.5 * rho[:, 0].sum() if K else 0.0
    sw2 = np.repeat(np.sqrt(rho[:, 1]), 2) if K else np.zeros(0)
    res = [sw2 * r]
    rows, cols, vals = [], [], []
    row0 = 2 * K
    if want_jac and K:
        fk = np.repeat(np.repeat(np.arange(F), np.diff(seq.kp_offset)), 2)            # frame of every reprojection row
        rr = np.arange(2 * K)
        rows.append(np.repeat(rr, NP_)); cols.append((fk[:, None] * NP_ + np.arange(NP_)[None, :]).ravel())
        vals.append((sw2[:, None] * J[:, :NP_]).ravel())
        if nb:
            rows.append(np.repeat(rr, nb)); cols.append(np.tile(F * NP_ + np.arange(nb), 2 * K))
            vals.append((sw2[:, None] * J[:, NP_:]).ravel())
    if beta_pose > 0:
        for f in range(F):
            rp, Jp, _ = O.pose_prior(ogmm, beta_pose, x[f, 7:], want_jac)
            res.append(rp); cost += 0.5 * rp @ rp
            if want_jac:
                m = len(rp)
                rows.append(np.repeat(row0 + np.arange(m), 69)); cols.append(np.tile(f * NP_ + 7 + np.arange(69), m))
                vals.append(np.asarray(Jp).ravel())
            row0 += len(rp)
    if beta_shape > 0 and nb:
        rs = beta_shape * beta
        res.append(rs); cost += 0.5 * rs @ rs
        if want_jac:
            rows.append(row0 + np.arange(nb)); cols.append(F * NP_ + np.arange(nb)); vals.append(np.full(nb, beta_shape))
        row0 += nb
    if lam > 0:
        src = np.concatenate([np.arange(4, 7), np.arange(1, 4), np.arange(7, NP_)])
        for f in range(F - 1):
            rt = lam * (x[f, src] - x[f + 1, src])
            res.append(rt); cost += 0.5 * rt @ rt
            if want_jac:
                rows.append(np.tile(row0 + np.arange(75), 2)); cols.append(np.concatenate([f * NP_ + src, (f + 1) * NP_ + src]))
                vals.append(np.concatenate([np.full(75, lam), np.full(75, -lam)]))
            row0 += 75
    rvec = np.concatenate(res)
    Js = None
    if want_jac:
        Js = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(len(rvec), n))
    return cost, rvec, Js


def solve(om, seq, x0, beta0, n_cols=86, use_shape=True, pose_blend=True, beta_pose=0.0, ogmm=None, beta_shape=0.0,
          lam=0.0, huber=3.0, max_iters=100, constant=None, scale_bounds=(0.3, 3.0), verbose=False, jac_mode=0, sparse=False):
    """One problem over all frames of `seq` with a shared beta.  Returns x, beta, info.
    jac_mode 0: analytic Jacobian; 1: the reference's stride-4 dual-number passes (DynamicAutoDiffCostFunction).
    sparse: scipy.sparse Jacobian / normal equations and a sparse LU (no pivoting: its diagonal is positive exactly when the
    dense Cholesky would succeed) instead of dense algebra — same rows, same LM."""
    rows_fn = _rows_sparse if sparse else _rows
    F = x0.shape[0]
    nb = n_cols - NP_
    x = x0.copy(); beta = np.array(beta0, float).copy() if nb else np.zeros(0)
    n = F * NP_ + nb
    free = np.ones(n, bool)
    if constant is not None:
        for f in range(F):
            free[f * NP_:(f + 1) * NP_] = ~np.asarray(constant, bool)
    args = (n_cols, use_shape, pose_blend, beta_pose, ogmm, beta_shape, lam, huber)
    cost, r, J = rows_fn(om, seq, x, beta, *args, True, jac_mode)
    info = dict(initial_cost=cost, iterations=0, n_ok=0, n_bad=0, termination=1)
    radius, dec = 1e4, 2.0
    scale = None
    for it in range(max_iters):
        H = (J.T @ J).tocsc() if sparse else J.T @ J
        g = J.T @ r
        hdiag = H.diagonal() if sparse else np.diag(H)
        if scale is None:
            scale = 1.0 / (1.0 + np.sqrt(hdiag))
        gp = g.copy()
        for f in range(F):
            s0 = x[f, 0]
            gp[f * NP_] = s0 - np.clip(s0 - g[f * NP_], *scale_bounds)
        if np.abs(gp[free]).max() <= 1e-10:
            info["termination"] = 0; break
        gs = g * scale
        idx = np.where(free)[0]
        try:
            ds = np.zeros(n)
            if sparse:
                import scipy.sparse as sp
                import scipy.sparse.linalg as spl
                S = sp.diags(scale)
                Hs = (S @ H @ S).tocsc()
                Hd = (Hs + sp.diags(np.clip(Hs.diagonal(), 1e-6, 1e32) / radius)).tocsc()
                lu = spl.splu(Hd[idx][:, idx].tocsc(), permc_spec="NATURAL", diag_pivot_thresh=0.0,
                              options=dict(SymmetricMode=True))
                if not np.all(lu.U.diagonal() > 0.0) or np.any(lu.perm_r != np.arange(len(idx))):
                    raise np.linalg.LinAlgError("not positive definite")
                ds[idx] = -lu.solve(gs[idx])
            else:
                Hs = H * np.outer(scale, scale)
                Hd = Hs + np.diag(np.clip(np.diag(Hs), 1e-6, 1e32) / radius)
                L = np.linalg.cholesky(Hd[np.ix_(idx, idx)])
                ds[idx] = -np.linalg.solve(L.T, np.linalg.solve(L, gs[idx]))
        except (np.linalg.LinAlgError, RuntimeError):
            radius /= dec; dec *= 2; info["n_bad"] += 1; info["iterations"] += 1
            continue
        d = ds * scale
        xn = x + d[:F * NP_].reshape(F, NP_)
        clipped = np.clip(xn[:, 0], *scale_bounds)
        if np.any(clipped != xn[:, 0]):
            d[np.arange(F) * NP_] = clipped - x[:, 0]
            xn[:, 0] = clipped
        bn = beta + d[F * NP_:] if nb else beta
        model = -(d @ g) - 0.5 * d @ (H @ d)
        if np.linalg.norm(d) <= 1e-8 * (np.sqrt((x ** 2).sum() + (beta ** 2).sum()) + 1e-8):
            info["termination"] = 0; break
        new_cost, rn, _ = rows_fn(om, seq, xn, bn, *args, False)
        info["iterations"] += 1
        change = cost - new_cost
        rho = change / model
        if np.isfinite(new_cost) and model > 0 and rho > 1e-3:
            x, beta = xn, bn
            old = cost
            cost, r, J = rows_fn(om, seq, x, beta, *args, True, jac_mode)
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2 * rho - 1) ** 3)); dec = 2.0
            info["n_ok"] += 1
            if verbose:
                print(f"[dense-lm] it {info['iterations']} cost {cost:.6e} rho {rho:.3f} radius {radius:.2e}")
            if abs(change) < 1e-6 * old:
                info["termination"] = 0; break
        else:
            radius /= dec; dec *= 2; info["n_bad"] += 1
            if radius < 1e-32:
                info["termination"] = 2; break
    info["final_cost"] = cost
    return x, beta, info
