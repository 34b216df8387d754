"""ORACLE (test infrastructure, not product code) -- CPU fp64 restatement of the NRMP QP.

PARITY UNPINNED at this boundary: the reference solves this problem with
cvxpylayers -> diffcp -> ECOS (neupan/blocks/nrmp.py:144, problem built at
nrmp.py:263-383 and neupan/robot/robot.py:73-236).  Those packages are third-party,
unpinned (pyproject.toml:12-24) and absent from this environment, and the reference
ships no test/golden vector for the solve.  What this file does instead:

  * restates the *same optimisation problem* from the same 58 parameter tensors, in the
    reference's variable set (s, u, d) plus one epigraph variable per hinge row -- i.e.
    the UNCONDENSED formulation with the dynamics kept as equality constraints, which is
    deliberately a different formulation from the condensed solver in the HIP kernel;
  * solves it in fp64 with a Mehrotra predictor-corrector interior-point method to KKT
    residual <= 1e-14 or stagnation (scaled; see dense_ipm).  Why 1e-14 and not 1e-12: the QP is nearly flat along
    high-frequency steering directions, and two solvers that both stop at 1e-12 still differ by up to 3e-5 in those
    entries (1.4e-4 from the limit point on an acker problem); one more Newton step (quadratic phase) takes them to
    <= 1e-8 of the limit point, which is what a comparison at 1e-5 needs;
  * provides `kkt_certificate`, an independently coded optimality check (adjoint gradient
    + NNLS multiplier recovery) that is applied to both this solver's and the GPU's output;
  * tests/golden/make_golden.py additionally cross-checks it against HiGHS' QP solver
    (third-party, bundled inside scipy) and commits those vectors.

The problem is strictly convex on its feasible set for the diff robot (q_s>0, bk>0), so any
point passing the certificate is *the* optimum ECOS converges to (to ECOS' 1e-8 tolerances).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

__all__ = ["NrmpProblem", "solve_nrmp_qp", "kkt_certificate", "dense_ipm"]


class NrmpProblem:
    """The numbers that define one NRMP solve.

    Mirrors the reference's parameter list (nrmp.py:152-166, robot.py:85-101,
    robot.py:104-133, nrmp.py:304-342) and its constants (robot.py:53-69):

      nom_s (3,T+1)  para_s             qref_s (3,T+1) para_gamma_a = q_s * ref_s
      puref (T,)     para_gamma_b = p_u * ref_us
      A (T,3,3) B (T,3,2) C (T,3)       linearised dynamics
      fa (T,M,2) fb (T,M)               para_gamma_c / para_zeta_a (None when no_obs)
      q_s scalar or (3,)  p_u  eta  d_max  d_min
      ro_obs, bk                        python constants baked into the problem
      speed_bound (2,), acce_bound (2,) (= max_acce*dt), kinematics ('omni' drops theta cost)
    """

    def __init__(self, nom_s, qref_s, puref, A, B, C, fa, fb, q_s, p_u, eta, d_max, d_min,
                 ro_obs, bk, speed_bound, acce_bound, kinematics="diff"):
        f = lambda a: None if a is None else np.asarray(a, dtype=np.float64)
        self.nom_s, self.qref_s, self.puref = f(nom_s), f(qref_s), f(puref).reshape(-1)
        self.A, self.B, self.C = f(A), f(B), f(C)
        self.T = self.A.shape[0]
        self.C = self.C.reshape(self.T, 3)
        self.fa, self.fb = f(fa), f(fb)
        if self.fb is not None:
            self.fb = self.fb.reshape(self.T, -1)
        self.no_obs = self.fa is None
        self.M = 0 if self.no_obs else self.fa.shape[1]
        q = np.asarray(q_s, dtype=np.float64).reshape(-1)
        self.q_s = np.repeat(q, 3) if q.size == 1 else q
        self.p_u, self.eta = float(p_u), float(eta)
        self.d_max, self.d_min = float(d_max), float(d_min)
        self.ro_obs, self.bk = float(ro_obs), float(bk)
        self.speed_bound = f(speed_bound).reshape(2)
        self.acce_bound = f(acce_bound).reshape(2)
        self.kinematics = kinematics

    # ---- objective pieces, written once and shared by solver / certificate -----------
    def state_weight(self):
        """diag weight w such that state cost = sum_t (q_s*s_t - qref_t)^2 restricted to
        the rows the reference sums (robot.py:161-164: omni uses rows 0:2 only)."""
        mask = np.ones(3)
        if self.kinematics == "omni":
            mask[2] = 0.0
        return mask

    def objective(self, s, u, d):
        """Value of the reference objective (robot.py:142-198, nrmp.py:344-383)."""
        m = self.state_weight()[:, None]
        J = np.sum(m * (self.q_s[:, None] * s - self.qref_s) ** 2)           # robot.py:151-166
        J += np.sum((self.p_u * u[0, :] - self.puref) ** 2)                   # robot.py:151,166
        J += 0.5 * self.bk * np.sum((s - self.nom_s) ** 2)                    # nrmp.py:350
        if not self.no_obs:
            J += -self.eta * np.sum(d)                                        # nrmp.py:382-383
            r = self.hinge_residual(s, d)
            J += 0.5 * self.ro_obs * np.sum(np.minimum(r, 0.0) ** 2)          # robot.py:183-198
        return J

    def hinge_residual(self, s, d):
        """I_dpp rows: fa_t @ s[0:2,t+1] - fb_t - d_t  (robot.py:190-193), shape (T,M)."""
        return np.einsum("tmk,kt->tm", self.fa, s[0:2, 1:]) - self.fb - np.reshape(d, (-1,))[:, None]


# --------------------------------------------------------------------------------------
# generic dense primal-dual interior point:  min 1/2 z'Pz + q'z  s.t. Az=b, Gz<=h
# --------------------------------------------------------------------------------------
# centring target never below this x the largest scaled residual: complementarity may not run more than 100x ahead of
# feasibility.  Without it the gap of ~0.5 % of the benchmark QPs collapsed to 1e-15 while the dual residual sat at 1e-10 ..
# 1e-12 -- the solve "stalled" and returned its best iterate, up to 2e-4 from the optimum in the controls along the flat
# steering directions of the car (found in round 5 through a step the KERNEL had right: tests/parity_tools.py, the kernel's
# own method carries the same floor, oracle/condensed_ipm.py SIGMA_MU_RES)
SIGMA_MU_RES = 0.01


def dense_ipm(P, q, A, b, G, h, tol=1e-14, max_iter=60):
    """Mehrotra predictor-corrector.  Stops when the scaled KKT residuals and the
    complementarity gap are all <= tol; because the reduced KKT matrix becomes extremely
    ill-conditioned as the gap closes (cond ~ 1/gap^2) it tracks the best iterate and
    returns it once three consecutive iterations fail to improve on it."""
    n, m, p = P.shape[0], G.shape[0], A.shape[0]
    z = np.zeros(n)
    y = np.zeros(p)
    w = np.maximum(h - G @ z, 1.0)
    lam = np.ones(m)
    scale_d = 1.0 + np.abs(q).max()
    scale_p = 1.0 + max(np.abs(h).max() if m else 0.0, np.abs(b).max() if p else 0.0)
    K = np.zeros((n + p, n + p))
    best = (np.inf, z, y, lam, w, 0)
    stall = 0
    for it in range(max_iter + 1):
        r_d = P @ z + q + G.T @ lam + A.T @ y
        r_p = G @ z + w - h
        r_e = A @ z - b
        mu = lam @ w / m
        res = max(np.abs(r_d).max() / scale_d, np.abs(r_p).max() / scale_p, (np.abs(r_e).max() if p else 0.0) / scale_p)
        merit = max(res, mu)
        if not np.isfinite(merit):
            break
        if merit < best[0]:
            best = (merit, z, y, lam, w, it)
            stall = 0
        else:
            stall += 1
        # (round 6: the rule of three is an END-GAME rule.  Far from convergence the merit is the complementarity gap, which may sit
        # still for several iterations while feasibility improves: on 7 of 66 560 benchmark QPs this solver -- and the kernel, which
        # shared the rule -- returned an iterate at merit 0.6 - 0.9 after four or five iterations.  Found by scanning 1024 scenes
        # per workload with the kernel's status word; profiles/r06_qp_robustness.txt)
        if merit <= tol or stall >= (3 if best[0] <= 1e-6 else 12) or it == max_iter:
            break
        D = lam / w
        K[:n, :n] = P + G.T @ (D[:, None] * G)
        K[:n, n:] = A.T
        K[n:, :n] = A
        K[n:, n:] = 0.0

        def solve(r_c):
            rhs = np.concatenate([-r_d + G.T @ ((r_c - lam * r_p) / w), -r_e])
            sol = np.linalg.solve(K, rhs)
            dz, dy = sol[:n], sol[n:]
            dw = -r_p - G @ dz
            dl = (-r_c - lam * dw) / w
            return dz, dy, dw, dl

        def max_step(v, dv):
            neg = dv < 0
            return 1.0 if not neg.any() else min(1.0, float(np.min(-v[neg] / dv[neg])))

        with np.errstate(all="ignore"):
            dz, dy, dw, dl = solve(lam * w)
            a_aff = min(max_step(w, dw), max_step(lam, dl))
            mu_aff = (lam + a_aff * dl) @ (w + a_aff * dw) / m
            sigma_mu = max((mu_aff / mu) ** 3 * mu, SIGMA_MU_RES * res)
            # (no second-order corrector on an affine step shorter than 0.1: the kernel's QP_CORRECTOR_MIN_AFF, same reason)
            dz, dy, dw, dl = solve(lam * w + (dw * dl if a_aff >= 0.1 else 0.0) - sigma_mu)
            a = min(1.0, 0.995 * min(max_step(w, dw), max_step(lam, dl)))
            # centrality safeguard of a blocked step (the same rule as the kernel's, QP_CENTRAL_*): no product lam w may fall below
            # 1e-3 x their mean -- one badly centred pair blocks every later step in turn
            if a < 0.9:
                for _ in range(6):
                    prod = (lam + a * dl) * (w + a * dw)
                    if prod.min() >= 1e-3 * prod.mean():
                        break
                    a *= 0.7
            z, y, w, lam = z + a * dz, y + a * dy, w + a * dw, lam + a * dl
    merit, z, y, lam, w, it_used = best
    return z, y, lam, w, {"iters": it_used, "merit": merit}


def _assemble_full(pb: NrmpProblem):
    """Uncondensed problem in z = [s (col-major 3(T+1)), u (2T), d (T), e (T*M)]."""
    T, M = pb.T, pb.M
    ns, nu = 3 * (T + 1), 2 * T
    nd = 0 if pb.no_obs else T
    ne = 0 if pb.no_obs else T * M
    n = ns + nu + nd + ne
    si = lambda k, t: 3 * t + k
    ui = lambda k, t: ns + 2 * t + k
    di = lambda t: ns + nu + t
    ei = lambda t, j: ns + nu + nd + t * M + j

    P = np.zeros((n, n))
    q = np.zeros(n)
    mask = pb.state_weight()
    for t in range(T + 1):
        for k in range(3):
            P[si(k, t), si(k, t)] += 2.0 * mask[k] * pb.q_s[k] ** 2 + pb.bk
            q[si(k, t)] += -2.0 * mask[k] * pb.q_s[k] * pb.qref_s[k, t] - pb.bk * pb.nom_s[k, t]
    for t in range(T):
        P[ui(0, t), ui(0, t)] += 2.0 * pb.p_u ** 2
        q[ui(0, t)] += -2.0 * pb.p_u * pb.puref[t]
    if not pb.no_obs:
        for t in range(T):
            q[di(t)] = -pb.eta
            for j in range(M):
                P[ei(t, j), ei(t, j)] = pb.ro_obs

    # equalities: s_0 = nom_s[:,0]  (robot.py:234);  dynamics (robot.py:200-221)
    A = np.zeros((3 * (T + 1), n))
    b = np.zeros(3 * (T + 1))
    for k in range(3):
        A[k, si(k, 0)] = 1.0
        b[k] = pb.nom_s[k, 0]
    for t in range(T):
        for k in range(3):
            r = 3 * (t + 1) + k
            A[r, si(k, t + 1)] = 1.0
            for c in range(3):
                A[r, si(c, t)] -= pb.A[t, k, c]
            for c in range(2):
                A[r, ui(c, t)] -= pb.B[t, k, c]
            b[r] = pb.C[t, k]

    rows, rhs = [], []

    def add(coeffs, bound):
        if not np.isfinite(bound):
            return
        row = np.zeros(n)
        for idx, v in coeffs:
            row[idx] += v
        rows.append(row)
        rhs.append(bound)

    for t in range(T):
        for k in range(2):
            add([(ui(k, t), 1.0)], pb.speed_bound[k])          # robot.py:233
            add([(ui(k, t), -1.0)], pb.speed_bound[k])
    for t in range(T - 1):
        for k in range(2):
            add([(ui(k, t + 1), 1.0), (ui(k, t), -1.0)], pb.acce_bound[k])   # robot.py:232
            add([(ui(k, t + 1), -1.0), (ui(k, t), 1.0)], pb.acce_bound[k])
    if not pb.no_obs:
        for t in range(T):
            add([(di(t), 1.0)], pb.d_max)                       # nrmp.py:376
            add([(di(t), -1.0)], -max(pb.d_min, 0.0))           # nrmp.py:375 + nonneg (nrmp.py:264)
            for j in range(M):
                # e >= -(fa.s_xy - fb - d)  <=>  -fa.s_xy + d - e <= -fb
                add([(si(0, t + 1), -pb.fa[t, j, 0]), (si(1, t + 1), -pb.fa[t, j, 1]),
                     (di(t), 1.0), (ei(t, j), -1.0)], -pb.fb[t, j])
    G = np.array(rows) if rows else np.zeros((0, n))
    h = np.array(rhs) if rhs else np.zeros(0)
    return P, q, A, b, G, h, (ns, nu, nd, ne)


def solve_nrmp_qp(pb: NrmpProblem, tol=1e-14, return_info=False):
    """Solve the NRMP problem in fp64.  Returns (s (3,T+1), u (2,T), d (1,T) | None)."""
    P, q, A, b, G, h, (ns, nu, nd, ne) = _assemble_full(pb)
    z, y, lam, w, info = dense_ipm(P, q, A, b, G, h, tol=tol)
    T = pb.T
    s = z[:ns].reshape(T + 1, 3).T.copy()
    u = z[ns:ns + nu].reshape(T, 2).T.copy()
    d = None if pb.no_obs else z[ns + nu:ns + nu + nd].reshape(1, T).copy()
    if return_info:
        info.update(lam=lam, slack=w)
        return s, u, d, info
    return s, u, d


# --------------------------------------------------------------------------------------
# independent optimality certificate
# --------------------------------------------------------------------------------------
def kkt_certificate(pb: NrmpProblem, s, u, d, act_tol=1e-3):
    """Optimality check that shares no code with the solvers.

    Works in the reduced space x=(u,d): gradient of the reference objective w.r.t. x by a
    backward (adjoint) sweep through the dynamics, feasibility of every bound, and
    recovery of non-negative multipliers for the near-active bounds by NNLS.

    Returns dict(dyn=max dynamics residual, feas=max bound violation,
                 stat=||grad + C_act' lam||_inf with lam>=0 from NNLS over the bounds whose
                 slack is <= act_tol, comp=max lam_i*slack_i over those bounds, n_active).
    An interior-point solution leaves weakly active bounds with slack ~1e-5 and a tiny
    multiplier, hence the generous act_tol paired with the complementarity product.
    """
    from scipy.optimize import nnls

    s = np.asarray(s, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    T = pb.T
    d = np.zeros(T) if d is None else np.asarray(d, dtype=np.float64).reshape(T)

    dyn = np.abs(s[:, 0] - pb.nom_s[:, 0]).max()
    for t in range(T):
        dyn = max(dyn, np.abs(s[:, t + 1] - (pb.A[t] @ s[:, t] + pb.B[t] @ u[:, t] + pb.C[t])).max())

    mask = pb.state_weight()
    dJds = 2.0 * (mask * pb.q_s)[:, None] * (pb.q_s[:, None] * s - pb.qref_s) + pb.bk * (s - pb.nom_s)
    gd = np.zeros(T)
    if not pb.no_obs:
        r = pb.hinge_residual(s, d)
        neg = np.minimum(r, 0.0)                      # d/dr 0.5*ro*min(r,0)^2 = ro*min(r,0)
        dJds[0:2, 1:] += pb.ro_obs * np.einsum("tm,tmk->kt", neg, pb.fa)
        gd = -pb.eta - pb.ro_obs * neg.sum(axis=1)
    gu = np.zeros((2, T))
    gu[0, :] = 2.0 * pb.p_u * (pb.p_u * u[0, :] - pb.puref)
    adj = np.zeros(3)
    for t in range(T - 1, -1, -1):                   # adjoint sweep
        adj = dJds[:, t + 1] + adj
        gu[:, t] += pb.B[t].T @ adj
        adj = pb.A[t].T @ adj
    grad = np.concatenate([gu.T.reshape(-1), gd if not pb.no_obs else np.zeros(0)])
    n = grad.size

    cons = []   # (row, slack)
    ui = lambda k, t: 2 * t + k
    def add(coeffs, bound, val):
        if not np.isfinite(bound):
            return
        row = np.zeros(n)
        for i, v in coeffs:
            row[i] = v
        cons.append((row, bound - val))
    for t in range(T):
        for k in range(2):
            add([(ui(k, t), 1.0)], pb.speed_bound[k], u[k, t])
            add([(ui(k, t), -1.0)], pb.speed_bound[k], -u[k, t])
    for t in range(T - 1):
        for k in range(2):
            dv = u[k, t + 1] - u[k, t]
            add([(ui(k, t + 1), 1.0), (ui(k, t), -1.0)], pb.acce_bound[k], dv)
            add([(ui(k, t + 1), -1.0), (ui(k, t), 1.0)], pb.acce_bound[k], -dv)
    if not pb.no_obs:
        for t in range(T):
            add([(2 * T + t, 1.0)], pb.d_max, d[t])
            add([(2 * T + t, -1.0)], -max(pb.d_min, 0.0), -d[t])
    feas = max([0.0] + [-sl for _, sl in cons])
    act = [(row, sl) for row, sl in cons if sl <= act_tol]
    comp = 0.0
    if act:
        Ca = np.array([row for row, _ in act])
        lam, _ = nnls(Ca.T, -grad, maxiter=50 * Ca.shape[0])
        stat = np.abs(grad + Ca.T @ lam).max()
        comp = float(np.max(lam * np.maximum(np.array([sl for _, sl in act]), 0.0)))
    else:
        stat = np.abs(grad).max()
    return {"dyn": float(dyn), "feas": float(feas), "stat": float(stat), "comp": comp,
            "n_active": len(act)}
