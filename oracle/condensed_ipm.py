"""ORACLE-SIDE PROTOTYPE (test infrastructure) of the algorithm the HIP QP kernel runs.

This is a numpy transliteration of `nrmp_qp_kernel` (neupan_amd/csrc/nrmp_qp.hip): the
NRMP problem (reference: neupan/blocks/nrmp.py:263-383, neupan/robot/robot.py:142-236)
condensed onto x = (u, d) by eliminating the states through the linearised dynamics, with
the squared-hinge rows handled through their own stationarity condition e = lam_f / ro_obs
(so no epigraph variables are carried), solved by a Mehrotra predictor-corrector.

It exists so that tests can (1) compare the condensed algorithm with the uncondensed
oracle (oracle/nrmp_qp.py) on the CPU, where failures are easy to debug, and (2) compare
the GPU kernel with the same algorithm iteration by iteration.  It is NOT on any product
path: only tests/ import it.
"""
from __future__ import annotations

import numpy as np

from .nrmp_qp import NrmpProblem


def condense(pb: NrmpProblem):
    """Return H (n,n), g (n), F (mf,n), f (mf), Crows (mc,n), c (mc), Phi, cvec with
    x = [u_0x,u_0y,...,u_{T-1}y, d_0..d_{T-1}], n = 3T (2T when no_obs)."""
    T, M = pb.T, pb.M
    nu = 2 * T
    n = nu + (0 if pb.no_obs else T)
    Phi = np.zeros((T + 1, 3, nu))
    cv = np.zeros((T + 1, 3))
    cv[0] = pb.nom_s[:, 0]
    for t in range(T):
        Phi[t + 1] = pb.A[t] @ Phi[t]
        Phi[t + 1][:, 2 * t:2 * t + 2] += pb.B[t]
        cv[t + 1] = pb.A[t] @ cv[t] + pb.C[t]
    mask = pb.state_weight()
    Ws = 2.0 * mask * pb.q_s ** 2 + pb.bk            # d2/ds2 of state + proximal cost
    H = np.zeros((n, n))
    g = np.zeros(n)
    for t in range(1, T + 1):
        H[:nu, :nu] += Phi[t].T @ (Ws[:, None] * Phi[t])
        lin = 2.0 * mask * pb.q_s * (pb.q_s * cv[t] - pb.qref_s[:, t]) + pb.bk * (cv[t] - pb.nom_s[:, t])
        g[:nu] += Phi[t].T @ lin
    for t in range(T):
        H[2 * t, 2 * t] += 2.0 * pb.p_u ** 2
        g[2 * t] += -2.0 * pb.p_u * pb.puref[t]
    F = np.zeros((0, n)); f = np.zeros(0)
    if not pb.no_obs:
        g[nu:] = -pb.eta
        F = np.zeros((T * M, n)); f = np.zeros(T * M)
        for t in range(T):
            for j in range(M):
                F[t * M + j, :nu] = pb.fa[t, j] @ Phi[t + 1][0:2]
                F[t * M + j, nu + t] = -1.0
                f[t * M + j] = pb.fb[t, j] - pb.fa[t, j] @ cv[t + 1][0:2]
    rows, rhs = [], []

    def add(coeffs, bound):
        if not np.isfinite(bound):
            return
        r = np.zeros(n)
        for i, v in coeffs:
            r[i] = v
        rows.append(r); rhs.append(bound)
    for t in range(T):
        for k in range(2):
            add([(2 * t + k, 1.0)], pb.speed_bound[k]); add([(2 * t + k, -1.0)], pb.speed_bound[k])
    for t in range(T - 1):
        for k in range(2):
            add([(2 * t + 2 + k, 1.0), (2 * t + k, -1.0)], pb.acce_bound[k])
            add([(2 * t + 2 + k, -1.0), (2 * t + k, 1.0)], pb.acce_bound[k])
    if not pb.no_obs:
        for t in range(T):
            add([(nu + t, 1.0)], pb.d_max); add([(nu + t, -1.0)], -max(pb.d_min, 0.0))
    C = np.array(rows) if rows else np.zeros((0, n)); c = np.array(rhs) if rhs else np.zeros(0)
    return H, g, F, f, C, c, Phi, cv


WARM_DELTA = 0.003      # QP_WARM_DELTA in nrmp_qp.hip
# a warm attempt is dropped (the solve restarts cold) when its merit exceeds these at iteration 0 / 3 / 6, and repeated
# cold when it ends above WARM_ACCEPT -- the kernel's rules (nrmp_qp.hip, "a warm start that is not paying off").  The
# gate on the PREVIOUS solve (it converged to 1e-12) is the caller's: pass warm=None when it fails.
WARM_DROP = {0: 0.05, 3: 3e-3, 6: 1e-4}
WARM_ACCEPT = 1e-10
# the kernel's interior-point heuristics (same names without the QP_ prefix; tests/tools/qp_step_study.py tuned them)
STEP_ETA = 0.995        # fraction of the step to the boundary: max(STEP_ETA, 1 - mu), capped at 1 - STEP_CAP
STEP_CAP = 1e-6
START_MU = 3.0          # cold start: multipliers = START_MU / slack
SIGMA_MU_MIN = 1e-15    # floor of the centring target sigma * mu
SIGMA_MU_RES = 0.01     # ... nor below this x the largest scaled residual: complementarity may not run more than 100x ahead of
                        #   feasibility.  Without it 0.2 - 0.5 % of the benchmark QPs drove mu to 1e-15 while the dual residual sat at
                        #   1e-10 .. 1e-12 and ended there (three non-improving iterations: "stalled"); with it 1 of 3 360 ends above
                        #   1e-13 (1.8e-13), for +0.2 .. +1.0 iterations per solve (profiles/r05_qp_stall_study.txt)
RETRY_MERIT = 1e-9      # a cold solve that ends above this is repeated once with unit multipliers (round 2's start)
# "the best iterate stands after STALL non-improving iterations" is an END-GAME rule (a merit that wanders at 1e-10 .. 1e-12); far from
# convergence the merit is the complementarity gap, which may sit still for several iterations while feasibility improves (heavy
# centring, short steps): there the patience is STALL_FAR.  (Round 6: three solves in 5 120 of the 8-edge hull's, ended at merit 0.65 -
# 0.86 by the rule of three, converge in 14 - 19 iterations with it.)
STALL_NEAR_MERIT = 1e-6
STALL_NEAR = 3
STALL_FAR = 8           # QP_STALL_FAR (profiles/r06_qp_stall_patience.txt: the solves that converged before are untouched)
# Centrality safeguard of the step (round 6): a step that leaves one complementarity product far behind the others (l_i w_i < CENTRAL_GAMMA x
# their mean) jams the method -- the next steps are blocked by that pair in turn (one solve in 512 of the moving-cloud workload sat at
# mu = 1e-3 for 27 iterations with every residual at 1e-10).  The step is shortened (x CENTRAL_SHRINK, at most CENTRAL_TRIES times) until
# the products stay inside the wide neighbourhood; checked only for steps shorter than CENTRAL_ALPHA (a full step of the end game
# keeps them balanced by itself).  CENTRAL_GAMMA = 0: off.
CENTRAL_GAMMA = 1e-3    # QP_CENTRAL_* of nrmp_qp_device.h (profiles/r06_qp_centrality.txt)
CENTRAL_SHRINK = 0.7
CENTRAL_TRIES = 6
CENTRAL_ALPHA = 0.9
# Mehrotra's second-order corrector is built from the affine direction; when that direction can only be followed for a few per cent of
# its length (a_aff < CORRECTOR_MIN_AFF) the products dw_aff * dl_aff describe a point the iterate never gets near, and the "correction"
# can drive the iterate round a cycle (round 6: one solve in 10 240 of the shipped polygon robot's, mu going 2e-3 -> 6e-3 -> 4e-3 -> 8e-3
# -> 2e-3 ... with every residual below 1e-9, both cold starts, 51 iterations).  Such a step is taken as a plain centring step
# (the known safeguard of Mehrotra-type methods for short affine steps).  0: off.
CORRECTOR_MIN_AFF = 0.1  # QP_CORRECTOR_MIN_AFF of nrmp_qp_device.h (profiles/r06_qp_corrector.txt)


def solve_condensed(pb: NrmpProblem, tol=1e-14, max_iter=40, trace=None, warm=None, _alt=False, sigma_mu_res=None):
    """warm = (x, lc, lf) of a previous, similar solve: the kernel's warm start across the PAN
    iterations of one forward call (multipliers and slacks floored at WARM_DELTA), with the kernel's drop rules: the
    attempt is abandoned for a cold start at iteration 0 / 6 when its merit is above WARM_DROP, and a warm-started
    solve that ends above WARM_ACCEPT is repeated cold.  info["warm_code"]: 0 cold, 1 warm start used, 2 / 3 dropped at
    iteration 0 / later, 4 not converged, 5 a cold solve that jammed and was repeated from unit multipliers (qp_info[15] of
    the kernel)."""
    H, g, F, f, C, c, Phi, cv = condense(pb)
    smr = SIGMA_MU_RES if sigma_mu_res is None else sigma_mu_res      # (the gradient path passes 0: the kernel's BWD instantiations do)
    n = H.shape[0]; T = pb.T; nu = 2 * T
    ro = pb.ro_obs
    mc, mf = C.shape[0], F.shape[0]
    m = mc + mf
    # start: u inside its box (nominal controls clipped), d mid-range
    x = np.zeros(n)
    if not pb.no_obs:
        x[nu:] = 0.5 * (max(pb.d_min, 0.0) + pb.d_max)
    # slacks >= 1, multipliers on the central path of mu = START_MU (the hinge slack contains its own multiplier,
    # w = F x - f + l/ro: one fixed-point round, as QP_COLD_INIT does it)
    wc = np.maximum(c - C @ x, 1.0)
    hx = F @ x - f
    wf = np.maximum(hx + 1.0 / ro, 1.0)
    if START_MU is None or _alt:        # round 2's start, unit multipliers: the second attempt of a jammed cold solve (and the study tool)
        lc = np.ones(mc); lf = np.ones(mf)
    else:
        lc = START_MU / wc
        lf = START_MU / wf
        wf = np.maximum(hx + lf / ro, 1.0); lf = START_MU / wf
    if warm is not None:
        x = np.array(warm[0], dtype=float)
        if not pb.no_obs:
            x[nu:] = np.clip(x[nu:], max(pb.d_min, 0.0), pb.d_max)
        lc = np.maximum(warm[1], WARM_DELTA); lf = np.maximum(warm[2], WARM_DELTA)
        wc = np.maximum(c - C @ x, WARM_DELTA); wf = np.maximum(F @ x - f + lf / ro, WARM_DELTA)
    scale_d = 1.0 + np.abs(g).max()
    scale_p = 1.0 + (np.abs(c).max() if mc else 0.0)
    best = (np.inf, x.copy(), 0)
    stall = 0
    lam_out = (lc, lf)
    for it in range(max_iter + 1):
        r1 = H @ x + g + C.T @ lc - F.T @ lf
        r2 = C @ x + wc - c
        r3 = F @ x - f + lf / ro - wf
        mu = (lc @ wc + lf @ wf) / max(m, 1)
        res = max(np.abs(r1).max() / scale_d, (np.abs(r2).max() if mc else 0.0) / scale_p,
                  (np.abs(r3).max() if mf else 0.0) / scale_p)
        merit = max(res, mu)
        if trace is not None:
            trace.append(dict(it=it, merit=merit, mu=mu, x=x.copy()))
        if not np.isfinite(merit):
            break
        if warm is not None and it in WARM_DROP and merit > WARM_DROP[it]:
            out = solve_condensed(pb, tol=tol, max_iter=max_iter, trace=trace, sigma_mu_res=sigma_mu_res)
            out[3]["warm_code"] = 2 if it == 0 else 3
            out[3]["iters_total"] = out[3]["iters_total"] + it
            return out
        if merit < best[0]:
            best = (merit, x.copy(), it); stall = 0
        else:
            stall += 1
        if merit <= tol or stall >= (STALL_NEAR if best[0] <= STALL_NEAR_MERIT else STALL_FAR) or it == max_iter or mu < 1e-17:
            break
        Dc = lc / wc
        Df = lf / (wf + lf / ro)
        K = H + C.T @ (Dc[:, None] * C) + F.T @ (Df[:, None] * F)
        try:
            L = np.linalg.cholesky(K)
        except np.linalg.LinAlgError:      # gap closed past fp64 resolution: keep the best iterate
            break

        def solve(r4c, r4f):
            rhs = -r1 - C.T @ ((lc * r2 - r4c) / wc) - F.T @ ((r4f + lf * r3) / (wf + lf / ro))
            dx = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            dwc = -r2 - C @ dx
            dlc = (-r4c - lc * dwc) / wc
            dlf = -(r4f + lf * r3 + lf * (F @ dx)) / (wf + lf / ro)
            dwf = F @ dx + dlf / ro + r3
            return dx, dwc, dlc, dwf, dlf

        def max_step(v, dv):
            neg = dv < 0
            return 1.0 if not neg.any() else min(1.0, float(np.min(-v[neg] / dv[neg])))

        dx, dwc, dlc, dwf, dlf = solve(lc * wc, lf * wf)
        a_aff = min(max_step(wc, dwc), max_step(lc, dlc), max_step(wf, dwf), max_step(lf, dlf))
        mu_aff = ((lc + a_aff * dlc) @ (wc + a_aff * dwc) + (lf + a_aff * dlf) @ (wf + a_aff * dwf)) / max(m, 1)
        sigma_mu = max((mu_aff / mu) ** 3 * mu, SIGMA_MU_MIN, smr * res)
        cw = 1.0 if a_aff >= CORRECTOR_MIN_AFF else 0.0
        dx, dwc, dlc, dwf, dlf = solve(lc * wc + cw * dwc * dlc - sigma_mu, lf * wf + cw * dwf * dlf - sigma_mu)
        eta = min(max(STEP_ETA, 1.0 - mu), 1.0 - STEP_CAP)
        a = min(1.0, eta * min(max_step(wc, dwc), max_step(lc, dlc), max_step(wf, dwf), max_step(lf, dlf)))
        if CENTRAL_GAMMA > 0.0 and a < CENTRAL_ALPHA:
            for _ in range(CENTRAL_TRIES):
                prod = np.concatenate([(lc + a * dlc) * (wc + a * dwc), (lf + a * dlf) * (wf + a * dwf)])
                if prod.min() >= CENTRAL_GAMMA * prod.mean():
                    break
                a *= CENTRAL_SHRINK
        x = x + a * dx; wc = wc + a * dwc; lc = lc + a * dlc; wf = wf + a * dwf; lf = lf + a * dlf
        lam_out = (lc, lf)
    if warm is not None and not best[0] <= WARM_ACCEPT:
        out = solve_condensed(pb, tol=tol, max_iter=max_iter, trace=trace, sigma_mu_res=sigma_mu_res)
        out[3]["warm_code"] = 4
        out[3]["iters_total"] = out[3]["iters_total"] + it
        return out
    if warm is None and not _alt and START_MU is not None and not best[0] <= RETRY_MERIT:      # jammed: the other start, once
        out = solve_condensed(pb, tol=tol, max_iter=max_iter, trace=trace, _alt=True, sigma_mu_res=sigma_mu_res)
        out[3]["warm_code"] = 5
        out[3]["iters_total"] = out[3]["iters_total"] + it
        return out
    merit, x, it_used = best
    u = x[:nu].reshape(T, 2).T.copy()
    s = np.stack([Phi[t] @ x[:nu] + cv[t] for t in range(T + 1)], axis=1)
    d = None if pb.no_obs else x[nu:].reshape(1, T).copy()
    return s, u, d, {"iters": it_used, "merit": merit, "warm": (x, lam_out[0], lam_out[1]), "iters_total": it,
                     "warm_code": 0 if warm is None else 1}
