"""Gradient of the NRMP solution w.r.t. the adjust parameters.

TEST INFRASTRUCTURE ONLY.  Reference: the adjust parameters are created with requires_grad=True
(neupan/blocks/nrmp.py:79-95) and cvxpylayers differentiates the solution map of the cone program
(nrmp.py:144, diffcp); example/LON/LON_corridor.py:94-127 trains p_u, eta, d_max through it.
PARITY UNPINNED against cvxpylayers/diffcp (not installable here).  For a strictly convex QP with
strict complementarity the derivative of the solution map is unique, so it is pinned instead by
central finite differences of the oracle's forward solve (`backward_fd`).

`backward_ipm` is the algorithm the HIP kernel runs: ONE extra solve with the Newton matrix of the
converged interior-point iterate.  With the perturbed KKT system
    H x + g + C^T lc - F^T lf = 0,   C x + wc = c,   lc*wc = mu,   lf*(F x - f + lf/ro) = mu
and K = H + C^T (lc/wc) C + F^T Df F the sensitivities are  dx = -K^-1 dg,  dx = K^-1 C^T (lc/wc) dc,
hence for upstream gradient gx = dL/dx and v = K^-1 gx:
    dL/dg = -v,      dL/dc = (lc/wc) * (C v),      dL/dtheta = -v^T d(Hx+g)/dtheta.
The states are s_t = Phi_t u + c_t, so gx collects dL/du, Phi^T dL/ds and dL/dd.

Parameter dependence followed (total derivative, as autograd sees it: the parameter tensors
q_s*ref_s and p_u*ref_us are themselves functions of q_s, p_u -- nrmp.py:158-160):
    q_s[i]: 4 q_i sum_t Phi_t[i,:]^T (s_it - ref_it)     (rows the cost sums: robot.py:161-164)
    p_u   : 4 p_u (u_0t - ref_us_t) on entry (0,t)
    eta   : g_d = -eta
    d_max : c of the rows d_t <= d_max;   d_min: c = -max(d_min, 0) of the rows -d_t <= -d_min
Only this direct dependence of ONE solve is covered: inside PAN.forward the reference's autograd
also reaches earlier iterations through the proximal centre nom_s and lam(R(nom_s)); those
recurrent terms are not part of this row.
"""
from __future__ import annotations

import copy

import numpy as np

from .condensed_ipm import condense, solve_condensed
from .nrmp_qp import NrmpProblem, solve_nrmp_qp


def backward_ipm(pb: NrmpProblem, gs, gu, gd, tol=1e-12):
    """gs (3,T+1), gu (2,T), gd (1,T)|None: upstream gradients.  Returns dict q_s (3,), p_u, eta,
    d_max, d_min."""
    H, g, F, f, C, c, Phi, cv = condense(pb)
    T, nu = pb.T, 2 * pb.T
    s, u, d, info = solve_condensed(pb, tol=tol)
    x, lc, lf = info["warm"]
    x = np.concatenate([u.T.reshape(-1), [] if pb.no_obs else d.reshape(-1)])
    ro = pb.ro_obs
    wc = np.maximum(c - C @ x, 1e-300)
    Dc = lc / wc
    K = H + C.T @ (Dc[:, None] * C)
    if not pb.no_obs:
        wf = F @ x - f + lf / ro
        Df = lf / (wf + lf / ro)
        K = K + F.T @ (Df[:, None] * F)
    gx = np.zeros(H.shape[0])
    gx[:nu] = np.asarray(gu, dtype=np.float64).T.reshape(-1)
    for t in range(T + 1):
        gx[:nu] += Phi[t].T @ np.asarray(gs, dtype=np.float64)[:, t]
    if not pb.no_obs and gd is not None:
        gx[nu:] = np.asarray(gd, dtype=np.float64).reshape(-1)
    v = np.linalg.solve(K, gx)
    dc = Dc * (C @ v)
    mask = pb.state_weight()
    out = dict(q_s=np.zeros(3), p_u=0.0, eta=0.0, d_max=0.0, d_min=0.0)
    ref = pb.qref_s / np.where(pb.q_s == 0, 1.0, pb.q_s)[:, None]
    for t in range(1, T + 1):
        sv = Phi[t] @ v[:nu]                                   # state image of v
        out["q_s"] += -4.0 * mask * pb.q_s * sv * (s[:, t] - ref[:, t])
    ref_us = pb.puref / (pb.p_u if pb.p_u != 0 else 1.0)
    out["p_u"] = float(-4.0 * pb.p_u * np.sum(v[0:nu:2] * (u[0] - ref_us)))
    if not pb.no_obs:
        out["eta"] = float(np.sum(v[nu:]))
        # rows are appended in condense(): 4T speed rows, 4(T-1) acceleration rows, then (d_max, d_min) pairs
        base = C.shape[0] - 2 * T
        out["d_max"] = float(np.sum(dc[base::2]))
        out["d_min"] = float(-np.sum(dc[base + 1::2])) if pb.d_min > 0 else 0.0
    return out


def _with(pb: NrmpProblem, **kw):
    q = copy.deepcopy(pb)
    for k, val in kw.items():
        setattr(q, k, val)
    return q


def backward_fd(pb: NrmpProblem, gs, gu, gd, eps=1e-6):
    """central differences of L = <gs,s> + <gu,u> + <gd,d> through the uncondensed oracle solve"""
    def loss(p):
        s, u, d = solve_nrmp_qp(p)
        L = float(np.sum(gs * s) + np.sum(gu * u))
        if d is not None and gd is not None:
            L += float(np.sum(np.asarray(gd).reshape(-1) * np.asarray(d).reshape(-1)))
        return L
    out = dict(q_s=np.zeros(3))
    ref = pb.qref_s / pb.q_s[:, None]
    for i in range(3):
        def pert(h):
            q = pb.q_s.copy(); q[i] += h
            return _with(pb, q_s=q, qref_s=q[:, None] * ref)
        out["q_s"][i] = (loss(pert(eps)) - loss(pert(-eps))) / (2 * eps)
    ref_us = pb.puref / pb.p_u
    out["p_u"] = (loss(_with(pb, p_u=pb.p_u + eps, puref=(pb.p_u + eps) * ref_us)) -
                  loss(_with(pb, p_u=pb.p_u - eps, puref=(pb.p_u - eps) * ref_us))) / (2 * eps)
    for k in ("eta", "d_max", "d_min"):
        out[k] = (loss(_with(pb, **{k: getattr(pb, k) + eps})) - loss(_with(pb, **{k: getattr(pb, k) - eps}))) / (2 * eps)
    return out
