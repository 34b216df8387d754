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
    nom_s : the proximal centre, 0.5 bk |s - nom_s|^2 (nrmp.py:350): d(Hx+g)/dnom_s[:,t] = -bk Phi_t^T,
            so dL/dnom_s[:,t] = bk Phi_t v   (t >= 1; column 0 is the pinned start state)

Recurrence inside PAN.forward (`pan_backward`).  Iteration k+1 receives iteration k's solution as
nom_s / nom_u (pan.py:137).  What of that stays on the reference's autograd graph:
  * para_s = nom_s, the proximal centre and the pinned start (robot.py:100, :178, :234)      -- ON the graph
  * A_t, B_t, C_t: built with torch.Tensor([[python floats]]) (robot.py:272-316)              -- detached
  * mu: evaluated under torch.no_grad() (dune.py:81); R: torch.tensor([[...]]) (pan.py:207),
    hence lam, fa, fb                                                                         -- detached
  * nom_u: only enters A, B, C                                                                -- detached
So the reference's gradient is the chain  theta -> s_1 -> (prox centre of solve 2) -> s_2 -> ... -> s_K
plus every solve's direct dependence on theta; `pan_backward` runs it in reverse.
"""
from __future__ import annotations

import copy

import numpy as np

from .condensed_ipm import condense, solve_condensed
from .nrmp_qp import NrmpProblem, solve_nrmp_qp


def backward_ipm(pb: NrmpProblem, gs, gu, gd, tol=1e-12):
    """(tol: the implicit gradient is taken at the interior-point iterate; at 1e-12 the barrier Newton matrix is still
    well enough conditioned for it -- the forward solves stop at 1e-14, oracle/nrmp_qp.py.)
    gs (3,T+1), gu (2,T), gd (1,T)|None: upstream gradients.  Returns dict q_s (3,), p_u, eta,
    d_max, d_min."""
    H, g, F, f, C, c, Phi, cv = condense(pb)
    T, nu = pb.T, 2 * pb.T
    s, u, d, info = solve_condensed(pb, tol=tol, sigma_mu_res=0.0)      # (the kernel's BWD instantiations: no residual floor)
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
    out = dict(q_s=np.zeros(3), p_u=0.0, eta=0.0, d_max=0.0, d_min=0.0, nom_s=np.zeros((3, T + 1)))
    ref = pb.qref_s / np.where(pb.q_s == 0, 1.0, pb.q_s)[:, None]
    for t in range(1, T + 1):
        sv = Phi[t] @ v[:nu]                                   # state image of v
        out["q_s"] += -4.0 * mask * pb.q_s * sv * (s[:, t] - ref[:, t])
        out["nom_s"][:, t] = pb.bk * sv
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
    out["nom_s"] = np.zeros((3, pb.T + 1))
    for k in range(3):
        for t in range(1, pb.T + 1):                      # the proximal centre only: A, B, C, fa, fb stay as recorded
            def pert(h):
                n = pb.nom_s.copy(); n[k, t] += h
                return _with(pb, nom_s=n)
            out["nom_s"][k, t] = (loss(pert(eps)) - loss(pert(-eps))) / (2 * eps)
    return out


THETA = ("q_s", "p_u", "eta", "d_max", "d_min")


def pan_backward(pbs, gs, gu, gd):
    """Gradient of L(s_K, u_K, d_K) w.r.t. the adjust parameters through ALL solves of one PAN.forward call
    (`pbs`: the K NrmpProblems in execution order), following the reference's autograd graph (module docstring)."""
    tot = dict(q_s=np.zeros(3), p_u=0.0, eta=0.0, d_max=0.0, d_min=0.0)
    for pb in reversed(pbs):
        r = backward_ipm(pb, gs, gu, gd)
        for k in THETA:
            tot[k] = tot[k] + r[k]
        gs, gu, gd = r["nom_s"], np.zeros_like(np.asarray(gu, dtype=np.float64)), None
    return tot


def pan_backward_fd(pbs, gs, gu, gd, eps=1e-5):
    """central differences over the same graph: solve k+1's proximal centre moves with solve k's state trajectory
    (columns 1..T; the fp32 cast between iterations, nrmp.py:145-148, is not differentiated), everything the
    reference detaches stays as recorded"""
    ref = pbs[0].qref_s / pbs[0].q_s[:, None]
    ref_us = pbs[0].puref / pbs[0].p_u
    th0 = dict(q_s=pbs[0].q_s.copy(), p_u=pbs[0].p_u, eta=pbs[0].eta, d_max=pbs[0].d_max, d_min=pbs[0].d_min)
    base = [solve_nrmp_qp(pb)[0] for pb in pbs]

    def loss(th):
        shift = None
        for pb, s0 in zip(pbs, base):
            q = np.asarray(th["q_s"], dtype=np.float64)
            p = _with(pb, q_s=q, qref_s=q[:, None] * ref, p_u=th["p_u"], puref=th["p_u"] * ref_us, eta=th["eta"],
                      d_max=th["d_max"], d_min=th["d_min"])
            if shift is not None:
                n = pb.nom_s.copy(); n[:, 1:] += shift[:, 1:]
                p.nom_s = n
            s, u, d = solve_nrmp_qp(p)
            shift = s - s0
        L = float(np.sum(gs * s) + np.sum(gu * u))
        if d is not None and gd is not None:
            L += float(np.sum(np.asarray(gd).reshape(-1) * np.asarray(d).reshape(-1)))
        return L

    def diff(key, idx=None):
        def at(h):
            th = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in th0.items()}
            if idx is None:
                th[key] = th[key] + h
            else:
                th[key][idx] += h
            return loss(th)
        return (at(eps) - at(-eps)) / (2 * eps)
    out = dict(q_s=np.array([diff("q_s", i) for i in range(3)]))
    for k in ("p_u", "eta", "d_max", "d_min"):
        out[k] = diff(k)
    return out
