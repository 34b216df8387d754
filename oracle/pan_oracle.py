"""ORACLE (test infrastructure, not product code) -- CPU restatement of the PAN loop.

Restates, in numpy, the reference's per-control-step path `PAN.forward`
(neupan/blocks/pan.py:109-147) and everything it calls, one function per reference
function, each citing the lines it follows.  All arithmetic the reference does in fp32
(neupan/configuration/__init__.py:27) is done in np.float32 here, in the same operation
order; the QP is solved in fp64 (the reference's solver works in double and casts back,
nrmp.py:145-148) by oracle/nrmp_qp.py.

Pinning (see tests/golden/make_golden.py, tests/test_oracle_golden.py):
  * every function except the QP solve is checked against outputs of the UNMODIFIED
    reference code imported in the build container (fixtures in tests/golden/*.npz);
  * G/h are checked against example/model/*/results.txt:1-8 of the reference;
  * the QP solve is PARITY-UNPINNED against ECOS (absent here); see oracle/nrmp_qp.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from math import cos, inf, sin, tan

import numpy as np

from .nrmp_qp import NrmpProblem, solve_nrmp_qp

f32 = np.float32


# ------------------------------------------------------------------ geometry (init time)
def cal_vertices(vertices=None, length=None, width=None, wheelbase=None):
    """robot.py:318-375: rectangle from length/width/wheelbase, or user vertices (2,N)."""
    if vertices is not None:
        v = np.array(vertices, dtype=np.float64)
        return v.T if isinstance(vertices, list) else v
    wb = 0.0 if wheelbase is None else wheelbase
    x0, y0 = -(length - wb) / 2.0, -width / 2.0
    return np.array([[x0, x0 + length, x0 + length, x0], [y0, y0, y0 + width, y0 + width]])


def gen_inequal_from_vertex(vertex):
    """util/__init__.py:161-206: un-normalised edge normals, G x <= h; CW input is
    re-ordered to CCW keeping vertex 0 first."""
    v = np.asarray(vertex, dtype=np.float64)
    n = v.shape[1]
    cross = []
    for i in range(n):
        a, b, c = v[:, i], v[:, (i + 1) % n], v[:, (i + 2) % n]
        cross.append((b[0] - a[0]) * (c[1] - b[1]) - (b[1] - a[1]) * (c[0] - b[0]))
    cross = np.array(cross)
    if not (np.all(cross >= 0) or np.all(cross <= 0)):
        return None, None
    if np.all(cross <= 0):  # clockwise
        v = np.hstack([v[:, 0:1], v[:, 1:][:, ::-1]])
    G, h = np.zeros((n, 2)), np.zeros((n, 1))
    for i in range(n):
        p, q = v[:, i], v[:, (i + 1) % n]
        G[i, 0], G[i, 1] = q[1] - p[1], -(q[0] - p[0])
        h[i, 0] = G[i, 0] * p[0] + G[i, 1] * p[1]
    return G, h


def downsample_decimation(mat, m):
    """util/__init__.py:285-305."""
    n = mat.shape[1]
    if m >= n:
        return mat
    return mat[:, np.linspace(0, n - 1, m).astype(int)]


# ------------------------------------------------------------------ DUNE network
class ObsPointNetWeights:
    """The 18 tensors of the reference checkpoint (obs_point_net.py:31-46), fp32."""

    LINEAR = (0, 3, 5, 8, 10, 13)
    NORM = (1, 6, 11)

    def __init__(self, state_dict):
        g = lambda k: np.asarray(state_dict[k], dtype=np.float32)
        self.W = [g(f"MLP.{i}.weight") for i in self.LINEAR]   # (out, in)
        self.b = [g(f"MLP.{i}.bias") for i in self.LINEAR]
        self.gamma = [g(f"MLP.{i}.weight") for i in self.NORM]
        self.beta = [g(f"MLP.{i}.bias") for i in self.NORM]

    @classmethod
    def from_checkpoint(cls, path):
        import torch
        sd = torch.load(path, map_location="cpu")            # dune.py:142
        return cls({k: v.detach().numpy() for k, v in sd.items()})


def _layernorm(x, gamma, beta, eps=f32(1e-5)):
    """torch.nn.LayerNorm(32): biased variance, eps inside the sqrt, affine."""
    mean = x.mean(axis=1, keepdims=True, dtype=np.float32)
    xc = x - mean
    var = (xc * xc).mean(axis=1, keepdims=True, dtype=np.float32)
    return xc / np.sqrt(var + eps) * gamma + beta


def obs_point_net(w: ObsPointNetWeights, x):
    """obs_point_net.py:31-49: Lin-LN-Tanh-Lin-ReLU-Lin-LN-Tanh-Lin-ReLU-Lin-LN-Tanh-Lin-ReLU.
    x: (rows, 2) fp32 -> (rows, E) fp32."""
    x = np.asarray(x, dtype=np.float32)
    lin = lambda i, v: v @ w.W[i].T + w.b[i]
    h = np.tanh(_layernorm(lin(0, x), w.gamma[0], w.beta[0]))
    h = np.maximum(lin(1, h), f32(0))
    h = np.tanh(_layernorm(lin(2, h), w.gamma[1], w.beta[1]))
    h = np.maximum(lin(3, h), f32(0))
    h = np.tanh(_layernorm(lin(4, h), w.gamma[2], w.beta[2]))
    return np.maximum(lin(5, h), f32(0))


# ------------------------------------------------------------------ PAN pieces
def generate_point_flow(nom_s, obs_points, point_velocities, T, dt, dune_max_num):
    """pan.py:150-212.  Returns lists (length T+1) of p0 (2,N), R (2,2), global pts (2,N)."""
    obs = np.asarray(obs_points, dtype=np.float32)
    vel = np.zeros_like(obs) if point_velocities is None else np.asarray(point_velocities, dtype=np.float32)
    if obs.shape[1] > dune_max_num:
        obs = downsample_decimation(obs, dune_max_num)
        vel = downsample_decimation(vel, dune_max_num)
    flow, Rl, pl = [], [], []
    vdt = vel * f32(dt)                                           # pan.py:182 (vel*dt first)
    for i in range(T + 1):
        pts = obs + f32(i) * vdt
        th = f32(nom_s[2, i])
        c, s_ = np.cos(th, dtype=np.float32), np.sin(th, dtype=np.float32)
        R = np.array([[c, -s_], [s_, c]], dtype=np.float32)       # pan.py:208
        p0 = R.T @ (pts - np.asarray(nom_s[0:2, i:i + 1], dtype=np.float32))   # pan.py:210
        flow.append(p0.astype(np.float32)); Rl.append(R); pl.append(pts)
    return flow, Rl, pl


def dune_forward(w, G, h, flow, Rl, pl):
    """dune.py:58-127.  Returns mu_list, lam_list, sort_point_list (each length T+1, sorted
    ascending by distance) and min_distance (slice 0 only, dune.py:97-98)."""
    G = np.asarray(G, dtype=np.float32); h = np.asarray(h, dtype=np.float32).reshape(-1, 1)
    total = np.hstack(flow)
    total_mu = obs_point_net(w, total.T).T                         # dune.py:78-82
    mu_l, lam_l, pt_l = [], [], []
    min_distance = inf
    for i, p0 in enumerate(flow):
        n = p0.shape[1]
        mu = total_mu[:, i * n:(i + 1) * n]
        lam = ((-Rl[i]) @ G.T) @ mu                                # dune.py:89
        dist = np.einsum("en,en->n", mu, (G @ p0 - h)).astype(np.float32)   # dune.py:109-127
        if i == 0:
            min_distance = dist.min() if n else inf
        idx = np.argsort(dist, kind="stable")                      # dune.py:100
        mu_l.append(mu[:, idx]); lam_l.append(lam[:, idx].astype(np.float32)); pt_l.append(pl[i][:, idx])
    return mu_l, lam_l, pt_l, min_distance


def generate_coefficient_parameter_value(mu_l, lam_l, pt_l, h, T, M):
    """nrmp.py:220-261.  Returns fa (T,M,2), fb (T,M) fp32 (zeros when no points)."""
    fa = np.zeros((T, M, 2), dtype=np.float32); fb = np.zeros((T, M), dtype=np.float32)
    if not mu_l:
        return fa, fb
    h = np.asarray(h, dtype=np.float32).reshape(-1)
    for t in range(T):
        mu, lam, pt = mu_l[t + 1], lam_l[t + 1], pt_l[t + 1]       # nrmp.py:244: slice t+1
        fa_all = lam.T
        fb_all = np.einsum("nk,nk->n", lam.T, pt.T).astype(np.float32) + mu.T @ h
        pn = min(mu.shape[1], M)
        fa[t, :pn] = fa_all[:pn]; fb[t, :pn] = fb_all[:pn]
        fa[t, pn:] = fa_all[0]; fb[t, pn:] = fb_all[0]             # nrmp.py:258-259
    return fa, fb


def generate_state_parameter_value(nom_s, nom_u, T, dt, kinematics, L=None):
    """robot.py:239-316.  fp32 tensors of python-double trig times fp32 scalars: every
    tensor*python-scalar product is rounded to fp32, pure-python products are rounded once."""
    A = np.zeros((T, 3, 3), dtype=np.float32); B = np.zeros((T, 3, 2), dtype=np.float32)
    C = np.zeros((T, 3), dtype=np.float32)
    for t in range(T):
        phi_t = f32(nom_s[2, t]); v = f32(nom_u[0, t]); psi = f32(nom_u[1, t])
        if kinematics == "omni":                                    # robot.py:304-316
            ph = psi; sp, cp = sin(float(ph)), cos(float(ph))
            A[t] = np.eye(3, dtype=np.float32)
            B[t] = [[f32(cp * dt), (-v) * f32(sp) * f32(dt)], [f32(sp * dt), v * f32(cp) * f32(dt)], [0, 0]]
            C[t] = [ph * v * f32(sp) * f32(dt), (-ph) * v * f32(cp) * f32(dt), 0]
            continue
        sp, cp = sin(float(phi_t)), cos(float(phi_t))
        A[t] = [[1, 0, (-v) * f32(dt) * f32(sp)], [0, 1, v * f32(dt) * f32(cp)], [0, 0, 1]]
        C01 = [phi_t * v * f32(sp) * f32(dt), (-phi_t) * v * f32(cp) * f32(dt)]
        if kinematics == "diff":                                    # robot.py:289-302
            B[t] = [[f32(cp * dt), 0], [f32(sp * dt), 0], [0, f32(dt)]]
            C[t] = [C01[0], C01[1], 0]
        elif kinematics == "acker":                                 # robot.py:272-286
            cps = cos(float(psi)) ** 2
            B[t] = [[f32(cp * dt), 0], [f32(sp * dt), 0],
                    [f32(tan(float(psi)) * dt / L), v * f32(dt) / f32(L * cps)]]
            C[t] = [C01[0], C01[1], (-psi) * v * f32(dt) / f32(L * cps)]
        else:
            raise ValueError("kinematics currently only supports acker, diff or omni")
    return A, B, C


class PanOracle:
    """CPU restatement of `PAN` (pan.py:28-147) for one scene, including the state it
    carries between calls (pan.py:100-105, 215-243)."""

    def __init__(self, T, dt, G, h, weights, kinematics="diff", L=None, iter_num=2,
                 dune_max_num=100, nrmp_max_num=10, iter_threshold=0.1,
                 speed_bound=(inf, inf), acce_bound=(inf, inf),
                 eta=10.0, d_max=1.0, d_min=0.1, q_s=1.0, p_u=1.0, ro_obs=400, bk=0.1):
        self.T, self.dt, self.G, self.h, self.w = T, dt, G, h, weights
        self.kin, self.L = kinematics, L
        self.iter_num, self.dune_max_num, self.M = iter_num, dune_max_num, nrmp_max_num
        self.iter_threshold = iter_threshold
        self.speed_bound, self.acce_bound = speed_bound, acce_bound
        # adjust parameters are fp32 tensors in the reference (nrmp.py:79-95)
        self.eta, self.d_max, self.d_min = f32(eta), f32(d_max), f32(d_min)
        self.q_s = np.asarray(q_s, dtype=np.float32); self.p_u = f32(p_u)
        self.ro_obs, self.bk = ro_obs, bk
        self.no_obs = nrmp_max_num == 0 or dune_max_num == 0       # pan.py:86
        self.current = [None, None, None, None]
        self.min_distance = inf
        self.iters_run = 0
        self.trace = []          # per-iteration (s,u,d) for tests

    def nrmp(self, nom_s, nom_u, ref_s, ref_us, mu_l, lam_l, pt_l):
        """nrmp.py:114-166: build the parameter values, solve in fp64, cast to fp32."""
        A, B, C = generate_state_parameter_value(nom_s, nom_u, self.T, self.dt, self.kin, self.L)
        q = self.q_s.reshape(-1, 1) if self.q_s.ndim else self.q_s
        qref = (q * np.asarray(ref_s, dtype=np.float32)).astype(np.float32)          # nrmp.py:158
        puref = (self.p_u * np.asarray(ref_us, dtype=np.float32)).astype(np.float32)
        if self.M > 0:
            fa, fb = generate_coefficient_parameter_value(mu_l, lam_l, pt_l, self.h, self.T, self.M)
        else:
            fa = fb = None
        pb = NrmpProblem(nom_s, qref, puref, A, B, C, fa, fb, self.q_s, self.p_u, self.eta,
                         self.d_max, self.d_min, self.ro_obs, self.bk,
                         self.speed_bound, self.acce_bound, self.kin)
        s, u, d = solve_nrmp_qp(pb)
        self.last_problem = pb
        cast = lambda a: None if a is None else a.astype(np.float32)                 # nrmp.py:145-148
        return cast(s), cast(u), cast(d)

    def stop_criteria(self, nom_s, nom_u, mu_l, lam_l):
        """pan.py:215-243."""
        if self.current[0] is None:
            self.current = [nom_s, nom_u, mu_l, lam_l]
            return False
        ds = np.linalg.norm(nom_s - self.current[0]); du = np.linalg.norm(nom_u - self.current[1])
        if len(mu_l) == 0 or len(self.current[2]) == 0:
            diff = ds ** 2 + du ** 2
        else:
            eff = min(mu_l[0].shape[1], self.current[2][0].shape[1], self.M)
            md = np.linalg.norm(np.vstack(mu_l)[:, :eff] - np.vstack(self.current[2])[:, :eff]) / eff
            ld = np.linalg.norm(np.vstack(lam_l)[:, :eff] - np.vstack(self.current[3])[:, :eff]) / eff
            diff = md ** 2 + ld ** 2
        self.current = [nom_s, nom_u, mu_l, lam_l]
        self.last_diff = float(diff)
        return bool(diff < self.iter_threshold)

    def forward(self, nom_s, nom_u, ref_s, ref_us, obs_points=None, point_velocities=None):
        """pan.py:109-147."""
        nom_s = np.asarray(nom_s, dtype=np.float32); nom_u = np.asarray(nom_u, dtype=np.float32)
        nom_d = None
        self.trace = []
        self.iters_run = 0
        for _ in range(self.iter_num):
            if obs_points is not None and not self.no_obs and np.asarray(obs_points).shape[1] > 0:
                flow, Rl, pl = generate_point_flow(nom_s, obs_points, point_velocities, self.T, self.dt,
                                                   self.dune_max_num)
                mu_l, lam_l, pt_l, self.min_distance = dune_forward(self.w, self.G, self.h, flow, Rl, pl)
                self.dune_points = pl[0]
                self.nrmp_points = pt_l[0][:, :self.M]
            else:
                mu_l, lam_l, pt_l = [], [], []
            nom_s, nom_u, nom_d = self.nrmp(nom_s, nom_u, ref_s, ref_us, mu_l, lam_l, pt_l)
            self.last_lists = (mu_l, lam_l, pt_l)
            self.trace.append((nom_s, nom_u, nom_d))
            self.iters_run += 1
            if self.stop_criteria(nom_s, nom_u, mu_l, lam_l):
                break
        return nom_s, nom_u, nom_d
