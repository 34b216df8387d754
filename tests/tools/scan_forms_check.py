"""Numerics of the scan forms the QP kernel could use instead of its dense Phi products and its serial P_t recursion
(DESIGN.md section 7), on the QPs of benchmark scenes, in fp64 on the CPU.  With A_t = I + a_t e_2' (a_t = (A02, A12, 0)):

  s = Phi u          theta_t = sum_{r<=t} B_r[2,:] u_r ;  xy_t = sum_{r<=t} (a_r theta_{r-1} + B_r[:2,:] u_r)      two prefix sums
  w = Phi' q         l01_t = sum_{r>=t} q_r[:2] ;  l2_t = sum_{r>=t} (q_r[2] + a_{r+1} . l01_{r+1}) ;  w_t = B_t' l_t   two suffix sums
  P_t (3x3)          P_t = S_t + A_{t+1}' P_{t+1} A_{t+1}: with c_t = sum_{r<=t} a_r the blocks are suffix sums of S, S c, c'S c

and reports the largest relative deviation from the dense / recursive evaluation.

    python tests/tools/scan_forms_check.py [workload] [scenes]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import CONFIGS, make_oracle
from neupan_amd.scenes import make_scene
from oracle import condensed_ipm as ci

name = sys.argv[1] if len(sys.argv) > 1 else "diff_1k_T10_K10"
scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = CONFIGS[name]
rng = np.random.default_rng(0)
worst = dict(phi=0.0, phit=0.0, P=0.0)
nqp = 0
for b in range(scenes):
    sc = make_scene(cfg, b, None)
    orc = make_oracle(cfg)
    orig = orc.nrmp

    def hook(*a):
        global nqp
        out = orig(*a)
        pb = orc.last_problem
        T = pb.T
        H, g, F, f, C, c, Phi, cv = ci.condense(pb)
        Phi = Phi[1:]                                        # (T, 3, 2T): s_{t+1} = Phi[t] u + c_{t+1}
        A, B = pb.A, pb.B                                    # (T,3,3), (T,3,2)
        a = np.stack([A[:, 0, 2], A[:, 1, 2]], axis=1)       # a_t
        assert np.allclose(A - np.eye(3)[None], np.pad(a[:, :, None], ((0, 0), (0, 1), (2, 0)))), "A_t = I + a_t e_2'"
        u = rng.standard_normal((T, 2)); q = rng.standard_normal((T, 3))
        # s = Phi u
        dense = np.einsum("tkc,c->tk", Phi, u.reshape(-1))
        th = np.cumsum(np.einsum("tj,tj->t", B[:, 2, :], u))
        thp = np.concatenate([[0.0], th[:-1]])
        xy = np.cumsum(a * thp[:, None] + np.einsum("tkj,tj->tk", B[:, :2, :], u), axis=0)
        scan = np.concatenate([xy, th[:, None]], axis=1)
        worst["phi"] = max(worst["phi"], np.abs(scan - dense).max() / max(1.0, np.abs(dense).max()))
        # w = Phi' q
        dense_t = np.einsum("tkc,tk->c", Phi, q)
        l01 = np.cumsum(q[::-1, :2], axis=0)[::-1]
        l01n = np.concatenate([l01[1:], np.zeros((1, 2))]); an = np.concatenate([a[1:], np.zeros((1, 2))])
        l2 = np.cumsum((q[:, 2] + np.einsum("tj,tj->t", an, l01n))[::-1])[::-1]
        lam = np.concatenate([l01, l2[:, None]], axis=1)
        w = np.einsum("tkj,tk->tj", B, lam).reshape(-1)
        worst["phit"] = max(worst["phit"], np.abs(w - dense_t).max() / max(1.0, np.abs(dense_t).max()))
        # P_t
        S = np.zeros((T, 3, 3))
        for t in range(T):
            v = rng.standard_normal((2, 3)); S[t, :2, :2] = v @ v.T * 10.0 ** rng.uniform(-6, 2)      # PSD xy blocks of very different size
        P = np.zeros((T, 3, 3)); acc = np.zeros((3, 3))
        for t in range(T - 1, -1, -1):
            if t < T - 1:
                At = A[t + 1]; acc = At.T @ acc @ At
            acc = acc + S[t]; P[t] = acc
        cc = np.cumsum(a, axis=0)                             # c_t
        Sxy = S[:, :2, :2]
        sS = np.cumsum(Sxy[::-1], axis=0)[::-1]
        Sc = np.einsum("tij,tj->ti", Sxy, cc); sSc = np.cumsum(Sc[::-1], axis=0)[::-1]
        cSc = np.einsum("ti,ti->t", cc, Sc); scSc = np.cumsum(cSc[::-1])[::-1]
        Pc = np.zeros((T, 3, 3))
        for t in range(T):
            Pc[t, :2, :2] = sS[t]
            v = sSc[t] - sS[t] @ cc[t]
            Pc[t, :2, 2] = v; Pc[t, 2, :2] = v
            Pc[t, 2, 2] = scSc[t] - 2.0 * cc[t] @ sSc[t] + cc[t] @ sS[t] @ cc[t]
        worst["P"] = max(worst["P"], max(np.abs(Pc[t] - P[t]).max() / max(1e-300, np.abs(P[t]).max()) for t in range(T)))
        nqp += 1
        return out
    orc.nrmp = hook
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
print("%s: %d QPs of %d scenes; largest relative deviation of the scan form from the dense / recursive one (fp64):" % (name, nqp, scenes))
print("   Phi u: %.1e   Phi' q: %.1e   P_t blocks (closed form with differences of prefix sums): %.1e" % (worst["phi"], worst["phit"], worst["P"]))
