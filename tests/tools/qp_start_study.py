"""CPU study for the interior-point start of the QP kernel (oracle/condensed_ipm.py is its transliteration): iteration
counts of the cold start as it is (u = 0) against a cold start at the NOMINAL controls the problem was linearised
around (clipped strictly inside the speed box), and of the warm start across PAN iterations with the kernel's gates,
on every QP the oracle's PAN loop produces for the first `scenes` scenes of a workload.

    python tests/tools/qp_start_study.py [workload] [scenes] [procs]
"""
import os, sys
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def job(arg):
    name, b = arg
    from helpers import CONFIGS, make_oracle
    from neupan_amd.scenes import make_scene
    from oracle import condensed_ipm as ci
    cfg = CONFIGS[name]
    sc = make_scene(cfg, b, None)
    orc = make_oracle(cfg)
    rows = []
    orig = orc.nrmp
    prev = {"warm": None, "u": None}

    def hook(nom_s, nom_u, *a):
        s, u, d = orig(nom_s, nom_u, *a)
        pb = orc.last_problem
        T = pb.T; nu = 2 * T
        t0 = []; ci.solve_condensed(pb, trace=t0)
        it_cold = t0[-1]["it"]
        # cold start at the nominal controls: u_t = nom_u[:, t] pulled 5 % inside the speed box, d mid-range,
        # unit multipliers, slacks max(c - Cx, 1)  (the warm path with floor 1 and unit multipliers is exactly that)
        sb = np.asarray(pb.speed_bound, dtype=float)
        un = np.clip(np.asarray(nom_u, dtype=float), -0.95 * sb[:, None], 0.95 * sb[:, None])
        x0 = np.zeros(nu + (0 if pb.no_obs else T))
        x0[:nu] = un.T.reshape(-1)
        if not pb.no_obs:
            x0[nu:] = 0.5 * (max(pb.d_min, 0.0) + pb.d_max)
        H, g, F, f, C, c, Phi, cv = ci.condense(pb)
        old = ci.WARM_DELTA
        ci.WARM_DELTA = 1.0
        t1 = []; ci.solve_condensed(pb, trace=t1, warm=(x0, np.ones(C.shape[0]), np.ones(F.shape[0])))
        ci.WARM_DELTA = old
        it_nom = t1[-1]["it"]
        du = float(np.abs(t1[-1]["x"][:nu] - t0[-1]["x"][:nu]).max())
        rows.append(dict(k=len(rows), it_cold=it_cold, it_nom=it_nom, du=du, m_cold=t0[-1]["merit"], m_nom=t1[-1]["merit"]))
        return s, u, d
    orc.nrmp = hook
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
    return rows


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "diff_1k_T10_K10"
    scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    with ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(job, [(name, b) for b in range(scenes)]))
    rows = [r for rs in res for r in rs]
    a = lambda k, sel=None: np.array([r[k] for r in rows if sel is None or sel(r)], dtype=float)
    print("%s, %d scenes, %d QPs (transliteration of the kernel's method, tol 1e-14)" % (name, scenes, len(rows)))
    for tag, sel in (("all PAN iterations", None), ("first PAN iteration (always a cold start in the kernel)", lambda r: r["k"] == 0),
                     ("later PAN iterations", lambda r: r["k"] > 0)):
        c, n = a("it_cold", sel), a("it_nom", sel)
        print("  %-58s cold start at u = 0: mean %.2f max %d | at the nominal controls: mean %.2f max %d | fewer %d equal %d more %d"
              % (tag, c.mean(), c.max(), n.mean(), n.max(), (n < c).sum(), (n == c).sum(), (n > c).sum()))
    print("  largest |du| between the two starts' solutions: %.2e; final merits max %.1e / %.1e" % (a("du").max(), a("m_cold").max(), a("m_nom").max()))
