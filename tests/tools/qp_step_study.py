"""CPU study behind the interior-point heuristics of the QP kernel (nrmp_qp.hip: QP_STEP_ETA / QP_STEP_CAP, QP_START_MU,
QP_SIGMA_MU_MIN, QP_WARM_DELTA, the warm-start drop rules).  oracle/condensed_ipm.py is the kernel's method in numpy and
carries the same constants; this tool replays it over EVERY QP the oracle's PAN loop produces on scenes of the four
benchmark workloads -- warm-started along the loop under the kernel's gate (previous solve converged to 1e-12) -- once
per rule set, and counts interior-point iterations.

    python tests/tools/qp_step_study.py [scenes per workload] [procs]     -> profiles/r03_qp_step_study.txt
    python tests/tools/qp_step_study.py --trace                            -> profiles/r03_qp_tail_trajectories.txt
                                  (the slowest cold solves of configs[1], iteration by iteration: why they are slow)

The kernel's time is proportional to these counts (one wave per scene, ~2 900 VALU instructions per iteration)."""
import os, sys
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

WORK = ("diff_1k_T10_K10", "acker_2k_T20_K15", "dyna_4k_T10_K10", "poly8_5k_T10_K10")
ROUND2 = dict(STALL_FAR=3, CENTRAL_GAMMA=0.0, STEP_CAP=0.005, START_MU=None, SIGMA_MU_MIN=0.0, WARM_DELTA=0.01, WARM_DROP={0: 0.05, 3: 1e-4, 7: 1e-8}, RETRY_MERIT=float("inf"))
SHIPPED = {}
ROUND5 = {"STALL_FAR": 3, "CENTRAL_GAMMA": 0.0}
RULES_STALL = [("shipped in round 5 (the best iterate stands after 3 non-improving iterations, anywhere)", ROUND5),
               ("patience 6 while the best merit is above 1e-6", {"STALL_FAR": 6, "CENTRAL_GAMMA": 0.0}),
               ("patience 8 while the best merit is above 1e-6", {"STALL_FAR": 8, "CENTRAL_GAMMA": 0.0}),
               ("patience 12 while the best merit is above 1e-6", {"STALL_FAR": 12, "CENTRAL_GAMMA": 0.0}),
               ("no stall rule while the best merit is above 1e-6", {"STALL_FAR": 1000, "CENTRAL_GAMMA": 0.0})]
RULES_CENTRAL = [("shipped in round 5 (rule of three anywhere, no safeguard)", ROUND5),
                 ("patience 8 far from convergence", {"STALL_FAR": 8, "CENTRAL_GAMMA": 0.0}),
                 ("+ centrality 1e-3 x0.7 x6, every step", {"STALL_FAR": 8, "CENTRAL_GAMMA": 1e-3, "CENTRAL_ALPHA": 2.0}),
                 ("+ centrality 1e-3 x0.7 x6, steps below 0.9  = SHIPPED in round 6", {"STALL_FAR": 8, "CENTRAL_GAMMA": 1e-3, "CENTRAL_ALPHA": 0.9}),
                 ("+ centrality 1e-3 x0.7 x6, steps below 0.5", {"STALL_FAR": 8, "CENTRAL_GAMMA": 1e-3, "CENTRAL_ALPHA": 0.5}),
                 ("+ centrality 1e-4 x0.7 x6, steps below 0.9", {"STALL_FAR": 8, "CENTRAL_GAMMA": 1e-4, "CENTRAL_ALPHA": 0.9}),
                 ("+ centrality 1e-2 x0.7 x6, steps below 0.9", {"STALL_FAR": 8, "CENTRAL_GAMMA": 1e-2, "CENTRAL_ALPHA": 0.9})]
RULES_CORR = [("shipped before (stall 8, centrality 1e-3 below 0.9, corrector always)", {"CORRECTOR_MIN_AFF": 0.0}),
              ("no corrector when the affine step is below 0.05", {"CORRECTOR_MIN_AFF": 0.05}),
              ("no corrector when the affine step is below 0.1  = SHIPPED", {"CORRECTOR_MIN_AFF": 0.1}),
              ("no corrector when the affine step is below 0.2", {"CORRECTOR_MIN_AFF": 0.2}),
              ("no corrector when the affine step is below 0.3", {"CORRECTOR_MIN_AFF": 0.3})]
RULES = [("round 2 (eta = 0.995, unit multipliers, floor 0.01, drops at 0 / 3 / 7)", ROUND2),
         ("+ centring target >= 1e-15", {**ROUND2, "SIGMA_MU_MIN": 1e-15}),
         ("+ eta = max(0.995, 1 - mu) <= 1 - 1e-6", {**ROUND2, "SIGMA_MU_MIN": 1e-15, "STEP_CAP": 1e-6}),
         ("+ multipliers 3 / slack at the cold start", {**ROUND2, "SIGMA_MU_MIN": 1e-15, "STEP_CAP": 1e-6, "START_MU": 3.0}),
         ("+ warm floor 0.003", {**ROUND2, "SIGMA_MU_MIN": 1e-15, "STEP_CAP": 1e-6, "START_MU": 3.0, "WARM_DELTA": 0.003}),
         ("+ drops at 0 / 6 only", {"RETRY_MERIT": float("inf")}),
         ("+ a cold solve that ends above 1e-9 is repeated from unit multipliers  = SHIPPED", SHIPPED),
         ("shipped, but no cap on eta", {"STEP_CAP": 0.0}),
         ("shipped, but warm floor 0.001", {"WARM_DELTA": 0.001}),
         ("shipped, but warm floor 0.01", {"WARM_DELTA": 0.01}),
         ("shipped, but no drop after iteration 0", {"WARM_DROP": {0: 0.05}}),
         ("shipped, but no floor on the centring target", {"SIGMA_MU_MIN": 0.0})]


def job(arg):
    name, b, stall = arg if len(arg) == 3 else (*arg, 0)
    rules_ = {0: RULES, 1: RULES_STALL, 2: RULES_CENTRAL, 3: RULES_CORR}[int(stall)]
    from helpers import CONFIGS, make_oracle
    from neupan_amd.scenes import make_scene
    from oracle import condensed_ipm as ci
    cfg = CONFIGS[name]
    sc = make_scene(cfg, b)
    orc = make_oracle(cfg)
    pbs = []
    orig = orc.nrmp

    def hook(*a):
        r = orig(*a)
        pbs.append(orc.last_problem)
        return r
    orc.nrmp = hook
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
    shipped = {k: getattr(ci, k) for k in ("STEP_ETA", "STEP_CAP", "START_MU", "SIGMA_MU_MIN", "WARM_DELTA", "WARM_DROP", "RETRY_MERIT", "STALL_FAR", "CENTRAL_GAMMA", "CENTRAL_ALPHA", "CORRECTOR_MIN_AFF")}
    out = []
    for _, rules in rules_:
        for k, v in {**shipped, **rules}.items():
            setattr(ci, k, v)
        rows = []; prev = None; prev_u = None
        for pb in pbs:
            warm = prev["warm"] if prev is not None and prev["merit"] <= 1e-12 else None      # (the kernel's gate, see nrmp_qp.hip)
            s, u, d, info = ci.solve_condensed(pb, warm=warm)
            info["step"] = float(np.abs(u - prev_u).max()) if prev_u is not None else 9.0     # (first nominal: not a solve's output)
            rows.append((info["iters_total"], info["warm_code"], float(info["merit"])))
            prev, prev_u = info, u
        out.append(rows)
    for k, v in shipped.items():
        setattr(ci, k, v)
    return name, out


def trace_job(b):
    from helpers import CONFIGS, make_oracle
    from neupan_amd.scenes import make_scene
    from oracle import condensed_ipm as ci
    cfg = CONFIGS["diff_1k_T10_K10"]
    sc = make_scene(cfg, b)
    orc = make_oracle(cfg)
    pbs = []
    orig = orc.nrmp

    def hook(*a):
        r = orig(*a)
        pbs.append(orc.last_problem)
        return r
    orc.nrmp = hook
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
    out = []
    for k, pb in enumerate(pbs):
        tr = []
        s, u, d, info = ci.solve_condensed(pb, trace=tr)
        if info["iters_total"] >= 17:
            nu = 2 * pb.T
            xf = tr[info["iters"]]["x"][:nu]
            out.append((b, k, info["iters_total"], [(t["it"], t["merit"], t["mu"], float(np.abs(t["x"][:nu] - xf).max())) for t in tr]))
    return out


def trace_main():
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    with ProcessPoolExecutor(os.cpu_count() or 1, mp_context=mp.get_context("spawn")) as ex:
        res = [r for rs in ex.map(trace_job, range(32)) for r in rs]
    lines = ["cold solves of configs[1] (32 scenes, every PAN iteration) that take >= 17 interior-point iterations: merit (max of the scaled",
             "residuals and mu), mu, and the distance of the controls from the solve's final point, per iteration.  In the tail mu falls by",
             "about x0.13 per iteration WITH full steps (no strict complementarity: the second-order term of the complementarity products is not",
             "negligible), and |u - u*| follows sqrt(mu): stopping at mu = 1e-10 would leave the controls 1e-5 from their limit."]
    for b, k, n, tr in res[:6]:
        lines.append(f"scene {b}, PAN iteration {k}: {n} iterations")
        for it, merit, mu, du in tr:
            lines.append(f"   it {it:2d}  merit {merit:8.1e}  mu {mu:8.1e}  |u - u*| {du:8.1e}")
    txt = "\n".join(lines)
    print(txt)
    with open(os.path.join(ROOT, "profiles", "r03_qp_tail_trajectories.txt"), "w") as f:
        f.write(txt + "\n")


def main():
    global RULES
    if "--trace" in sys.argv:
        return trace_main()
    if "--stall" in sys.argv:                 # (round 6: the patience of the stall rule far from convergence -> profiles/r06_qp_stall_patience.txt)
        RULES = RULES_STALL
        sys.argv.remove("--stall")
    if "--central" in sys.argv:               # (round 6: the centrality safeguard of the step -> profiles/r06_qp_centrality.txt)
        RULES = RULES_CENTRAL
        sys.argv.remove("--central")
    if "--corrector" in sys.argv:             # (round 6: no second-order corrector on a short affine step -> profiles/r06_qp_corrector.txt)
        RULES = RULES_CORR
        sys.argv.remove("--corrector")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    with ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(job, [(w, b, 1 if RULES is RULES_STALL else (2 if RULES is RULES_CENTRAL else (3 if RULES is RULES_CORR else 0))) for w in WORK for b in range(n)]))
    lines = [f"interior-point iterations per QP over the oracle's PAN loop, {n} scenes per workload, every QP of every scene (tests/tools/qp_step_study.py)",
             "columns: mean / max over all QPs | mean / max over the LAST QP of a call | solves by warm code (0 cold, 1 warm used, 2 / 3 dropped at the first / a later checkpoint, 4 repeated cold, 5 cold retry) | solves that end above 1e-12, largest final merit"]
    for w in WORK:
        lines.append(w)
        for i, (label, _) in enumerate(RULES):
            rows = [r for name, out in res if name == w for r in out[i]]
            last = [out[i][-1][0] for name, out in res if name == w]
            it = np.array([r[0] for r in rows]); code = np.array([r[1] for r in rows]); bad = sum(r[2] > 1e-12 for r in rows)
            by = " ".join(f"{c}:{(code == c).sum()}x{it[code == c].mean():.1f}" for c in range(6) if (code == c).any())
            lines.append(f"  {label:82s} {it.mean():6.2f} /{it.max():3d} | {np.mean(last):6.2f} /{max(last):3d} | {by} | {bad}, {max(r[2] for r in rows):.1e}")
    txt = "\n".join(lines)
    print(txt)
    with open(os.path.join(ROOT, "profiles", "r06_qp_stall_patience.txt" if RULES is RULES_STALL else ("r06_qp_centrality.txt" if RULES is RULES_CENTRAL else ("r06_qp_corrector.txt" if RULES is RULES_CORR else "r03_qp_step_study.txt"))), "w") as f:
        f.write(txt + "\n")


if __name__ == "__main__":
    main()
