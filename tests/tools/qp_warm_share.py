"""Where do the interior-point iterations of a benchmark step go, and what could an active-set iteration for the warm solves
save?  (CPU study; oracle/condensed_ipm.py is the QP kernel's method in numpy.)

bench.py resets the planner's state every step, so the K QPs of a scene run cold -> warm as the PAN iteration settles.  This
tool replays every QP the oracle's PAN loop produces on N scenes of a workload under the kernel's warm-start rules, once
with the gate the kernel ACTUALLY applies (previous solve converged; see the note in nrmp_qp.hip) and once with the gate
rounds 2 / 3 meant to apply (... and moved the controls by < 0.1), and reports per PAN iteration: solves by start code, mean
iterations; then the share of all iterations spent in warm-started solves -- the ceiling of what replacing them with a
1.2-factorisation active-set iteration (profiles/r03_qp_active_set_study.txt) could remove.

    python tests/tools/qp_warm_share.py [scenes per workload] [procs]      -> profiles/r04_qp_warm_share.txt"""
import os, sys
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
WORK = ("diff_1k_T10_K10", "acker_2k_T20_K15")


def job(arg):
    name, b = arg
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
    out = {}
    for gate in ("converged", "converged and step < 0.1"):
        rows, prev, prev_u, step = [], None, None, 9.0
        for pb in pbs:
            ok = prev is not None and prev["merit"] <= 1e-12 and (gate == "converged" or step < 0.1)
            s, u, d, info = ci.solve_condensed(pb, warm=prev["warm"] if ok else None)
            step = float(np.abs(u - prev_u).max()) if prev_u is not None else 9.0
            rows.append((info["iters_total"], info["warm_code"]))
            prev, prev_u = info, u
        out[gate] = rows
    return name, out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    with ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(job, [(w, b) for w in WORK for b in range(n)]))
    lines = [f"interior-point iterations of the K QPs of a forward call that starts from a cleared state (what every bench step does), {n} scenes per workload",
             "per PAN iteration k: solves by start code (0 cold, 1 warm start used, 2 / 3 warm attempt dropped at iteration 0 / 6, 4 repeated cold, 5 cold retry) x mean iterations"]
    for w in WORK:
        for gate in ("converged", "converged and step < 0.1"):
            rows = [out[gate] for name, out in res if name == w]
            K = len(rows[0])
            it = np.array([[r[k][0] for k in range(K)] for r in rows]); code = np.array([[r[k][1] for k in range(K)] for r in rows])
            lines.append(f"{w}   warm start attempted when the previous solve: {gate}" + ("   [the kernel's rule]" if gate == "converged" else "   [rounds 2 / 3 intended this]"))
            for k in range(K):
                by = " ".join(f"{c}:{(code[:, k] == c).sum()}x{it[:, k][code[:, k] == c].mean():.1f}" for c in range(6) if (code[:, k] == c).any())
                lines.append(f"   k={k:2d}  mean {it[:, k].mean():5.2f}  max {it[:, k].max():2d}   {by}")
            warm = it[code == 1].sum(); tot = it.sum()
            lines.append(f"   all QPs: mean {it.mean():.2f} iterations; warm-started solves: {(code == 1).mean() * 100:.0f} % of the solves, "
                         f"{warm / tot * 100:.0f} % of the iterations (mean {it[code == 1].mean():.1f} each)")
            # an active-set iteration that converges on 90 % of the warm solves after 1.2 factorisations of ~0.8 iteration-equivalents each
            saved = 0.9 * (it[code == 1] - 1.2 * 0.8).clip(min=0).sum()
            lines.append(f"   ceiling of an active-set iteration for the warm solves (90 % success, 1.2 factorisations x 0.8 iteration-equivalents): "
                         f"-{saved / tot * 100:.0f} % of the QP kernel's iterations")
    txt = "\n".join(lines)
    print(txt)
    with open(os.path.join(ROOT, "profiles", "r04_qp_warm_share.txt"), "w") as f:
        f.write(txt + "\n")


if __name__ == "__main__":
    main()
