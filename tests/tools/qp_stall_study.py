"""Solves that stop short: how many of the benchmark QPs end above 1e-13, and what the centring floor tied to the residual
(QP_SIGMA_MU_RES, csrc/nrmp_qp_device.h; SIGMA_MU_RES in oracle/condensed_ipm.py) does about it.  CPU replay of the kernel's
method (oracle/condensed_ipm.py) over every QP of a forward call from a cleared state, the kernel's warm-start rules included.

    python tests/tools/qp_stall_study.py [scenes per workload] [procs]      -> profiles/r05_qp_stall_study.txt"""
import os, sys
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
WORK = ("diff_1k_T10_K10", "acker_2k_T20_K15", "dyna_4k_T10_K10")


def job(arg):
    name, b, smr = arg
    from helpers import CONFIGS, make_oracle
    from neupan_amd.scenes import make_scene
    from oracle import condensed_ipm as ci
    ci.SIGMA_MU_RES = smr
    cfg = CONFIGS[name]
    sc = make_scene(cfg, b)
    orc = make_oracle(cfg)
    pbs = []
    orig = orc.nrmp

    def hook(*a):
        r = orig(*a); pbs.append(orc.last_problem); return r
    orc.nrmp = hook
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
    recs, prev = [], None
    for k, pb in enumerate(pbs):
        ok = prev is not None and prev["merit"] <= 1e-12
        s, u, d, info = ci.solve_condensed(pb, warm=prev["warm"] if ok else None)
        recs.append((k, info["warm_code"], info["iters_total"], info["merit"]))
        prev = info
    return name, smr, recs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    floors = (0.0, 0.003, 0.01, 0.1)
    with ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(job, [(w, b, f) for f in floors for w in WORK for b in range(n)]))
    lines = [f"final merit and iterations of every QP of a forward call ({n} scenes per workload; kernel's method, warm-start rules included)",
             "centring target: sigma mu >= max(1e-15, FLOOR x largest scaled residual); FLOOR = 0 is round 4's kernel, 0.01 is round 5's"]
    for f in floors:
        lines.append(f"FLOOR = {f}")
        for w in WORK:
            rows = [r for name, smr, recs in res if name == w and smr == f for r in recs]
            m = np.array([r[3] for r in rows]); it = np.array([r[2] for r in rows])
            lines.append(f"   {w:18s} solves {len(rows):5d}  ended above 1e-13: {(m > 1e-13).sum():3d}  above 1e-12: {(m > 1e-12).sum():3d}  above 1e-10: {(m > 1e-10).sum():3d}  "
                         f"worst {m.max():.1e} | iterations mean {it.mean():5.2f}  p99 {np.percentile(it, 99):3.0f}  max {it.max():3d}")
    txt = "\n".join(lines)
    print(txt)
    with open(os.path.join(ROOT, "profiles", "r05_qp_stall_study.txt"), "w") as fo:
        fo.write(txt + "\n")


if __name__ == "__main__":
    main()
