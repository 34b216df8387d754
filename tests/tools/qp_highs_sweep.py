"""The oracle's QP solve (oracle/nrmp_qp.py, the stand-in for cvxpylayers -> ECOS at neupan/blocks/nrmp.py:144) against
HiGHS on EVERY QP of whole benchmark batches: all K solves of the first `scenes` scenes of each workload, as the oracle's
own PAN loop produces them (CPU only, multi-process).  Writes the summary to profiles/r02_qp_highs.json.

    python tests/tools/qp_highs_sweep.py [procs]

Per QP: max |u_oracle - u_HiGHS|, objective difference (oracle - HiGHS; <= 0 means the oracle's point is at least as
good), the oracle's KKT certificate, HiGHS' status.  HiGHS stops at ~1e-7 objective accuracy, which along the flat
steering directions of this QP is 1e-6 .. 1e-4 in u: the objective difference is the sharper statement."""
import json, os, sys, time
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):      # before numpy loads: one BLAS thread per worker
    os.environ.setdefault(_k, "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

WORK = [("diff_1k_T10_K10", 256, {}), ("acker_2k_T20_K15", 24, {}), ("dyna_4k_T10_K10", 24, {}),
        ("diff_1k_T10_K10", 16, dict(omni=True))]
OMNI = dict(kinematics="omni", length=1.6, width=2.0, max_speed=[8, 6.28], max_acce=[3, 3])


def job(arg):
    name, b, opt = arg
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = "1"
    from helpers import CONFIGS, make_oracle
    from neupan_amd.scenes import make_scene
    from oracle import pan_oracle as po
    from oracle.nrmp_qp import kkt_certificate
    from qp_highs import compare_with_highs
    cfg = CONFIGS[name]
    sc = make_scene(cfg, b, 300 if opt.get("omni") else None)
    orc = make_oracle(cfg, robot_kw=OMNI if opt.get("omni") else None, **(dict(dune_max_num=300, iter_num=4) if opt.get("omni") else {}))
    out = []
    orig = orc.nrmp

    def hook(*a):
        s, u, d = orig(*a)
        pb = orc.last_problem
        from oracle.nrmp_qp import solve_nrmp_qp
        s64, u64, d64 = solve_nrmp_qp(pb)            # the fp64 solution (orig returns the fp32 cast)
        r = compare_with_highs(pb, s64, u64, d64)
        c = kkt_certificate(pb, s64, u64, d64)
        out.append(dict(du=r["du"], obj_diff=r["obj_diff"], rel=r["obj_diff"] / max(1.0, abs(r["obj"])), status=r["status"],
                        stat=c["stat"], comp=c["comp"], feas=c["feas"], dyn=c["dyn"]))
        return s, u, d
    orc.nrmp = hook
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
    return name + ("_omni" if opt.get("omni") else ""), out


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    jobs = [(n, b, o) for n, cnt, o in WORK for b in range(cnt)]
    t0 = time.time()
    with ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(job, jobs, chunksize=2))
    by = {}
    for name, out in res:
        by.setdefault(name, []).extend(out)
    summ = {}
    for name, rows in by.items():
        du = np.array([r["du"] for r in rows]); od = np.array([r["obj_diff"] for r in rows]); rel = np.array([r["rel"] for r in rows])
        summ[name] = {"qps": len(rows), "highs_optimal": int(sum(r["status"] == "Optimal" for r in rows)),
                      "du_median": float(np.median(du)), "du_p99": float(np.quantile(du, 0.99)), "du_max": float(du.max()),
                      "obj_diff_max": float(od.max()), "obj_diff_rel_max": float(rel.max()), "oracle_not_worse_1e-9_rel": int((rel <= 1e-9).sum()),
                      "kkt_stat_max": float(max(r["stat"] for r in rows)), "kkt_comp_max": float(max(r["comp"] for r in rows)),
                      "kkt_feas_max": float(max(r["feas"] for r in rows)), "kkt_dyn_max": float(max(r["dyn"] for r in rows))}
    out = {"what": "oracle/nrmp_qp.py (fp64 IPM, tol 1e-14) vs HiGHS (scipy's bundled QP solver, feasibility tolerances 1e-10) on every QP "
                   "of the oracle's PAN loop over whole benchmark batches; obj_diff = objective(oracle) - objective(HiGHS)",
           "seconds": round(time.time() - t0, 1), "workloads": summ}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_qp_highs.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
