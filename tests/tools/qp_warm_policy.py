"""Which rule should abandon a warm-started interior-point solve?  (CPU study; oracle/condensed_ipm.py is the QP kernel's method.)

A launch of the QP kernel lasts as long as its slowest scene, and the slowest scenes are the warm attempts that are dropped late
(3 or 6 iterations spent, then a full cold solve: 18 - 25 iterations against a mean of 6 - 8, profiles/r04_qp_warm_share.txt).
This tool records, for every QP the oracle's PAN loop produces on N scenes of a workload, the merit trajectory of the warm-
started solve run to its end WITHOUT any drop rule, and the cost of the cold solve of the same QP; drop rules are then
evaluated offline on those records: mean iterations per solve and the mean over "launches" (random groups of 256 / 1280
solves of one PAN iteration) of the per-launch maximum.

    python tests/tools/qp_warm_policy.py [scenes per workload] [procs]      -> profiles/r05_qp_warm_policy.txt"""
import os, sys, json
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
WORK = ("diff_1k_T10_K10", "acker_2k_T20_K15", "dyna_4k_T10_K10")


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
    recs, prev = [], None
    saved = dict(ci.WARM_DROP)
    for k, pb in enumerate(pbs):
        tr_c = []
        s, u, d, info_c = ci.solve_condensed(pb, trace=tr_c)
        rec = {"k": k, "cold": info_c["iters_total"], "warm": None}
        if prev is not None and prev["merit"] <= 1e-12:
            ci.WARM_DROP.clear()                      # no drop rule: the warm attempt runs to its own end
            tr = []
            try:
                s2, u2, d2, info_w = ci.solve_condensed(pb, warm=prev["warm"], trace=tr)
            finally:
                ci.WARM_DROP.update(saved)
            n_warm = next((i for i, t in enumerate(tr) if t["it"] == 0 and i > 0), len(tr))      # (a failed attempt is followed by the cold trace)
            mer = [t["merit"] for t in tr[:n_warm]]
            rec["warm"] = {"merit": mer, "ok": info_w["warm_code"] == 1, "iters": n_warm - 1}
        recs.append(rec)
        prev = info_c           # (the next solve is warm-started from THIS solve's solution, whichever path reached it)
    return name, recs


def evaluate(recs_by_k, rule, rng, launch=256, trials=200):
    """rule: dict it -> threshold (drop when merit_it > threshold).  Returns mean iterations per solve, mean per-launch max."""
    per_k_cost = []
    for k, recs in sorted(recs_by_k.items()):
        cost = []
        for r in recs:
            w = r["warm"]
            if w is None:
                cost.append(r["cold"]); continue
            c = None
            for i, m in enumerate(w["merit"]):
                if i in rule and m > rule[i]:
                    c = i + r["cold"]; break
            if c is None:
                c = w["iters"] if w["ok"] else w["iters"] + r["cold"]
            cost.append(c)
        per_k_cost.append(np.array(cost))
    mean = float(np.mean(np.concatenate(per_k_cost)))
    mx = []
    for c in per_k_cost:
        mx.append(np.mean([rng.choice(c, size=launch, replace=True).max() for _ in range(trials)]))
    return mean, float(np.sum(mx)), [float(v) for v in mx]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    cache = f"/tmp/qp_warm_policy_{n}.json"
    if os.path.exists(cache):
        res = json.load(open(cache))
    else:
        from concurrent.futures import ProcessPoolExecutor
        import multiprocessing as mp
        with ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as ex:
            res = list(ex.map(job, [(w, b) for w in WORK for b in range(n)]))
        json.dump(res, open(cache, "w"))
    rng = np.random.default_rng(0)
    lines = [f"drop rules for the warm-started interior-point solve, evaluated on recorded merit trajectories ({n} scenes per workload, every QP of a forward call from a cleared state)",
             "per rule: mean iterations per solve | sum over the K launches of a forward call of the expected per-launch maximum (256-scene / 1280-scene launches)"]
    rules = {
        "kernel (round 4): it0 > 0.05, it3 > 3e-3, it6 > 1e-4": {0: 0.05, 3: 3e-3, 6: 1e-4},
        "none (warm runs to its end)": {},
        "it0 > 0.05 only": {0: 0.05},
        "it0 > 0.05, it2 > 1e-2": {0: 0.05, 2: 1e-2},
        "it0 > 0.05, it2 > 3e-3": {0: 0.05, 2: 3e-3},
        "it0 > 0.05, it1 > 2e-2, it2 > 3e-3": {0: 0.05, 1: 2e-2, 2: 3e-3},
        "it0 > 0.02, it2 > 3e-3": {0: 0.02, 2: 3e-3},
        "it0 > 0.02, it1 > 1e-2, it2 > 1e-3": {0: 0.02, 1: 1e-2, 2: 1e-3},
        "it0 > 0.01": {0: 0.01},
        "it0 > 0.01, it2 > 1e-3": {0: 0.01, 2: 1e-3},
        "it0 > 0.005, it2 > 1e-3": {0: 0.005, 2: 1e-3},
        "it0 > 0.05, it2 > 3e-3, it4 > 1e-4": {0: 0.05, 2: 3e-3, 4: 1e-4},
        "it0 > 0.02, it2 > 3e-3, it4 > 1e-5": {0: 0.02, 2: 3e-3, 4: 1e-5},
    }
    for w in WORK:
        by_k = {}
        for name, recs in res:
            if name == w:
                for r in recs:
                    by_k.setdefault(r["k"], []).append(r)
        lines.append(w)
        # how the warm attempts end without any rule
        allw = [r["warm"] for rs in by_k.values() for r in rs if r["warm"] is not None]
        okw = [x for x in allw if x["ok"]]
        lines.append(f"   warm attempts {len(allw)}, converged {len(okw)} ({100 * len(okw) / max(len(allw), 1):.1f} %), iterations of the converged ones: "
                     f"mean {np.mean([x['iters'] for x in okw]):.2f} p90 {np.percentile([x['iters'] for x in okw], 90):.0f} max {max(x['iters'] for x in okw)}; "
                     f"of the failed ones: mean {np.mean([x['iters'] for x in allw if not x['ok']] or [0]):.1f}; cold solves: mean "
                     f"{np.mean([r['cold'] for rs in by_k.values() for r in rs]):.2f} max {max(r['cold'] for rs in by_k.values() for r in rs)}")
        for tag, rule in rules.items():
            m, s256, _ = evaluate(by_k, rule, rng, 256)
            _, s1280, _ = evaluate(by_k, rule, rng, 1280)
            lines.append(f"   {tag:58s} mean {m:5.2f} | launch max summed over K: {s256:6.1f} / {s1280:6.1f}")
    txt = "\n".join(lines)
    print(txt)
    with open(os.path.join(ROOT, "profiles", "r05_qp_warm_policy.txt"), "w") as f:
        f.write(txt + "\n")


if __name__ == "__main__":
    main()
