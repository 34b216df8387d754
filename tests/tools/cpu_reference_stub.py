"""CPU baseline in the form SURVEY.md section 8(d) names: the REFERENCE'S OWN Python for the point flow, DUNE, the parameter
build and the PAN loop (pan.py:109-147, imported unmodified from /root/reference under the inert stubs of
tests/golden/ref_stub_loader.py) with the oracle's fp64 interior-point solver substituted for the CvxpyLayer call (nrmp.py:144)
-- "reference code with substituted solver" -- timed on THIS host (the build container: /root/reference does not exist on the
GPU box, so this figure cannot ride in bench.py's line; it is committed as profiles/r06_cpu_reference_stub.json, labelled with
its host).  Workload: BASELINE configs[1] (diff robot, 1000 points, T = 10, K = 10, iter_threshold = 0), scenes 0 .. n-1.

  (i)  one scene stream, torch.set_num_threads(nproc)
  (ii) nproc worker processes x 1 thread over independent scenes (the fairest CPU throughput)

    python tests/tools/cpu_reference_stub.py [--scenes 32] [--out profiles/r06_cpu_reference_stub.json]
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
WORKLOAD = "diff_1k_T10_K10"


def _build(threads):
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = str(threads)
    import torch
    torch.set_num_threads(threads)
    import make_golden as mg                                   # imports the reference under stubs
    cfg = mg.CONFIGS[WORKLOAD]
    pan, _ = mg.build_pan(cfg, iter_num=cfg.iter_num)
    return mg, cfg, pan


def _plan(mg, cfg, pan, b):
    sc = mg.make_scene(cfg, b)
    t = mg.t
    t0 = time.perf_counter()
    pan(t(sc["nom_s"]), t(sc["nom_u"]), t(sc["ref_s"]), t(sc["ref_us"]), t(sc["points"]), t(sc["velocities"]))
    return time.perf_counter() - t0


def _worker(args):
    lo, hi = args
    mg, cfg, pan = _build(1)
    _plan(mg, cfg, pan, 10_000)                                 # imports, lazy initialisation: outside the clock
    return [_plan(mg, cfg, pan, b) for b in range(lo, hi)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_cpu_reference_stub.json"))
    a = ap.parse_args()
    nproc = os.cpu_count() or 1
    # (ii) first: worker processes must not inherit a parent that already holds torch's thread pool
    per = max(1, a.scenes // nproc)
    chunks = [(w * per, (w + 1) * per) for w in range(nproc)]
    with mp.get_context("spawn").Pool(nproc) as pool:
        pool.map(_worker, [(0, 0)] * nproc)                      # start + warm every worker
        t0 = time.perf_counter()
        times = pool.map(_worker, chunks)
        wall = time.perf_counter() - t0
    flat = sorted(x for ts in times for x in ts)
    # the wall above includes each worker's warm-up plan again; the rate is taken from the per-plan clocks instead
    rate_workers = nproc / (sum(flat) / len(flat))
    mg, cfg, pan = _build(nproc)
    _plan(mg, cfg, pan, 10_000)
    single = [_plan(mg, cfg, pan, b) for b in range(min(a.scenes, 16))]
    rec = {
        "what": "reference code (pan.py:109-147 under stubs) with the oracle QP substituted for CvxpyLayer (nrmp.py:144)",
        "workload": WORKLOAD, "iter_threshold": 0.0, "solver": "oracle/nrmp_qp.py (fp64 interior point)",
        "host": {"node": platform.node(), "cpu_count": nproc, "machine": platform.machine(),
                 "note": "the BUILD CONTAINER (8 vCPU), not the GPU box's host: /root/reference does not travel"},
        "single_stream": {"torch_threads": nproc, "plans": len(single), "seconds_per_plan_median": sorted(single)[len(single) // 2],
                          "plans_per_s": len(single) / sum(single)},
        "worker_processes": {"workers": nproc, "threads_each": 1, "plans": len(flat), "seconds_per_plan_median": flat[len(flat) // 2],
                             "plans_per_s": rate_workers, "wall_s_including_worker_warmup": wall},
    }
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
