"""GPU + CPU diagnostic: for every scene of a workload whose GPU control differs from the oracle's by more than
1e-4, how the ORACLE itself behaves there -- movement of its own PAN iteration at the last steps and how far its
output moves when every obstacle coordinate changes by one float32 ulp (the yardstick of DESIGN.md section 5).

    timeout 300 python tests/tools/parity_outliers.py [workload] [scenes] [workers]
"""
import os, sys
# the worker processes re-import this module: one thread each, or 64 workers x (host cores) BLAS/OpenMP threads
# oversubscribe the box by orders of magnitude (a run without these lines did not finish in 15 minutes)
for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import CONFIGS, make_oracle
from neupan_amd.scenes import make_batch, make_scene

W = sys.argv[1] if len(sys.argv) > 1 else "acker_2k_T20_K15"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
WORKERS = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = CONFIGS[W]


def _one_thread():
    import torch
    torch.set_num_threads(1)


def job(b):
    _one_thread()
    sc = make_scene(cfg, b)
    o = make_oracle(cfg)
    s, u, d = o.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
    hist = [float(np.linalg.norm(o.trace[i][1] - o.trace[i - 1][1])) for i in range(1, len(o.trace))]
    return b, u, hist


def job_ulp(b):
    _one_thread()
    sc = make_scene(cfg, b)
    pts = np.nextafter(sc["points"], np.float32(np.inf))
    o = make_oracle(cfg)
    s, u, d = o.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], pts, sc["velocities"])
    return b, u


if __name__ == "__main__":
    import torch
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    from gpu_helpers import make_gpu_pan
    pan = make_gpu_pan(cfg)
    batch = make_batch(cfg, 0, N)
    args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
    out = pan.forward_batch(*args, batch.get("velocities"))
    u_gpu = out["opt_u"].cpu().numpy().astype(np.float64)
    info = pan.last_qp_info()
    with ProcessPoolExecutor(WORKERS, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(job, range(N)))
        err = np.array([np.linalg.norm(u_gpu[b] - u) for b, u, _ in res])
        bad = [b for b in range(N) if err[b] > 1e-4]
        ulp = dict(ex.map(job_ulp, bad))
    print(f"{W}: {N} scenes, median {np.median(err):.2e}, {len(bad)} above 1e-4; key mode {pan.key_mode()}")
    for b in bad:
        _, u, hist = res[b]
        print(f"scene {b:3d}: gpu-vs-oracle {err[b]:.2e}  oracle 1-ulp self-sensitivity {np.linalg.norm(ulp[b] - u):.2e}  "
              f"oracle movement last 4 iterations {['%.1e' % h for h in hist[-4:]]}  gpu qp status {int(info[b, 3])}")
