"""Per-scene control L2 (HIP path vs oracle) as a function of the number of PAN iterations:
shows whether a large final deviation is a solver error or amplification by a non-contracting
PAN fixed-point iteration (grows from ~1e-7 by a constant factor per iteration)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_helpers import make_gpu_pan
from helpers import CONFIGS, make_oracle
from neupan_amd.scenes import make_batch
cfg = CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "diff_1k_T10_K10"]; B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
if len(sys.argv) > 3:
    import dataclasses
    cfg = dataclasses.replace(cfg, n_points=int(sys.argv[3]))
batch = make_batch(cfg, 0, B)
args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
orc_u = []
for b in range(B):
    o = make_oracle(cfg)
    o.forward(*[a[b] for a in args], None if batch['velocities'] is None else batch['velocities'][b])
    orc_u.append([t[1] for t in o.trace])
errs = np.zeros((B, cfg.iter_num))
for K in range(1, cfg.iter_num + 1):
    pan = make_gpu_pan(cfg, iter_num=K)
    u = pan.forward_batch(*args, batch["velocities"])["opt_u"].cpu().numpy()
    info = pan.last_qp_info()
    if (info[:, 3] != 0).any() or info[:, 1].max() > 1e-8:
        print("K", K, "QP status", info[:, 3], "merit", info[:, 1])
    for b in range(B):
        errs[b, K - 1] = np.linalg.norm(u[b].astype(np.float64) - orc_u[b][K - 1])
np.set_printoptions(linewidth=200, formatter={"float": lambda v: "%.1e" % v})
worst = np.argsort(-errs[:, -1])[:6]
for b in worst:
    du = [float(np.linalg.norm(orc_u[b][k] - orc_u[b][k - 1])) for k in range(1, cfg.iter_num)]
    print("scene", b, "L2 by K:", errs[b], " oracle |u_k - u_{k-1}|:", np.array(du))
print("median final", np.median(errs[:, -1]), "frac<=1e-4", (errs[:, -1] <= 1e-4).mean())
