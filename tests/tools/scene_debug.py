"""Debugging aid of the scene kernel (csrc/pan_scene.hip): the two-launch path against NPA_SCENE_KERNEL=1 on one batch, printing
where they differ instead of asserting.  `--lib=<path>` loads a variant library, `--k=<n>` limits the PAN iterations, `--cfg=<name>`
picks the configuration (default diff_1k_T10_K10), `--b=<n>` the batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import neupan_amd._lib as L
for a in sys.argv[1:]:
    if a.startswith("--lib="):
        L.LIB_PATH = a[6:]
K = next((int(a[4:]) for a in sys.argv[1:] if a.startswith("--k=")), 0)
B = next((int(a[4:]) for a in sys.argv[1:] if a.startswith("--b=")), 96)
from gpu_helpers import make_gpu_pan
from helpers import CONFIGS
from neupan_amd.scenes import make_batch
cfg = CONFIGS[next((a[6:] for a in sys.argv[1:] if a.startswith("--cfg=")), "diff_1k_T10_K10")]
over = dict(iter_num=K) if K else {}
two = make_gpu_pan(cfg, **over)
os.environ["NPA_SCENE_KERNEL"] = "1"
one = make_gpu_pan(cfg, **over)
del os.environ["NPA_SCENE_KERNEL"]
batch = make_batch(cfg, 12000, B)
args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
print("lib", L.LIB_PATH, "K", two.iter_num, "B", B, flush=True)
for rep in range(3):
    a = two.forward_batch(*args)
    b = one.forward_batch(*args)
    import torch
    torch.cuda.synchronize()
    for k in ("opt_s", "opt_u", "opt_d", "min_distance", "iters", "nrmp_points"):
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        bad = ~np.isclose(x, y, rtol=0, atol=0, equal_nan=True)
        print(" call", rep, k, "equal" if not bad.any() else "DIFFERS in %d of %d entries, scenes %s, max |d| %.3g" %
              (bad.sum(), bad.size, np.unique(np.nonzero(bad)[0])[:12].tolist(), np.nanmax(np.abs(x.astype(np.float64) - y))), flush=True)
    qa, qb = two.last_qp_info(), one.last_qp_info()
    print(" call", rep, "qp_info[:, :5] equal:", np.array_equal(qa[:, :5], qb[:, :5]), "status", np.bincount(qb[:, 3].astype(int)).tolist(), flush=True)
print("done", flush=True)
