"""Iteration statistics of the QP solves along the PAN iterations of one forward call."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_helpers import make_gpu_pan
from helpers import CONFIGS
from neupan_amd.scenes import make_batch
cfg = CONFIGS["diff_1k_T10_K10"]; B = 256
batch = make_batch(cfg, 0, B)
args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
for K in (1, 2, 3, 5, 10):
    pan = make_gpu_pan(cfg, iter_num=K)
    pan.forward_batch(*args)
    info = pan.last_qp_info()
    print("K=%2d last-QP iterations: mean %.2f max %d  status!=0: %d  merit max %.1e" %
          (K, info[:, 4].mean(), info[:, 4].max(), (info[:, 3] != 0).sum(), info[:, 1].max()))
pan = make_gpu_pan(cfg, iter_num=10)
pan.forward_batch(*args)
info = pan.last_qp_info()
it = info[:, 4].astype(int); best = info[:, 0].astype(int)
print("iterations histogram:", np.bincount(it).tolist())
print("best-iterate index histogram:", np.bincount(best).tolist())
print("iters - best:", np.bincount(it - best).tolist())
print("status:", np.bincount(info[:, 3].astype(int)).tolist(), " merit pct50/90/100:", np.percentile(info[:, 1], [50, 90, 100]))
tot = info[:, 14].astype(int); code = info[:, 15].astype(int)
print("total iterations incl. dropped warm attempts: mean %.2f max %d; warm code histogram (0 cold, 1 warm used, 2/3 dropped at it 0/3, 4 not converged):" % (tot.mean(), tot.max()), np.bincount(code, minlength=5).tolist())
for c in range(5):
    if (code == c).any(): print("  code", c, "total iterations: mean %.1f max %d" % (tot[code == c].mean(), tot[code == c].max()))
