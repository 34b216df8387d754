"""GPU diagnostics of the active-set iteration inside the QP kernel (NPA_QP_ASET=1): per PAN iteration, how the solves started /
ended (qp_info[15], [5..7]), iterations, and the distance of the controls from the default interior-point path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_helpers import make_gpu_pan
from helpers import CONFIGS
from neupan_amd.scenes import make_batch

name = sys.argv[1] if len(sys.argv) > 1 else "diff_1k_T10_K10"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = CONFIGS[name]
batch = make_batch(cfg, 0, B)
args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")] + [batch.get("velocities")]
ref = make_gpu_pan(cfg)
r0 = ref.forward_batch_trace(*args)
os.environ["NPA_QP_ASET"] = "1"
pan = make_gpu_pan(cfg)
r1 = pan.forward_batch_trace(*args)
u0, u1 = r0["trace_u"].cpu().numpy().astype(np.float64), r1["trace_u"].cpu().numpy().astype(np.float64)
q0, q1 = r0["trace_qp_info"].cpu().numpy(), r1["trace_qp_info"].cpu().numpy()
K = u0.shape[1]
for k in range(K):
    du = np.sqrt(((u1[:, k] - u0[:, k]) ** 2).sum(axis=(1, 2)))
    code = q1[:, k, 15].astype(int); why = q1[:, k, 7].astype(int); gs = q1[:, k, 5].astype(int)
    print(f"k={k}: its ipm {q0[:, k, 14].mean():5.2f} (max {q0[:, k, 14].max():.0f})  aset-build {q1[:, k, 14].mean():5.2f} (max {q1[:, k, 14].max():.0f}) | "
          f"codes {np.bincount(code, minlength=7).tolist()} why {np.bincount(why, minlength=5).tolist()} guesses {np.bincount(gs, minlength=6).tolist()} | "
          f"status!=0 {(q1[:, k, 3] != 0).sum()} merit max {q1[:, k, 1].max():.1e} | |du| median {np.median(du):.1e} p90 {np.quantile(du, .9):.1e} max {du.max():.1e}"
          f" | left(why=2) med {np.median(q1[:, k, 6][why == 2]) if (why == 2).any() else 0:.1e}")
w = q1[:, 1:, 7].astype(int); fm = q1[:, 1:, 8]
for c, nm in ((1, "accepted"), (2, "repeated guess, residual left"), (3, "guesses used up")):
    if (w == c).any():
        v = fm[w == c]
        print(f"first-pass merit of the seeded point, {nm}: n {v.size} median {np.median(v):.2e} p10 {np.quantile(v, .1):.2e} p90 {np.quantile(v, .9):.2e} max {v.max():.2e}")
taken = q1[:, :, 15] == 6
print("taken overall: %.3f of all solves; iterations of taken solves: mean %.2f" % (taken.mean(), q1[:, :, 14][taken].mean() if taken.any() else 0))
# timing of the two builds on this batch
for tag, p in (("ipm", ref), ("aset", pan)):
    torch.cuda.synchronize(); import time
    for _ in range(3): p.reset_stop_state(); p.forward_batch(*args)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): p.reset_stop_state(); p.forward_batch(*args)
    torch.cuda.synchronize(); print(tag, "forward_batch %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))
