"""Every QP of a traced forward call on many scenes: which (scene, PAN iteration) ended with status != 0 or above 1e-9.

    python tests/tools/qp_status_scan.py <workload> <scenes> [first scene] [robot: omni | polygon] [T]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import CONFIGS
from gpu_helpers import make_gpu_pan
from neupan_amd.scenes import make_batch

wl, n = sys.argv[1], int(sys.argv[2])
s0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = CONFIGS[wl]
robot = sys.argv[4] if len(sys.argv) > 4 else ""
kw = {}
if robot == "omni":
    kw = dict(robot_kw=dict(kinematics="omni", length=1.6, width=2.0, max_speed=[8, 6.28], max_acce=[3, 3]))
elif robot == "polygon":
    kw = dict(robot_kw=dict(kinematics="diff", vertices=[[-0.8, -1.0], [-1.8, 1.0], [1.8, 1.0], [0.8, -1.0]], max_speed=[8, 3], max_acce=[8, 3]))
    from helpers import ckpt_path
    kw["checkpoint"] = ckpt_path("polygon_robot")
if len(sys.argv) > 5:
    import dataclasses
    cfg = dataclasses.replace(cfg, T=int(sys.argv[5]))
pan = make_gpu_pan(cfg, **kw)
b = make_batch(cfg, s0, n)
out = pan.forward_batch_trace(b["nom_s"], b["nom_u"], b["ref_s"], b["ref_us"], b["points"], b["velocities"])
qi = out["trace_qp_info"].cpu().numpy()
bad = np.argwhere((qi[:, :, 3] != 0) | (qi[:, :, 1] > 1e-9))
print(wl, robot, "T", cfg.T, n, "scenes from", s0, "; QPs:", qi.shape[0] * qi.shape[1], "bad:", len(bad), "worst merit", qi[:, :, 1].max(),
      "share above 1e-13:", float((qi[:, :, 1] > 1e-13).mean()))
for s, k in bad[:12]:
    print("   scene", s0 + s, "iteration", k, "status", qi[s, k, 3], "merit", qi[s, k, 1], "iterations", qi[s, k, 14], "warm code", qi[s, k, 15], "best it", qi[s, k, 0])
