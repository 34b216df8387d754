"""GPU stress of the whole forward call on the round-6 QP kernel: many scenes of every T = 10 workload (seeds the test suite does not use)
and of the car, the call repeated: outputs bitwise identical from run to run, every last solve converged (status 0, merit <= 1e-9), every
output finite and inside its bounds.

    python tests/tools/qp_stress.py [reps] [scenes]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from helpers import CONFIGS
from gpu_helpers import make_gpu_pan
from neupan_amd.scenes import make_batch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
bad_total = 0
for name, seed in (("diff_1k_T10_K10", 7000), ("dyna_4k_T10_K10", 8000), ("poly8_5k_T10_K10", 9000), ("acker_2k_T20_K15", 10000)):
    cfg = CONFIGS[name]
    pan = make_gpu_pan(cfg)
    nb = B if not name.startswith(("poly8", "acker")) else B // 4
    batch = make_batch(cfg, seed, nb)
    args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
    kw = {"velocities": batch["velocities"]} if batch.get("velocities") is not None else {}
    ref, bad, worst, nstat = None, 0, 0.0, 0
    for rep in range(reps):
        out = pan.forward_batch(*args, reset_state=True, **kw)
        cur = {k: out[k].cpu().numpy() for k in ("opt_s", "opt_u", "opt_d")}
        info = pan.last_qp_info()
        nstat += int((info[:, 3] != 0).sum()); worst = max(worst, float(info[:, 1].max()))
        assert all(np.isfinite(v).all() for v in cur.values()), name
        if ref is None:
            ref = cur
        else:
            bad += int(sum((cur[k] != ref[k]).sum() for k in cur))
    print(f"{name:22s} scenes={nb:5d} reps={reps}: differing values {bad}, solves with status != 0: {nstat}, worst final merit {worst:.2e}")
    bad_total += bad + nstat + (worst > 1e-9)
print("TOTAL", bad_total)
sys.exit(1 if bad_total else 0)
