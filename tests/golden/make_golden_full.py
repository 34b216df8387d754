"""Full-size pan_<case>.npz vectors: BASELINE.json's configurations at their own sizes, recorded from the reference's PAN.forward
(pan.py:109-147, unmodified, CvxpyLayer call substituted by the oracle QP -- the recipe of make_golden.py).

    python tests/golden/make_golden_full.py          (build container only; ~10 minutes of CPU)

The PAN loop does not contract on every scene (tests/parity_tools.py): a golden vector is only a fair target where the reference
algorithm's answer is stable to rounding.  So the scenes are CHOSEN: for each workload the oracle ensemble (8 runs on inputs moved
by +-1 float32 ulp, 4 with the hidden units permuted) is run on a block of candidate scenes and the first two whose members agree
to 1e-5 at every iteration are recorded.  Workloads:

    diff_n1000_k10       configs[1]: diff robot, 1000 points, T = 10, K = 10
    acker_n2000_T20_k15  configs[2]: car, 2000 points, T = 20, K = 15 (one forward-gear and one reverse-gear scene)
    dyna_n4000_k10       configs[3]: 4000 moving points, T = 10, K = 10
    polygon_n5000_k10    configs[4]'s size on the polygon the reference SHIPS a checkpoint for (the 4-edge trapezoid,
                         example/polygon_robot): 5000 points, T = 10, K = 10  (the 8-edge hull's checkpoint is ours, not the
                         reference's, so no reference-recorded vector can exist for it)
"""
import json
import os
import sys

import numpy as np

from make_golden import CKPT, CONFIGS, HERE, ROOT, make_scene, pan_case  # noqa: F401  (imports the reference under stubs)

sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_tools as pt  # noqa: E402

WORK = [("diff_n1000_k10", "diff_1k_T10_K10", dict(iter_num=10)),
        ("acker_n2000_T20_k15", "acker_2k_T20_K15", dict(iter_num=15, receding=20)),
        ("dyna_n4000_k10", "dyna_4k_T10_K10", dict(iter_num=10)),
        ("polygon_n5000_k10", "polygon_5k_T10_K10", dict(iter_num=10, dune_checkpoint=CKPT["polygon_robot"]))]
FIRST, BLOCK, TOL = 400, 12, 1e-5


def pick(workload, want_gears=None):
    scenes = list(range(FIRST, FIRST + BLOCK))
    base, members, _, _ = pt.run_ensemble(workload, scenes, cores=min(8, os.cpu_count() or 1), sweep=False)
    sp = pt.spreads(base, members)                                   # [S, K] max pairwise control L2
    ok = [(b, float(sp[i].max())) for i, b in enumerate(scenes) if sp[i].max() <= TOL]
    print(workload, "well-posed (spread <= %g at every iteration):" % TOL, ok)
    if want_gears:                                                   # the car: one scene per gear (odd scene numbers reverse)
        out = []
        for par in (0, 1):
            out += [b for b, _ in ok if b % 2 == par][:1]
        return out, base, scenes
    return [b for b, _ in ok][:2], base, scenes


if __name__ == "__main__":
    report = {}
    for name, workload, over in WORK:
        cfg = CONFIGS[workload]
        chosen, base, scenes = pick(workload, want_gears=workload.startswith("acker"))
        assert len(chosen) == 2, (name, chosen)
        for i, b in enumerate(chosen):
            case = f"{name}_s{i}"
            pan_case(case, cfg, b, cfg.n_points, **over)
            g = np.load(os.path.join(HERE, f"pan_{case}.npz"))
            u_or = base[scenes.index(b), -1]                          # the oracle's own controls of that scene
            d = float(np.linalg.norm(g["c0_opt_u"].astype(np.float64) - u_or))
            print(f"  {case}: scene {b}, reference-recorded vs oracle control L2 = {d:.2e}")
            report[case] = {"scene": b, "ref_vs_oracle_ctrl_l2": d}
    print(json.dumps(report))
