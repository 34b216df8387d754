"""GPU stress: the DUNE stage repeated many times on every bench workload must give bitwise identical rows
(the encode kernel hands tiles to waves dynamically; see the hazard note in DESIGN.md section 3.1)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from helpers import CONFIGS
from gpu_helpers import make_gpu_pan
from neupan_amd.scenes import make_batch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad_total = 0
for name, B in (("diff_1k_T10_K10", 256), ("dyna_4k_T10_K10", 96), ("poly8_5k_T10_K10", 64), ("acker_2k_T20_K15", 96),
                ("corridor_diff_small", 7)):
    cfg = CONFIGS[name]
    pan = make_gpu_pan(cfg)
    batch = make_batch(cfg, 123, B)
    ref = None
    bad = 0
    for rep in range(reps):
        r = pan.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"))
        cur = {k: v.cpu().numpy() for k, v in r.items()}
        if ref is None:
            ref = cur
        else:
            bad += int(sum((cur[k] != ref[k]).sum() for k in ("mu", "lam", "pts", "dist")))
    print(f"{name:22s} B={B:4d} reps={reps}: differing values {bad}")
    bad_total += bad
print("TOTAL", bad_total)
sys.exit(1 if bad_total else 0)
