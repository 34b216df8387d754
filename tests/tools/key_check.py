"""GPU diagnostic: determinism of the DUNE stage and agreement of the reduced-precision-key selection with the
exact-fp32-key selection.   python tests/tools/key_check.py out.npz [B] [other.npz] [config]   (run once per key mode)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from helpers import CONFIGS
from gpu_helpers import make_gpu_pan, wall_batch
from neupan_amd.scenes import make_batch

out = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = CONFIGS[sys.argv[4] if len(sys.argv) > 4 else "diff_1k_T10_K10"]
pan = make_gpu_pan(cfg)
print("key mode", pan.key_mode())
batch = wall_batch(cfg, B) if os.environ.get("KEY_CHECK_WALLS") else make_batch(cfg, 1000, B)
ref = None
nd = 0
for rep in range(8):
    r = pan.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"), batch.get("n_points"))
    cur = {k: v.cpu().numpy() for k, v in r.items()}
    if ref is None:
        ref = cur
    else:
        bad = int((cur["pts"] != ref["pts"]).any(axis=(2, 3)).sum())
        nd += bad
        print("rep", rep, "slices differing from rep 0:", bad)
print("nondeterministic slices total", nd)
np.savez(out, **ref)
if len(sys.argv) > 3 and sys.argv[3] not in ("", "-"):
    other = np.load(sys.argv[3])
    diff = (other["pts"] != ref["pts"]).any(axis=(2, 3))
    print("slices whose selection differs from", sys.argv[3], ":", int(diff.sum()), "of", diff.size)
    dd = np.abs(other["dist"] - ref["dist"])
    print("max |dist difference|", float(dd.max()))
