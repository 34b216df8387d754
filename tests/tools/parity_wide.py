"""Wide parity legs for the configurations the bench line judges on 16 scenes x 6 members: 64 scenes x 12 ensemble members
(8 runs on inputs moved by +-1 ulp, 4 with permuted hidden units) for configs[2] (car), configs[3] (moving cloud) and
configs[4] (8-edge hull), HIP traces against the oracle, verdicts A / C and the one-step verdict D with every step above
tolerance explained (tests/test_gpu_parity._ensemble_verdict: the function the -m gpu suite runs on 24 / 24 / 16 scenes).

    python tests/tools/parity_wide.py [scenes] > gpurun_out/r06_parity_wide.json        (GPU box; ~10 minutes, mostly host CPU)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_parity as tg
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    out = {"scenes": scenes, "members": 12}
    for wl in ("acker_2k_T20_K15", "dyna_4k_T10_K10", "poly8_5k_T10_K10"):
        rep = tg._ensemble_verdict(wl, scenes, step_tol=1e-5 if wl.startswith("acker") else None)
        d = rep["one_step"]
        out[wl] = {
            "ctrl_l2_vs_oracle_median": rep["ctrl_l2_vs_oracle_median"], "max": rep["max"], "frac_le_1e-4": rep["frac_le_1e-4"],
            "scenes_well_posed": rep["scenes_well_posed"], "max_over_well_posed": rep["max_over_well_posed"],
            "A": rep["A_well_posed_all_le_tol"], "C": rep["C_le_1e-5_until_ensemble_diverges"],
            "max_hip_before_divergence": rep["max_hip_before_divergence"],
            "one_step": {k: d[k] for k in ("steps", "median", "max", "frac_le_tol", "unexplained", "stalled", "tol") if k in d},
            "one_step_classes": {w.get("explained"): sum(1 for v in d.get("above_tol", []) if v.get("explained") == w.get("explained"))
                                 for w in d.get("above_tol", [])},
        }
        print(wl, out[wl], file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
