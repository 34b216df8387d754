"""Prints the explanation record of every single step above 1e-5 on the car workload (tests/parity_tools._explain_step)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if __name__ == "__main__":
    import test_gpu_parity as t
    rep = t._ensemble_verdict(sys.argv[1] if len(sys.argv) > 1 else "acker_2k_T20_K15", 24, step_tol=1e-5)
    for w in rep["one_step"]["above_tol"]:
        print(json.dumps(w))
