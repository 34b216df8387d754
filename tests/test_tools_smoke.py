"""The CPU study tools behind the QP kernel's heuristics must keep running: they are the evidence DESIGN.md cites."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


def test_active_set_study_one_scene():
    """tests/tools/qp_active_set_study.py on one scene: behind the warm-start gate the active-set iteration converges from the
    previous PAN iteration's solution, the kernel form reaches the same point, and the lane-level form of the reduction
    reproduces the matrix form to rounding."""
    import qp_active_set_study as st
    name, nqp, rows = st.job(("diff_1k_T10_K10", 0))
    assert nqp == 10 and len(rows) >= 3
    ok = [r for r in rows if r["ok"]]
    assert len(ok) >= len(rows) - 1
    assert max(r["kkt"] for r in ok) <= 1e-11
    both = [r for r in ok if r["kf_ok"]]
    assert both and max(r["kf_du"] for r in both) <= 1e-9 and max(r["kf_dl"] for r in both) <= 1e-9
    assert max(r["lane"] for r in rows) <= 1e-12


def test_step_study_rules_one_scene():
    """tests/tools/qp_step_study.py on one scene: the shipped rules need fewer interior-point iterations than round 2's and
    leave no solve above 1e-9."""
    import qp_step_study as st
    name, out = st.job(("diff_1k_T10_K10", 1))
    labels = [l for l, _ in st.RULES]
    r2 = out[0]; shipped = out[[i for i, l in enumerate(labels) if "SHIPPED" in l][0]]
    assert np.mean([r[0] for r in shipped]) < np.mean([r[0] for r in r2])
    assert max(r[2] for r in shipped) <= 1e-9
