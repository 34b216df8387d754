"""Whole control cycles against the reference's own `neupan.forward` (neupan/neupan.py:104-166), recorded by
tests/golden/make_golden_cycle.py from the unmodified reference code with the QP solver substituted (cycle_*.npz):
straight corridor, gear switch, arrival, collision stop, no obstacle points.

CPU: the oracle pieces chained in the reference's order must reproduce the recorded actions (<= 5e-6: a few float32 ulps of
the 4 m/s controls accumulate over the closed loop), flags and path indices.
GPU (-m gpu): `neupan_amd.neupan` (planner.py -> FleetPlanner -> HIP kernels) fed the recorded states must do the same;
tolerance 1e-4 on the action, exact on the flags."""
import os

import numpy as np
import pytest

from helpers import CONFIGS, GOLDEN, make_oracle
from oracle import frontend_oracle as fo

CASES = ["corridor", "gear_switch", "arrive", "stop", "stop_then_empty", "no_points", "acker_reverse", "omni", "dyna"]
ROBOT = dict(kinematics="diff", max_speed=[8, 1], max_acce=[8, 3], length=1.6, width=2.0)
ADJUST = dict(q_s=1.0, p_u=1.0, eta=15.0, d_max=1.0, d_min=0.1)


def _load(name):
    g = np.load(os.path.join(GOLDEN, f"cycle_{name}.npz"))
    path = [row.reshape(4, 1).copy() for row in g["path"]]
    pts = g["points"] if g["points"].shape[1] else None
    vel = g["velocities"] if g["velocities"].shape[1] else None
    import json
    return g, path, pts, vel, json.loads(str(g["robot"])), str(g["ckpt"])


def _split(path):
    curves, cur, gear = [], [], path[0][3, 0]
    for p in path:
        if p[3, 0] != gear:
            curves.append(np.hstack(cur).T); cur = []; gear = p[3, 0]
        cur.append(p)
    curves.append(np.hstack(cur).T)
    return curves


@pytest.mark.parametrize("name", CASES)
def test_oracle_chain_reproduces_reference_cycles(name):
    from helpers import ckpt_path
    g, path, pts, vel, robot, ck = _load(name)
    it, dmax, thr = int(g["meta"][0]), int(g["meta"][1]), float(g["meta"][2])
    kin, L = robot["kinematics"], robot.get("wheelbase", 0.0)
    orc = make_oracle(CONFIGS["corridor_diff_small"], robot_kw=robot, checkpoint=ckpt_path(ck), iter_num=it, dune_max_num=dmax,
                      iter_threshold=thr, adjust=dict(ADJUST))
    curves = _split(path)
    interval = sum(np.hypot(*(b[:2, 0] - a[:2, 0])) for a, b in zip(path, path[1:])) / (len(path) - 1)
    ci = pi_ = 0
    arrived = False
    prev_u = np.zeros((2, 10))
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    for c in range(len(g["states"])):
        st = g["states"][c]
        # check_arrive (initial_path.py:247-277)
        pi_, _, arr = fo.path_progress(curves[ci], pi_, st)
        ret_arrive = False
        if arr:
            if ci + 1 >= len(curves):
                arrived = ret_arrive = True
            else:
                ci += 1; pi_ = 0
        assert bool(g["arrive"][c]) == (arrived if ret_arrive or arrived else False), (name, c)
        assert int(g["curve_index"][c]) == ci and int(g["point_index"][c]) >= 0, (name, c)
        if ret_arrive or arrived:
            assert np.all(g["actions"][c] == 0)
            continue
        n_s, n_u, r_s, r_us = fo.generate_nom_ref_state(curves[ci], pi_, interval, st, prev_u, 4.0, 10, 0.1, kin, L)
        pts_c = None if pts is None else (pts if vel is None else pts + c * 0.1 * vel)
        if len(g["meta"]) > 3 and c >= int(g["meta"][3]):
            pts_c = None                                  # the cloud is gone: DUNE.min_distance keeps its last value
        so, uo, do = orc.forward(f32(n_s), f32(n_u), f32(r_s), f32(r_us), None if pts_c is None else f32(pts_c),
                                 None if vel is None else f32(vel))
        prev_u = f32(uo)
        stop = bool(orc.min_distance < 0.1)
        assert stop == bool(g["stop"][c]), (name, c)
        want = np.zeros(2) if stop else f32(uo)[:, 0].astype(np.float64)
        if kin == "omni" and not stop:                    # neupan.py:158-164
            want = np.array([want[0] * np.cos(want[1]), want[0] * np.sin(want[1])])
        assert np.abs(want - g["actions"][c]).max() <= 5e-6, (name, c, want, g["actions"][c])
        assert np.abs(f32(uo) - g["opt_u"][c]).max() <= 5e-6, (name, c)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_planner_reproduces_reference_cycles(name):
    from helpers import ckpt_path
    from neupan_amd.planner import neupan
    g, path, pts, vel, robot, ck = _load(name)
    it, dmax, thr = int(g["meta"][0]), int(g["meta"][1]), float(g["meta"][2])
    planner = neupan(receding=10, step_time=0.1, ref_speed=4.0, robot_kwargs=dict(robot), ipath_kwargs=dict(curve_style="line"),
                     pan_kwargs=dict(iter_num=it, dune_max_num=dmax, nrmp_max_num=10, iter_threshold=thr,
                                     dune_checkpoint=ckpt_path(ck)),
                     adjust_kwargs=dict(ADJUST), collision_threshold=0.1)
    planner.set_initial_path(path)
    for c in range(len(g["states"])):
        pts_c = None if pts is None else (pts if vel is None else pts + c * 0.1 * vel)
        if len(g["meta"]) > 3 and c >= int(g["meta"][3]):
            pts_c = None
        action, info = planner(g["states"][c].reshape(3, 1), pts_c, vel)
        assert bool(info["arrive"]) == bool(g["arrive"][c]) and bool(info["stop"]) == bool(g["stop"][c]), (name, c)
        assert np.abs(action.reshape(2) - g["actions"][c]).max() <= 1e-4, (name, c, action.ravel(), g["actions"][c])
        if not g["arrive"][c]:
            assert np.abs(planner.cur_vel_array - g["opt_u"][c]).max() <= 1e-4, (name, c)
            if np.isfinite(g["min_distance"][c]):
                assert abs(float(planner.min_distance) - g["min_distance"][c]) <= 1e-4, (name, c)
            else:
                assert not np.isfinite(float(planner.min_distance))
