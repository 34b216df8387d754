"""The single-robot `neupan` class (neupan_amd/planner.py, mirror of neupan/neupan.py:30-420): a planner.yaml in
the reference's own format is loaded unchanged and driven for several control cycles; every cycle is compared
with the same cycle assembled from the oracle pieces (path progress, rollout, PAN).  Tolerance: action <= 1e-4."""
import os

import numpy as np
import pytest

from helpers import CONFIGS, GOLDEN, make_oracle
from oracle import frontend_oracle as fo

YAML = """
# mpc
receding: 10
step_time: 0.1
ref_speed: 4
device: 'cpu'
time_print: False
collision_threshold: 0.1
robot:
  kinematics: 'diff'
  max_speed: [8, 1]
  max_acce: [8, 3]
  length: 1.6
  width: 2.0
ipath:
  waypoints: [[0, 20, 0], [30, 20, 0]]
  curve_style: 'line'
  min_radius: 4.0
  loop: False
  arrive_threshold: 0.1
  close_threshold: 0.1
  ind_range: 10
  arrive_index_threshold: 1
pan:
  iter_num: 2
  dune_max_num: 100
  nrmp_max_num: 10
  iter_threshold: 0.1
  dune_checkpoint: 'checkpoints/diff_robot_default_model_5000.pth'
adjust:
  q_s: 1.0
  p_u: 1.0
  eta: 15.0
  d_max: 1.0
  d_min: 0.1
"""


def _write_yaml(tmp_path):
    d = os.path.join(GOLDEN, "_tmp_yaml")          # below tests/golden so the relative checkpoint name resolves
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, f"planner_{os.getpid()}.yaml")
    with open(p, "w") as f:
        f.write(YAML)
    return p


def test_line_curve_and_yaml_keys():
    """host-only: the way-point path has the configured spacing, consistent headings and gear +1"""
    from neupan_amd.planner import _consistent_angles, line_curve
    path = line_curve([[0, 0, 0], [2.0, 0, 0], [2.0, 1.0, 0]], 0.4)
    _consistent_angles(path)
    xy = np.hstack(path)[:2].T
    seg = np.hypot(*np.diff(xy, axis=0).T)
    assert np.all(seg <= 0.4 + 1e-12) and np.all(seg > 0.3)
    assert np.allclose(xy[0], [0, 0]) and np.allclose(xy[-1], [2.0, 1.0])
    assert path[0][2, 0] == 0.0 and abs(path[-1][2, 0] - np.pi / 2) < 1e-12 and all(p[3, 0] == 1.0 for p in path)


@pytest.mark.gpu
def test_yaml_planner_closed_loop_vs_oracle(tmp_path):
    from neupan_amd.fleet import FleetPlanner
    from neupan_amd.planner import neupan
    try:
        planner = neupan.init_from_yaml(_write_yaml(tmp_path))
        cfg = CONFIGS["corridor_diff_small"]
        T, dt = 10, 0.1
        orc = make_oracle(cfg, iter_num=2, dune_max_num=100, iter_threshold=0.1)
        state = np.array([[0.0], [20.0], [0.0]])
        rng = np.random.default_rng(3)
        pts = np.stack([rng.uniform(3, 28, 300), 20 + rng.choice([-1, 1], 300) * rng.uniform(2.4, 4.0, 300)])
        pts[:, :3] = [[9.0, 14.0, 19.0], [20.3, 19.6, 20.2]]        # things to steer around
        prev_u = np.zeros((2, T))
        curve = pidx = None
        f32 = lambda a: np.asarray(a, dtype=np.float32)
        for cyc in range(8):
            action, info = planner.forward(state, pts)
            if curve is None:
                curve = FleetPlanner._split_by_gear(planner.initial_path)[0]
                pidx = 0
                assert abs(FleetPlanner._average_interval(planner.initial_path) - 0.4) < 1e-9 and len(curve) == 76
            pidx, _, arr = fo.path_progress(curve, pidx, state[:, 0])
            assert not arr and not info["arrive"]
            n_s, n_u, r_s, r_us = fo.generate_nom_ref_state(curve, pidx, 0.4, state[:, 0], prev_u, 4.0, T, dt, "diff", 0.0)
            so, uo, do = orc.forward(f32(n_s), f32(n_u), f32(r_s), f32(r_us), f32(pts))
            want = np.zeros((2, 1)) if orc.min_distance < 0.1 else uo[:, 0:1]
            assert np.abs(action - want).max() <= 1e-4, (cyc, action.ravel(), want.ravel())
            assert info["stop"] == bool(orc.min_distance < 0.1)
            assert abs(float(planner.min_distance) - float(orc.min_distance)) <= 1e-4
            assert len(planner.opt_trajectory) == T + 1 and planner.opt_trajectory[0].shape == (3, 1)
            assert planner.dune_points.shape[0] == 2 and planner.nrmp_points.shape == (2, 10)
            prev_u = f32(uo)
            v, w = float(action[0, 0]), float(action[1, 0])
            state = state + dt * np.array([[v * np.cos(state[2, 0])], [v * np.sin(state[2, 0])], [w]])
        assert state[0, 0] > 1.5                                   # it drove
        # adjust parameters and reset behave like the reference's
        planner.update_adjust_parameters(eta=12.0, d_max=0.8)
        assert abs(float(planner.adjust_parameters[2]) - 12.0) < 1e-6
        planner.reset()
        assert not planner.info["arrive"] and np.all(planner.cur_vel_array == 0)
        # scan -> points through the same object
        ang = np.linspace(-np.pi, np.pi, 90)
        scan = dict(ranges=np.full(90, 3.0), angle_min=-np.pi, angle_max=np.pi, range_min=0.1, range_max=10.0)
        p = planner.scan_to_point(state, scan)
        want = fo.scan_to_point(state[:, 0], scan["ranges"], -np.pi, np.pi, 0.1, 10.0)
        assert p.shape == want.shape and np.abs(p - want).max() < 1e-5
    finally:
        import shutil
        shutil.rmtree(os.path.join(GOLDEN, "_tmp_yaml"), ignore_errors=True)


@pytest.mark.gpu
def test_lon_style_step_through_the_facade(tmp_path):
    """example/LON/LON_corridor.py:94-127 as written: adjust parameters that require grad, a loss on
    info['distance_tensor'], backward, an optimiser step -- the plan itself is unchanged by tracking gradients"""
    import torch
    from neupan_amd.planner import neupan
    try:
        yaml_file = _write_yaml(tmp_path)
        planner, plain = neupan.init_from_yaml(yaml_file), neupan.init_from_yaml(yaml_file)
        q_s, p_u, eta, d_max, d_min = planner.adjust_parameters
        for p in (p_u, eta, d_max):
            p.requires_grad_(True)
        opt = torch.optim.Adam([p_u, eta, d_max], lr=5e-3)
        state = np.array([[0.0], [20.0], [0.0]])
        rng = np.random.default_rng(4)
        pts = np.stack([rng.uniform(3, 20, 120), 20 + rng.choice([-1, 1], 120) * rng.uniform(1.6, 3.0, 120)])
        opt.zero_grad()
        action, info = planner(state, pts)
        action0, _ = plain(state, pts)
        assert np.array_equal(action, action0)
        loss = 10 * (50 - torch.sum(info["distance_tensor"]))
        loss.backward()
        before = [p.item() for p in (p_u, eta, d_max)]
        assert all(p.grad is not None and np.isfinite(float(p.grad)) for p in (p_u, eta, d_max))
        assert eta.grad.item() < 0                         # a larger eta buys more clearance
        opt.step()
        assert eta.item() > before[1]
        action, info = planner(state, pts)                 # the updated values are what the next cycle solves with
        assert np.isfinite(action).all() and not np.array_equal(action, action0)
    finally:
        import shutil
        shutil.rmtree(os.path.join(GOLDEN, "_tmp_yaml"), ignore_errors=True)
