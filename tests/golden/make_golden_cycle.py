"""Generate tests/golden/cycle_*.npz: whole control cycles of the reference's `neupan` class (neupan/neupan.py:104-166:
check_arrive -> generate_nom_ref_state -> PAN.forward -> warm start -> stop test -> action), executed unmodified under
the stubs of ref_stub_loader.py EXCEPT the CvxpyLayer call, which is the oracle QP ("reference code with substituted
solver", as in make_golden.py).  The path is installed with the reference's own set_initial_path (gctl not needed).

    python tests/golden/make_golden_cycle.py          (build container only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import CKPT, OracleLayer, REAL  # noqa: E402  (imports the reference under stubs)
from neupan.neupan import neupan as RefNeupan  # noqa: E402

ROBOT = dict(kinematics="diff", max_speed=[8, 1], max_acce=[8, 3], length=1.6, width=2.0)
ADJUST = dict(q_s=1.0, p_u=1.0, eta=15.0, d_max=1.0, d_min=0.1)


def line(n, step, y=0.0, gear=1.0, x0=0.0, sgn=1.0):
    return [np.array([[x0 + sgn * i * step], [y], [0.0], [gear]]) for i in range(n)]


ACKER = dict(kinematics="acker", max_speed=[8, 1], max_acce=[8, 0.5], length=4.6, width=1.6, wheelbase=3.0)
OMNI = dict(kinematics="omni", max_speed=[8, 3.2], max_acce=[8, 8], length=1.6, width=2.0)


def build(iter_num, dune_max_num, iter_threshold, collision_threshold=0.1, robot=None, ckpt="diff_robot_default"):
    p = RefNeupan(receding=10, step_time=0.1, ref_speed=4.0, device="cpu", robot_kwargs=dict(robot or ROBOT),
                  ipath_kwargs=dict(waypoints=None, curve_style="line", loop=False),
                  pan_kwargs=dict(iter_num=iter_num, dune_max_num=dune_max_num, nrmp_max_num=10,
                                  dune_checkpoint=CKPT[ckpt], iter_threshold=iter_threshold),
                  adjust_kwargs=dict(ADJUST), train_kwargs=dict(), collision_threshold=collision_threshold, time_print=False)
    if not REAL:
        p.pan.nrmp_layer.nrmp_layer = OracleLayer(p.pan.nrmp_layer)
    return p


def step(kin, L, state, action):
    """the simulated robot: one Euler step of the reference's own motion models (initial_path.py:388-444); for omni the
    action is (vx, vy) (neupan.py:158-164)"""
    x, y, th = state[:, 0]
    a, b = float(action[0, 0]), float(action[1, 0])
    if kin == "diff":
        d = [a * np.cos(th), a * np.sin(th), b]
    elif kin == "acker":
        d = [a * np.cos(th), a * np.sin(th), a * np.tan(b) / L]
    else:
        d = [a, b, 0.0]
    return state + 0.1 * np.array(d).reshape(3, 1)


def run(name, path, state0, points, cycles, iter_num=2, dune_max_num=100, iter_threshold=0.1, velocities=None, robot=None,
        ckpt="diff_robot_default", points_until=None):
    """points_until: the cloud is delivered only in the first `points_until` cycles, None afterwards."""
    robot = dict(robot or ROBOT)
    p = build(iter_num, dune_max_num, iter_threshold, robot=robot, ckpt=ckpt)
    p.set_initial_path([q.copy() for q in path])
    state = np.asarray(state0, dtype=np.float64).reshape(3, 1)
    rec = dict(states=[], actions=[], arrive=[], stop=[], min_distance=[], opt_u=[], ref_s=[], point_index=[], curve_index=[])
    for c in range(cycles):
        rec["states"].append(state[:, 0].copy())
        pts_c = None if points is None else (points if velocities is None else points + c * 0.1 * velocities)
        if points_until is not None and c >= points_until:
            pts_c = None
        action, info = p(state.copy(), None if pts_c is None else pts_c.copy(), velocities)
        rec["actions"].append(np.asarray(action, dtype=np.float64).reshape(2))
        rec["arrive"].append(bool(info["arrive"])); rec["stop"].append(bool(info["stop"]))
        md = p.min_distance
        rec["min_distance"].append(float(md) if not isinstance(md, float) or np.isfinite(md) else np.inf)
        rec["opt_u"].append(np.asarray(p.cur_vel_array, dtype=np.float64).copy())
        rec["ref_s"].append(info["ref_state_tensor"].numpy().astype(np.float64) if "ref_state_tensor" in info else np.zeros((3, 11)))
        rec["point_index"].append(int(p.ipath.point_index)); rec["curve_index"].append(int(p.ipath.curve_index))
        state = step(robot["kinematics"], robot.get("wheelbase", 0.0), state, np.asarray(action, dtype=np.float64))
    out = {k: np.array(v) for k, v in rec.items()}
    out["path"] = np.hstack(path).T
    out["points"] = np.zeros((2, 0)) if points is None else points
    out["meta"] = np.array([iter_num, dune_max_num, iter_threshold, cycles if points_until is None else points_until], dtype=np.float64)
    out["velocities"] = np.zeros((2, 0)) if velocities is None else velocities
    import json
    out["robot"] = np.array(json.dumps(robot)); out["ckpt"] = np.array(ckpt)
    out["solver"] = np.array("reference" if REAL else "reference code with substituted solver")
    np.savez_compressed(os.path.join(HERE, f"cycle_{name}.npz"), **out)
    print(f"cycle_{name}.npz: {cycles} cycles, arrive {out['arrive'].astype(int).tolist()}, stop {out['stop'].astype(int).tolist()}, "
          f"curve {out['curve_index'].tolist()}")


if __name__ == "__main__":
    rng = np.random.default_rng(11)
    walls = np.stack([rng.uniform(2, 28, 300), rng.choice([-1, 1], 300) * rng.uniform(2.4, 4.0, 300)])
    walls[:, :3] = [[8.0, 13.0, 18.0], [0.35, -0.4, 0.25]]
    run("corridor", line(76, 0.4), [0.0, 0.05, 0.02], walls, 8)
    # a short forward curve followed by a reverse-gear curve: arrival of the first curve switches to the second
    two = line(5, 0.4, y=0.5) + line(20, 0.4, y=0.5, gear=-1.0, x0=1.6, sgn=-1.0)
    run("gear_switch", two, [1.56, 0.5, 0.0], walls[:, 3:153], 5)
    # the path ends right ahead: arrive flag, zero action from then on
    run("arrive", line(4, 0.4, y=-0.2), [1.0, -0.2, 0.0], walls[:, 3:103], 4)
    # an obstacle inside the collision threshold: stop flag
    close = np.concatenate([np.array([[0.83], [0.0]]), walls[:, 3:60]], axis=1)
    run("stop", line(40, 0.4), [0.0, 0.0, 0.0], close, 3)
    # ... and then the cloud disappears: DUNE.min_distance keeps its last value (dune.py:97-98), the robot stays stopped
    run("stop_then_empty", line(40, 0.4), [0.0, 0.0, 0.0], close, 4, points_until=2)
    # no obstacle points at all
    run("no_points", line(40, 0.4), [0.0, 0.3, 0.1], None, 3)
    # car-like robot backing up along a reverse-gear path
    run("acker_reverse", line(40, 0.4, gear=-1.0, sgn=-1.0), [0.0, 0.1, 0.03], walls[:, 3:203] - np.array([[30.0], [0.0]]), 5,
        robot=ACKER, ckpt="acker_robot_default")
    # omnidirectional robot: the action is (vx, vy)
    run("omni", line(60, 0.4), [0.0, -0.1, 0.0], walls, 5, robot=OMNI)
    # moving obstacle points (scan_to_point_velocity's output): iter_num 3, no early exit
    vel = rng.uniform(-1, 1, (2, 200))
    run("dyna", line(76, 0.4), [0.0, 0.0, 0.0], walls[:, :200], 5, iter_num=3, iter_threshold=0.0, velocities=vel)
