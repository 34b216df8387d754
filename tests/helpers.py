"""Shared test plumbing: fixture paths, oracle construction from a SceneConfig."""
import os

import numpy as np

from neupan_amd.scenes import CONFIGS, SceneConfig
from oracle.pan_oracle import (ObsPointNetWeights, PanOracle, cal_vertices,
                               gen_inequal_from_vertex)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ckpt_path(name):
    """reference checkpoints (example/model/<name>/model_5000.pth, copied as data fixtures) or the quick fit
    made by tests/golden/make_poly8_checkpoint.py"""
    p = os.path.join(GOLDEN, "checkpoints", f"{name}_model_5000.pth")
    return p if os.path.exists(p) else os.path.join(GOLDEN, "checkpoints", f"{name}_model_quick.pth")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def robot_numbers(robot_kw, dt):
    """(G, h, speed_bound, acce_bound, L) the way the reference's robot class derives them
    (robot.py:53-69): acker steering capped at 1.57, acce_bound = max_acce*dt."""
    v = cal_vertices(robot_kw.get("vertices"), robot_kw.get("length"), robot_kw.get("width"),
                     robot_kw.get("wheelbase"))
    G, h = gen_inequal_from_vertex(v)
    sp = np.array(robot_kw.get("max_speed", [np.inf, np.inf]), dtype=float)
    if robot_kw["kinematics"] == "acker" and sp[1] >= 1.57:
        sp[1] = 1.57
    ac = np.array(robot_kw.get("max_acce", [np.inf, np.inf]), dtype=float) * dt
    return G, h, sp, ac, robot_kw.get("wheelbase")


def make_oracle(cfg: SceneConfig, robot_kw=None, checkpoint=None, **over):
    robot_kw = dict(cfg.robot if robot_kw is None else robot_kw)
    T = over.pop("receding", cfg.T)
    G, h, sp, ac, L = robot_numbers(robot_kw, cfg.dt)
    w = ObsPointNetWeights.from_checkpoint(checkpoint or ckpt_path(cfg.checkpoint))
    kw = dict(iter_num=cfg.iter_num, dune_max_num=cfg.n_points, nrmp_max_num=cfg.nrmp_max_num,
              iter_threshold=0.0)
    adjust = dict(cfg.adjust)
    adjust.update(over.pop("adjust", {}))
    kw.update(over)
    return PanOracle(T, cfg.dt, G, h, w, robot_kw["kinematics"], L, speed_bound=sp, acce_bound=ac,
                     **kw, **adjust)
