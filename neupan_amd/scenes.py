"""Deterministic synthetic planning scenes for the benchmark configurations of
BASELINE.json.  Pure numpy; shared by bench.py, the tests and the
golden-vector generator so that every leg sees bit-identical inputs.

A *scene* is one independent planning instance: the tensors `PAN.forward` consumes
(reference: neupan/blocks/pan.py:109-127) --
    nom_s (3,T+1)  nom_u (2,T)  ref_s (3,T+1)  ref_us (T,)  points (2,N)  velocities (2,N)|None
The nominal state is the kinematic rollout of the warm-start controls, as the reference's
initial-path block produces it (neupan/blocks/initial_path.py:87-113, 388-444).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from math import cos, sin, tan

import numpy as np

SEED0 = 20260923


@dataclass
class SceneConfig:
    name: str
    kinematics: str = "diff"
    T: int = 10
    dt: float = 0.1
    n_points: int = 1000
    iter_num: int = 10
    nrmp_max_num: int = 10
    moving: bool = False
    reverse_half: bool = False
    robot: dict = field(default_factory=dict)
    adjust: dict = field(default_factory=dict)
    checkpoint: str = "diff_robot_default"
    ref_speed: float = 4.0
    x_range: tuple = (-2.0, 14.0)    # extent of the corridor walls
    wall_half_width: tuple = (3.0, 4.5)
    obs_x: tuple = (2.5, 12.0)       # obstacle centres (mirrored for reverse gear)
    obs_gap: tuple = (1.25, 2.6)     # lateral distance of an obstacle's surface from the lane
    obs_radius: tuple = (0.4, 1.0)
    obs_count: tuple = (3, 7)
    cloud: str = "corridor"          # "corridor": walls + round obstacles (below); "uniform": SURVEY.md 8(d)'s uniform cloud


# robot + adjust values: reference example/corridor/diff/planner.yaml,
# example/reverse/acker/planner.yaml, example/dyna_non_obs/diff/planner.yaml
CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "diff_1k_T10_K10": SceneConfig(
        name="diff_1k_T10_K10",
        robot=dict(kinematics="diff", length=1.6, width=2.0, max_speed=[8, 1], max_acce=[8, 3]),
        adjust=dict(q_s=1.0, p_u=1.0, eta=15.0, d_max=1.0, d_min=0.1, bk=0.1, ro_obs=400),
    ),
    # the same configuration on the cloud SURVEY.md section 8(d) literally proposes for config 2: N points uniform over
    # x in [-2, 12], y in [-6, 6], rejecting |x| < 1.3 and |y| < 1.5 around the start pose.  6 points per square metre with
    # a 1.6 x 2.0 m robot driving through them: no collision-free plan exists on most scenes (d sits at d_min, in-collision
    # points tie at distance 0), the PAN iteration is ill posed on most of them -- reported next to the corridor workload as
    # the worst case of the selection (candidates per slice) and of the parity verdicts, not as the headline
    "uniform_1k_T10_K10": SceneConfig(
        name="uniform_1k_T10_K10", cloud="uniform",
        robot=dict(kinematics="diff", length=1.6, width=2.0, max_speed=[8, 1], max_acce=[8, 3]),
        adjust=dict(q_s=1.0, p_u=1.0, eta=15.0, d_max=1.0, d_min=0.1, bk=0.1, ro_obs=400),
    ),
    # configs[0]-like: reference-default sizes (<=200 pts after decimation)
    "corridor_diff_small": SceneConfig(
        name="corridor_diff_small", n_points=200,
        robot=dict(kinematics="diff", length=1.6, width=2.0, max_speed=[8, 1], max_acce=[8, 3]),
        adjust=dict(q_s=1.0, p_u=1.0, eta=15.0, d_max=1.0, d_min=0.1, bk=0.1, ro_obs=400),
    ),
    # configs[2]
    "acker_2k_T20_K15": SceneConfig(
        name="acker_2k_T20_K15", kinematics="acker", T=20, n_points=2000, iter_num=15,
        reverse_half=True, checkpoint="acker_robot_default",
        robot=dict(kinematics="acker", length=4.6, width=1.6, wheelbase=3, max_speed=[8, 1],
                   max_acce=[8, 0.5]),
        adjust=dict(q_s=1.0, p_u=1.0, eta=15.0, d_max=1.0, d_min=0.1, bk=0.1, ro_obs=400),
        # a 4.6 m car with a 3 m wheelbase needs a wider lane than the 1.6 m diff robot
        x_range=(-6.0, 24.0), wall_half_width=(4.5, 6.0), obs_x=(7.0, 20.0), obs_gap=(1.4, 2.8),
        obs_count=(2, 5),
    ),
    # configs[3]: moving points (dyna_non_obs), 4000 pts/scene
    "dyna_4k_T10_K10": SceneConfig(
        name="dyna_4k_T10_K10", n_points=4000, moving=True,
        robot=dict(kinematics="diff", length=1.6, width=2.0, max_speed=[8, 1], max_acce=[8, 3]),
        adjust=dict(q_s=0.5, p_u=1.0, eta=15.0, d_max=1.0, d_min=0.1, bk=1.0, ro_obs=400),
    ),
    # configs[4]: 8-vertex hull, 5000 pts.  The reference ships no 8-edge robot or E = 8 checkpoint: the hull is
    # ours and the weights are a quick fit to closed-form labels (tests/golden/make_poly8_checkpoint.py)
    "poly8_5k_T10_K10": SceneConfig(
        name="poly8_5k_T10_K10", n_points=5000, checkpoint="poly8",
        robot=dict(kinematics="diff", vertices=[[-0.6, -0.8], [0.6, -0.8], [1.0, -0.4], [1.0, 0.4], [0.6, 0.8],
                                                  [-0.6, 0.8], [-1.0, 0.4], [-1.0, -0.4]],
                   max_speed=[8, 1], max_acce=[8, 3]),
        adjust=dict(q_s=1.0, p_u=1.0, eta=15.0, d_max=1.0, d_min=0.1, bk=0.1, ro_obs=400),
    ),
    # configs[4]'s size on the polygon the reference SHIPS a checkpoint for (the 4-edge trapezoid of example/polygon_robot,
    # planner.yaml:16 / example/model/polygon_robot): what a reference-recorded full-size vector can be made of
    # (tests/golden/make_golden_full.py)
    "polygon_5k_T10_K10": SceneConfig(
        name="polygon_5k_T10_K10", n_points=5000, checkpoint="polygon_robot",
        robot=dict(kinematics="diff", vertices=[[-0.8, -1.0], [-1.8, 1.0], [1.8, 1.0], [0.8, -1.0]],
                   max_speed=[8, 3], max_acce=[8, 3]),
        adjust=dict(q_s=1.0, p_u=1.0, eta=15.0, d_max=1.0, d_min=0.1, bk=0.1, ro_obs=400),
    ),
}


def rollout(kinematics, state0, nom_u, dt, L=None):
    """Nominal-state rollout, fp64 (initial_path.py:388-444)."""
    T = nom_u.shape[1]
    s = np.zeros((3, T + 1))
    s[:, 0] = state0
    for t in range(T):
        x, y, th = s[:, t]
        v, w = nom_u[0, t], nom_u[1, t]
        if kinematics == "diff":
            ds = np.array([v * cos(th), v * sin(th), w])
        elif kinematics == "acker":
            ds = np.array([v * cos(th), v * sin(th), v * tan(w) / L])
        else:  # omni: u = (speed, heading)
            ds = np.array([v * cos(w), v * sin(w), 0.0])
        s[:, t + 1] = s[:, t] + ds * dt
    return s


def _obstacle_cloud(rng, cfg, N, gear):
    """Lidar-like cloud: two gently curved corridor walls plus a handful of round obstacles
    that intrude from either side of the lane y=0, sampled on their perimeters.  Surfaces
    stay >= obs_gap[0] from the lane so a collision-free plan exists, but within reach of
    d_max so that the hinge rows of the QP are active.  Moving configs give every obstacle
    its own velocity in [-1,1]^2 (IR-SIM's vxmax=vymax=1, example/dyna_non_obs/diff/env.yaml:42)."""
    n_obs = int(rng.integers(cfg.obs_count[0], cfg.obs_count[1] + 1))
    n_wall = int(0.4 * N)
    counts = np.full(n_obs, (N - n_wall) // n_obs)
    counts[: (N - n_wall) - counts.sum()] += 1
    hw = rng.uniform(*cfg.wall_half_width)
    ph = rng.uniform(0, 2 * np.pi, 2)
    xs = rng.uniform(*cfg.x_range, n_wall)
    side = np.where(np.arange(n_wall) % 2 == 0, 1.0, -1.0)
    ys = side * (hw + 0.3 * np.sin(0.5 * xs + np.where(side > 0, ph[0], ph[1])))
    px, py = [gear * xs], [ys]
    vx, vy = [np.zeros(n_wall)], [np.zeros(n_wall)]
    for k in range(n_obs):
        r = rng.uniform(*cfg.obs_radius)
        g = rng.uniform(*cfg.obs_gap)
        sd = 1.0 if rng.uniform() < 0.5 else -1.0
        cx, cy = gear * rng.uniform(*cfg.obs_x), sd * (g + r)
        ang = rng.uniform(0, 2 * np.pi, counts[k])
        px.append(cx + r * np.cos(ang)); py.append(cy + r * np.sin(ang))
        v = rng.uniform(-1.0, 1.0, 2)
        vx.append(np.full(counts[k], v[0])); vy.append(np.full(counts[k], v[1]))
    pts = np.vstack([np.concatenate(px), np.concatenate(py)])
    vel = np.vstack([np.concatenate(vx), np.concatenate(vy)]) if cfg.moving else None
    perm = rng.permutation(N)           # a lidar sweep does not deliver points grouped by obstacle
    return pts[:, perm], (None if vel is None else vel[:, perm])


def _uniform_cloud(rng, cfg, N):
    """x ~ U(-2, 12), y ~ U(-6, 6), rejecting |x| < 1.3 and |y| < 1.5 (a feasible start)."""
    xs, ys = np.zeros(0), np.zeros(0)
    while xs.size < N:
        x = rng.uniform(-2.0, 12.0, N); y = rng.uniform(-6.0, 6.0, N)
        keep = ~((np.abs(x) < 1.3) & (np.abs(y) < 1.5))
        xs, ys = np.concatenate([xs, x[keep]]), np.concatenate([ys, y[keep]])
    pts = np.vstack([xs[:N], ys[:N]])
    vel = rng.uniform(-1.0, 1.0, (2, N)) if cfg.moving else None
    return pts, vel


def make_scene(cfg: SceneConfig, b: int, n_points: int | None = None):
    """Scene number `b` of configuration `cfg`, fp32 arrays."""
    rng = np.random.default_rng(SEED0 + b)
    T, dt = cfg.T, cfg.dt
    N = cfg.n_points if n_points is None else n_points
    th0 = rng.uniform(-0.3, 0.3)
    v0 = rng.uniform(2.0, 4.0)
    w0 = rng.uniform(-0.3, 0.3)
    gear = 1.0
    if cfg.reverse_half and (b % 2 == 1):
        gear, v0 = -1.0, -v0
    nom_u = np.tile(np.array([[v0], [w0]]), (1, T))
    nom_s = rollout(cfg.kinematics, np.array([0.0, 0.0, th0]), nom_u, dt, cfg.robot.get("wheelbase"))
    step = gear * cfg.ref_speed * dt
    ref_s = np.zeros((3, T + 1))
    ref_s[0, :] = step * np.arange(T + 1)
    ref_us = np.full((T,), gear * cfg.ref_speed)

    pts, vel = _uniform_cloud(rng, cfg, N) if cfg.cloud == "uniform" else _obstacle_cloud(rng, cfg, N, gear)
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    return dict(nom_s=f(nom_s), nom_u=f(nom_u), ref_s=f(ref_s), ref_us=f(ref_us), points=f(pts),
                velocities=f(vel))


def make_batch(cfg: SceneConfig, first: int, count: int, n_points: int | None = None):
    """`count` consecutive scenes starting at global index `first`, stacked on a leading
    batch axis (the layout the C-ABI consumes).  velocities is None for static configs."""
    scenes = [make_scene(cfg, first + i, n_points) for i in range(count)]
    out = {k: np.stack([s[k] for s in scenes]) for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")}
    out["velocities"] = None if scenes[0]["velocities"] is None else np.stack([s["velocities"] for s in scenes])
    return out
