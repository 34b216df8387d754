"""Construction of the HIP-backed PAN for the -m gpu tests."""
import numpy as np

from helpers import CONFIGS, ckpt_path
from neupan_amd.robot import Robot


def make_gpu_pan(cfg, robot_kw=None, checkpoint=None, **over):
    from neupan_amd.pan import PAN
    robot_kw = dict(cfg.robot if robot_kw is None else robot_kw)
    T = over.pop("receding", cfg.T)
    rb = Robot(T, cfg.dt, **robot_kw)
    adjust = dict(cfg.adjust)
    adjust.update(over.pop("adjust", {}))
    kw = dict(iter_num=cfg.iter_num, dune_max_num=cfg.n_points, nrmp_max_num=cfg.nrmp_max_num, iter_threshold=0.0,
              dune_checkpoint=checkpoint or ckpt_path(cfg.checkpoint), adjust_kwargs=adjust)
    kw.update(over)
    return PAN(T, cfg.dt, rb, **kw)


def l2(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)))


def wall_batch(cfg, B, seed=7):
    """Scenes that stress the selection: the make_batch trajectories with the cloud replaced by two walls parallel
    to the path sampled every 2 cm (dozens of points at practically the same distance from the robot's side: many
    more candidates than one tile inside the key margin), and for every fourth scene a tight blob of 48 points
    with n_points = 48 (more than half of the slice inside the margin: the whole-slice path)."""
    from neupan_amd.scenes import make_batch
    batch = make_batch(cfg, 1000, B)
    rng = np.random.default_rng(seed)
    N = batch["points"].shape[2]
    pts = np.zeros((B, 2, N), dtype=np.float32)
    n_points = np.full(B, N, dtype=np.int32)
    for b in range(B):
        if b % 4 == 3:
            pts[b, :, :48] = (np.array([[3.0], [2.2]]) + rng.normal(0, 2e-3, (2, 48))).astype(np.float32)
            n_points[b] = 48
            continue
        half = N // 2
        xs = -2.0 + 0.02 * np.arange(half)
        off = 1.8 + 0.4 * rng.random()
        pts[b, 0, :half], pts[b, 1, :half] = xs, off + rng.normal(0, 1e-3, half)
        pts[b, 0, half:], pts[b, 1, half:] = -2.0 + 0.02 * np.arange(N - half), -off + rng.normal(0, 1e-3, N - half)
    batch["points"], batch["n_points"], batch["velocities"] = pts, n_points, None
    return batch
