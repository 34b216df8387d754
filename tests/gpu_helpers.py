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
