"""-m gpu: a control cycle entirely on the device -- lidar scan -> points (npa_scan_to_points), previous
controls -> nominal/reference states (npa_nominal_ref_states), PAN loop (npa_forward_batch) -- for a
small fleet over several cycles, against the same chain on the oracle (the reference's
neupan.forward order: neupan.py:104-137).  Tolerance: control L2 <= 1e-4 per robot and cycle."""
import numpy as np
import pytest

from helpers import CONFIGS, make_oracle
from oracle import frontend_oracle as fo

pytestmark = pytest.mark.gpu


def _scan(rng, pose, beams=360):
    """synthetic scan: corridor walls at y = +-3.5 in the world and a few discs, ray-cast coarsely"""
    ang = np.linspace(-np.pi, np.pi, beams)
    r = np.full(beams, 10.0)
    for i, a in enumerate(ang):
        d = np.array([np.cos(a + pose[2]), np.sin(a + pose[2])])
        for wall in (3.5, -3.5):
            if abs(d[1]) > 1e-6:
                t = (wall - pose[1]) / d[1]
                if 0.2 < t < r[i]:
                    r[i] = t
    r += rng.normal(0, 0.01, beams)
    return ang, np.clip(r, 0.05, 10.0)


def test_fleet_cycles_scan_rollout_pan_vs_oracle():
    import torch
    from gpu_helpers import make_gpu_pan
    from neupan_amd.frontend import NominalBatch, scan_to_point_batch
    cfg = CONFIGS["corridor_diff_small"]
    B, T, dt, cycles = 6, cfg.T, cfg.dt, 3
    rng = np.random.default_rng(42)
    # one straight path per robot, point spacing = ref_speed * dt (index stepping) for even robots and
    # a coarser one (circle-segment stepping) for odd robots
    curves, intervals = [], []
    for b in range(B):
        step = 0.4 if b % 2 == 0 else 0.9
        n = 80
        xs = np.arange(n) * step
        curves.append(np.column_stack([xs, np.full(n, 0.2 * b - 0.5), np.zeros(n), np.ones(n)]))
        intervals.append(step)
    poses = np.column_stack([rng.uniform(0, 1, B), rng.uniform(-0.6, 0.6, B), rng.uniform(-0.2, 0.2, B)])
    pidx = np.zeros(B, dtype=int)
    pan = make_gpu_pan(cfg, iter_num=3, dune_max_num=400)
    orcs = [make_oracle(cfg, iter_num=3, dune_max_num=400) for _ in range(B)]
    nb = NominalBatch(T, dt, "diff")
    nb.set_curves(curves, intervals, pidx)
    prev_u_gpu = None
    prev_u_orc = [np.zeros((2, T)) for _ in range(B)]
    for cyc in range(cycles):
        scans = [_scan(rng, poses[b]) for b in range(B)]
        ranges = np.stack([s[1] for s in scans])
        # ---- device chain
        pts, npts = scan_to_point_batch(poses, ranges, -np.pi, np.pi, 0.1, 10.0, max_points=400)
        nom = nb.generate_nom_ref_state(poses, prev_u_gpu, 4.0)
        out = pan.forward_batch(*nom, pts, None, npts)
        u_gpu = out["opt_u"].cpu().numpy()
        # ---- oracle chain (float64 front end, float32 at the PAN boundary as in neupan.py:121)
        for b in range(B):
            p = fo.scan_to_point(poses[b], ranges[b], -np.pi, np.pi, 0.1, 10.0)
            n_s, n_u, r_s, r_us = fo.generate_nom_ref_state(curves[b], int(pidx[b]), intervals[b], poses[b], prev_u_orc[b],
                                                            4.0, T, dt, "diff", 0.0)
            f32 = lambda a: np.asarray(a, dtype=np.float32)
            so, uo, do = orcs[b].forward(f32(n_s), f32(n_u), f32(r_s), f32(r_us), f32(p))
            assert int(npts[b]) == p.shape[1]
            assert np.linalg.norm(u_gpu[b].astype(np.float64) - uo) <= 1e-4, (cyc, b)
            prev_u_orc[b] = f32(uo)
        prev_u_gpu = out["opt_u"]
        # ---- advance the fleet with the first control of the plan (diff drive), closest path point on the host
        for b in range(B):
            v, w = float(u_gpu[b, 0, 0]), float(u_gpu[b, 1, 0])
            poses[b] += dt * np.array([v * np.cos(poses[b, 2]), v * np.sin(poses[b, 2]), w])
            pidx[b] = int(np.argmin(np.hypot(curves[b][:, 0] - poses[b, 0], curves[b][:, 1] - poses[b, 1])))
        nb.set_point_index(pidx)
        for o in orcs:
            pass          # the oracle keeps its stop-criterion memory across cycles like the GPU planner
