"""Timing of the front-end kernels beside the CPU restatement.

    python tests/tools/frontend_bench.py            # prints one JSON object

GPU: average of `reps` launches between two events on the launch stream, inputs resident.
CPU: oracle/frontend_oracle.py (the reference's algorithm, numpy/python, 1 thread) on a sample.
Both steps are latency-bound at these sizes; the algorithmic HBM bytes are reported to show it."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def _time_gpu(fn, reps=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us per call (includes the wrapper's small H2D copies)


def measure(B=256, T=10, beams=1080, cpu_sample=16):
    from neupan_amd.frontend import NominalBatch, scan_to_point_batch
    from neupan_amd import _lib
    from oracle import frontend_oracle as fo
    import ctypes as C
    rng = np.random.default_rng(3)
    # ---- nominal
    curves, itv, pidx, states = [], [], [], []
    for b in range(B):
        n = 80
        step = 0.4
        head = np.cumsum(rng.uniform(-0.05, 0.05, n))
        xy = np.cumsum(np.stack([step * np.cos(head), step * np.sin(head)], axis=1), axis=0)
        curves.append(np.column_stack([xy, head, np.ones(n)])); itv.append(step if b % 2 else 1.0); pidx.append(3)
        states.append([xy[3, 0], xy[3, 1] + 0.1, head[3]])
    states = np.asarray(states)
    vel = np.stack([rng.uniform(2, 5, (B, T)), rng.uniform(-0.3, 0.3, (B, T))], axis=1).astype(np.float32)
    nb = NominalBatch(T, 0.1, "diff")
    nb.set_curves(curves, itv, pidx)
    st_d = torch.from_numpy(states).cuda(); vel_d = torch.from_numpy(vel).cuda()
    nom_us = _time_gpu(lambda: nb.generate_nom_ref_state(st_d, vel_d, 4.0))
    t0 = time.perf_counter()
    for b in range(cpu_sample):
        fo.generate_nom_ref_state(curves[b], pidx[b], itv[b], states[b], vel[b], 4.0, T, 0.1, "diff", 0.0)
    nom_cpu_us = (time.perf_counter() - t0) / cpu_sample * 1e6
    nom_bytes = B * (24 + 8 * T + 8 + 16 + (T + 2) * 32 + 4 * (3 * (T + 1) * 2 + 2 * T + T))
    # ---- scan
    ranges = rng.uniform(0.2, 9.0, (B, beams)); ranges[rng.random((B, beams)) < 0.2] = 10.0
    poses = np.column_stack([rng.uniform(-5, 5, B), rng.uniform(-5, 5, B), rng.uniform(-3, 3, B)])
    r_d = torch.from_numpy(ranges).cuda()
    scan_us = _time_gpu(lambda: scan_to_point_batch(poses, r_d, -np.pi, np.pi, 0.1, 10.0))
    t0 = time.perf_counter()
    for b in range(cpu_sample):
        fo.scan_to_point(poses[b], ranges[b], -np.pi, np.pi, 0.1, 10.0)
    scan_cpu_us = (time.perf_counter() - t0) / cpu_sample * 1e6
    scan_bytes = B * (beams * 8 + int(0.8 * beams) * 8 + 104)
    # ---- a whole control cycle (neupan.forward's order) for B robots: reference defaults iter_num = 2,
    #      <= 200 points (planner.yaml of the corridor example), 1080-beam scans
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import CONFIGS, ckpt_path
    from neupan_amd.fleet import FleetPlanner
    from neupan_amd.robot import Robot
    cfg = CONFIGS["corridor_diff_small"]
    robot = Robot(cfg.T, cfg.dt, **cfg.robot)
    fleet = FleetPlanner(robot, cfg.T, cfg.dt, 4.0, dune_checkpoint=ckpt_path(cfg.checkpoint), iter_num=2, dune_max_num=200,
                         nrmp_max_num=cfg.nrmp_max_num, iter_threshold=0.1, adjust_kwargs=dict(cfg.adjust))
    fleet.set_paths([[np.array([[i * 0.4], [0.3 * (b % 5) - 0.6], [0.0], [1.0]]) for i in range(80)] for b in range(B)])
    fposes = np.column_stack([rng.uniform(0, 2, B), rng.uniform(-0.6, 0.6, B), rng.uniform(-0.1, 0.1, B)])

    def cycle():
        pts, npts = fleet.scan_to_point(fposes, r_d, -np.pi, np.pi, 0.1, 10.0, max_points=1080)
        fleet.forward(fposes, pts, None, npts)
    cyc_us = _time_gpu(cycle, reps=20)
    return {
        "fleet_cycle": {"robots": B, "us_per_cycle": round(cyc_us, 1), "robot_cycles_per_s": round(B / cyc_us * 1e6),
                        "what": "scan->points, path progress, nominal rollout, PAN (K=2, <=200 of 1080 scan points after "
                                "decimation, default iter_threshold), stop test; includes one host read of the arrival flags",
                        "reference": "README: ~15 Hz per robot on an i7 CPU"},
        "nominal_ref_states": {"scenes": B, "T": T, "us_per_call": round(nom_us, 1),
                               "scenes_per_s": round(B / nom_us * 1e6), "algorithmic_bytes": nom_bytes,
                               "hbm_GBps": round(nom_bytes / nom_us * 1e-3, 3),
                               "cpu_oracle_us_per_scene": round(nom_cpu_us, 1), "bound": "latency (serial T-step chain per scene)"},
        "scan_to_points": {"scans": B, "beams": beams, "us_per_call": round(scan_us, 1),
                           "scans_per_s": round(B / scan_us * 1e6), "algorithmic_bytes": scan_bytes,
                           "hbm_GBps": round(scan_bytes / scan_us * 1e-3, 3),
                           "cpu_oracle_us_per_scan": round(scan_cpu_us, 1), "bound": "latency (one small launch)"},
        "note": "us_per_call includes the Python wrapper (parameter upload, output allocation)",
    }


if __name__ == "__main__":
    print(json.dumps(measure()))
