"""`FleetPlanner`: B robots through one control cycle per call, on the device -- the batched counterpart of
`neupan.forward` (neupan/neupan.py:104-166).  Per cycle, in the reference's order:

    check_arrive        closest path point, arrival test            initial_path.py:247-277 -> npa_path_progress
    nominal / reference rollout of the previous controls            initial_path.py:68-126  -> npa_nominal_ref_states
    PAN loop            DUNE + NRMP, K iterations                   blocks/pan.py:109-147   -> npa_forward_batch
    warm start          cur_vel_array <- opt_u (not shifted)        neupan.py:137
    stop test           min_distance < collision_threshold          neupan.py:150-154, :169
    action              opt_u[:, 0] (omni: v cos, v sin)            neupan.py:156-164

Path generation (gctl curves) stays with the caller: `set_paths` takes each robot's initial path as the
reference's `set_initial_path` does (list of (4,1) points x, y, theta, gear) and splits it by gear like
`split_path_with_gear` (initial_path.py:289-315); switching a robot to its next curve is host bookkeeping.
"""
from __future__ import annotations

from math import cos, hypot, sin

import numpy as np
import torch

from .frontend import NominalBatch, scan_to_point_batch
from .pan import PAN


class FleetPlanner:
    def __init__(self, robot, receding=10, step_time=0.1, ref_speed=4.0, collision_threshold=0.1, device="cuda",
                 close_threshold=0.1, ind_range=10, arrive_threshold=0.1, arrive_index_threshold=1, loop=False, **pan_kwargs):
        """`robot`: as for PAN (G, h, kinematics, L, speed_bound, acce_bound); `pan_kwargs`: PAN's keyword arguments
        (iter_num, dune_max_num, nrmp_max_num, dune_checkpoint, iter_threshold, adjust_kwargs)."""
        self.robot, self.T, self.dt = robot, int(receding), float(step_time)
        self.ref_speed, self.collision_threshold = float(ref_speed), float(collision_threshold)
        self.close_threshold, self.ind_range = float(close_threshold), int(ind_range)
        self.arrive_threshold, self.arrive_index_threshold = float(arrive_threshold), int(arrive_index_threshold)
        self.loop = bool(loop)
        self.device = torch.device(device)
        self.pan = PAN(receding, step_time, robot, device=device, **pan_kwargs)
        self.nb = NominalBatch(receding, step_time, robot.kinematics, getattr(robot, "L", 0.0) or 0.0, device=device)
        self.B = 0

    # ------------------------------------------------------------------ paths
    @staticmethod
    def _split_by_gear(path):
        """initial_path.py:289-315"""
        pts = [np.asarray(p, dtype=np.float64).reshape(4) for p in path]
        curves, cur, gear = [], [], pts[0][3]
        for p in pts:
            if p[3] != gear:
                curves.append(np.array(cur)); cur = []; gear = p[3]
            cur.append(p)
        if cur:
            curves.append(np.array(cur))
        return curves

    @staticmethod
    def _average_interval(path):
        """initial_path.py:144-158"""
        pts = [np.asarray(p, dtype=np.float64).reshape(4) for p in path]
        if len(pts) < 2:
            return 0.0
        return sum(hypot(b[0] - a[0], b[1] - a[1]) for a, b in zip(pts, pts[1:])) / (len(pts) - 1)

    def set_paths(self, paths):
        """paths: one initial path per robot (list of (4,1) / length-4 points).  Resets every robot to the start
        of its first curve and clears the warm start, like `set_initial_path` + `reset` (neupan.py:286-302)."""
        self.B = len(paths)
        self.curve_lists = [self._split_by_gear(p) for p in paths]
        self.intervals = [self._average_interval(p) for p in paths]
        self.curve_index = [0] * self.B
        self._upload()
        self.cur_vel = None                          # zeros on the first cycle (neupan.py:73)
        self.arrived = np.zeros(self.B, dtype=bool)

    def _upload(self, point_index=None):
        self.nb.set_curves([cl[i] for cl, i in zip(self.curve_lists, self.curve_index)], self.intervals, point_index)

    # ------------------------------------------------------------------ one control cycle
    def forward(self, states, points=None, velocities=None, n_points=None):
        """states [B,3]; points [B,2,N] float32 (global frame) with optional n_points [B] / velocities [B,2,N].
        Returns (action [B,2] float32 device tensor, info dict)."""
        B, dev = self.B, self.device
        st = np.asarray(states.cpu() if isinstance(states, torch.Tensor) else states, dtype=np.float64).reshape(B, -1)[:, :3]
        # 1. progress along the path, arrival (host bookkeeping only when a curve ends)
        pidx, _, arr = self.nb.progress(st, self.close_threshold, self.ind_range, self.arrive_threshold,
                                        self.arrive_index_threshold)
        arr_h = arr.cpu().numpy().astype(bool)
        if arr_h.any():
            pidx_h = pidx.cpu().numpy().copy()
            switched = False
            for b in np.nonzero(arr_h & ~self.arrived)[0]:
                if self.curve_index[b] + 1 >= len(self.curve_lists[b]):
                    if self.loop:
                        self.curve_index[b] = 0; pidx_h[b] = 0; switched = True
                    else:
                        self.arrived[b] = True
                else:
                    self.curve_index[b] += 1; pidx_h[b] = 0; switched = True
            if switched:
                self._upload(pidx_h)
        # 2. nominal / reference states from the previous plan
        nom_s, nom_u, ref_s, ref_us = self.nb.generate_nom_ref_state(st, self.cur_vel, self.ref_speed)
        # 3. PAN
        if any(p.requires_grad for p in self.pan.nrmp_layer.adjust_parameters):
            # LON use (example/LON/LON_corridor.py:94-127): the plan stays connected to the adjust parameters
            gs, gu, gd = self.pan.forward_batch_grad(nom_s, nom_u, ref_s, ref_us, points, velocities, n_points)
            out = dict(self.pan.last_out, opt_s=gs, opt_u=gu, opt_d=None if self.pan.no_obs else gd)
        else:
            out = self.pan.forward_batch(nom_s, nom_u, ref_s, ref_us, points, velocities, n_points)
        opt_u = out["opt_u"]
        done = torch.from_numpy(self.arrived.copy()).to(dev)
        # 4. warm start (arrived robots keep theirs: the reference returns before this line)
        warm = opt_u.detach()
        self.cur_vel = warm if self.cur_vel is None else torch.where(done[:, None, None], self.cur_vel, warm)
        # 5. stop test and action
        md = self.pan.current_min_distance()          # keeps the last value over cycles without points (dune.py:97-98)
        if md is None:
            md = out["min_distance"]
        stop = md < self.collision_threshold
        act = opt_u[:, :, 0]
        if self.robot.kinematics == "omni":
            act = torch.stack([act[:, 0] * torch.cos(act[:, 1]), act[:, 0] * torch.sin(act[:, 1])], dim=1)
        act = torch.where((done | stop)[:, None], torch.zeros_like(act), act)
        info = dict(arrive=done, stop=stop & ~done, opt_u=opt_u, opt_s=out["opt_s"], opt_d=out["opt_d"], min_distance=md,
                    ref_s=ref_s, ref_us=ref_us, point_index=pidx)
        return act, info

    def scan_to_point(self, states, ranges, angle_min, angle_max, range_min, range_max, **kw):
        """B x neupan.scan_to_point (neupan.py:173-222) -> (points, n_points) for `forward`."""
        return scan_to_point_batch(states, ranges, angle_min, angle_max, range_min, range_max, device=self.device, **kw)
