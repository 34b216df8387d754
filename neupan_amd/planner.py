"""`neupan`: the reference's user-facing planner class (neupan/neupan.py:30-420) for ONE robot, with every
compute step on the device -- the class a script written against the reference constructs
(`neupan.init_from_yaml("planner.yaml")`) and calls once per control cycle (`forward(state, points)`).

It is `FleetPlanner` with B = 1 plus the host bookkeeping of the reference's `InitialPath` that is not
compute (way-points -> initial path, gear split, loop/arrival flags, initial_path.py:30-66, :247-386):

    reference member                               here
    ---------------------------------------------  -------------------------------------------------------
    init_from_yaml            neupan.py:88-102     same keys (robot / ipath / pan / adjust / train)
    forward                   neupan.py:104-166    FleetPlanner.forward (npa_path_progress,
                                                   npa_nominal_ref_states, npa_forward_batch)
    scan_to_point(_velocity)  neupan.py:173-281    npa_scan_to_points
    set_initial_path, set_initial_path_from_state, update_initial_path_from_goal / _from_waypoints,
    set_reference_speed, update_adjust_parameters, reset, train_dune       neupan.py:283-359
    min_distance, dune_points, nrmp_points, initial_path, adjust_parameters, waypoints, opt_trajectory,
    ref_trajectory            neupan.py:361-420

Curves between way-points are the job of the third-party `gctl` package in the reference
(`curve_generator.generate_curve`, initial_path.py:337-339; pinned gctl==1.2, absent from this image and
from /root/reference).  `curve_style: line` is generated here (equally spaced points of each segment, the
headings then set by the reference's own `_ensure_consistent_angles`, initial_path.py:476-497) -- PARITY
UNPINNED against gctl's sampling; `dubins` / `reeds` need gctl and are delegated to it when it imports.
"""
from __future__ import annotations

import os
import sys
from math import atan2, ceil, cos, hypot, sin, tan

import numpy as np
import torch
import yaml

from .fleet import FleetPlanner
from .frontend import scan_to_point_batch, scan_to_point_velocity_batch
from .robot import Robot


def _find_file(name, extra_roots=()):
    """util.file_check (util/__init__.py:58-94): as given, next to the script, in the cwd, then below the
    package root -- here: below any ancestor of the YAML file that names it."""
    if name is None or name == "None":
        return None
    cands = [name, os.path.join(sys.path[0], name), os.path.join(os.getcwd(), name)]
    for root in extra_roots:
        d = os.path.abspath(root)
        while True:
            cands.append(os.path.join(d, name))
            up = os.path.dirname(d)
            if up == d:
                break
            d = up
    for c in cands:
        if os.path.exists(c):
            return os.path.abspath(c)
    raise FileNotFoundError("File not found: " + str(name))


def line_curve(waypoints, interval):
    """Straight segments between way-points sampled every `interval` metres; points are (4,1): x, y, heading
    of the segment, gear +1.  (gctl's 'line' style; see the module docstring on parity.)"""
    wp = [np.asarray(w, dtype=np.float64).reshape(-1)[:3] for w in waypoints]
    out = []
    for a, b in zip(wp, wp[1:]):
        d = hypot(b[0] - a[0], b[1] - a[1])
        if d == 0.0:
            continue                                       # the start state repeated as first way-point
        n = max(int(ceil(d / interval - 1e-9)), 1) if interval > 0 else 1
        th = atan2(b[1] - a[1], b[0] - a[0])
        for i in range(n):
            f = i / n
            out.append(np.array([[a[0] + f * (b[0] - a[0])], [a[1] + f * (b[1] - a[1])], [th], [1.0]]))
    last = wp[-1]
    th = out[-1][2, 0] if out else float(last[2])
    out.append(np.array([[last[0]], [last[1]], [th], [1.0]]))
    return out


def _consistent_angles(path):
    """initial_path.py:476-497: heading of point i = direction to point i+1; the last repeats."""
    if path is None or len(path) < 2:
        return
    for p, q in zip(path, path[1:]):
        p[2, 0] = atan2(q[1, 0] - p[1, 0], q[0, 0] - p[0, 0])
    path[-1][2, 0] = path[-2][2, 0]


def generate_curve(style, waypoints, interval, min_radius):
    try:                                                   # the reference's generator when it is installed
        from gctl import curve_generator
        return curve_generator().generate_curve(style, waypoints, interval, min_radius, True)
    except ImportError:
        pass
    if style != "line":
        raise NotImplementedError(f"curve_style '{style}' needs the gctl package (pip gctl==1.2); without it only "
                                  "'line' is generated here -- or hand a finished path to set_initial_path()")
    return line_curve(waypoints, interval)


class neupan(torch.nn.Module):
    def __init__(self, receding=10, step_time=0.1, ref_speed=4.0, device="cuda", robot_kwargs=None, ipath_kwargs=None,
                 pan_kwargs=None, adjust_kwargs=None, train_kwargs=None, **kwargs):
        super().__init__()
        robot_kwargs, ipath_kwargs, pan_kwargs = dict(robot_kwargs or {}), dict(ipath_kwargs or {}), dict(pan_kwargs or {})
        self.T, self.dt, self.ref_speed = int(receding), float(step_time), float(ref_speed)
        if str(device) == "cpu":
            # the shipped planner.yaml files say device: 'cpu'; there is no CPU path in this build
            device = "cuda"
        self.collision_threshold = kwargs.get("collision_threshold", 0.1)
        self.time_print = kwargs.get("time_print", False)
        self.robot = Robot(receding, step_time, **robot_kwargs)
        ck = pan_kwargs.get("dune_checkpoint")
        if ck is not None and ck != "None":
            pan_kwargs["dune_checkpoint"] = _find_file(ck, kwargs.get("_search_roots", ()))
        pan_kwargs["adjust_kwargs"] = adjust_kwargs
        pan_kwargs["train_kwargs"] = train_kwargs
        self.dune_train_kwargs = train_kwargs
        # InitialPath's settings (initial_path.py:41-66)
        ip = ipath_kwargs
        self.waypoints_ = [np.c_[p] if isinstance(p, list) else p for p in (ip.get("waypoints") or [])]
        self.loop, self.curve_style = bool(ip.get("loop", False)), ip.get("curve_style", "line")
        if "min_radius" in ip:
            self.min_radius = ip["min_radius"]
        elif self.robot.kinematics == "acker":
            self.min_radius = self.robot.L / tan(float(self.robot.max_speed[1, 0]))      # initial_path.py:465-474
        else:
            self.min_radius = 0.0
        self.interval = ip.get("interval", self.dt * self.ref_speed)
        self.fleet = FleetPlanner(self.robot, receding, step_time, ref_speed, self.collision_threshold, device,
                                  close_threshold=ip.get("close_threshold", 0.1), ind_range=ip.get("ind_range", 10),
                                  arrive_threshold=ip.get("arrive_threshold", 0.1),
                                  arrive_index_threshold=ip.get("arrive_index_threshold", 1), loop=self.loop, **pan_kwargs)
        self.pan = self.fleet.pan
        self.device = self.fleet.device
        self.initial_path_ = None
        self.cur_vel_array = np.zeros((2, self.T))
        self.info = {"stop": False, "arrive": False, "collision": False}

    @classmethod
    def init_from_yaml(cls, yaml_file, **kwargs):
        """neupan.py:88-102; relative file names inside the YAML are also looked up below the YAML's ancestors."""
        path = _find_file(yaml_file)
        with open(path, "r") as f:
            config = yaml.safe_load(f)
        config.update(kwargs)
        config["robot_kwargs"] = config.pop("robot", dict())
        config["ipath_kwargs"] = config.pop("ipath", dict())
        config["pan_kwargs"] = config.pop("pan", dict())
        config["adjust_kwargs"] = config.pop("adjust", dict())
        config["train_kwargs"] = config.pop("train", dict())
        config.setdefault("_search_roots", (os.path.dirname(path),))
        return cls(**config)

    # ------------------------------------------------------------------ initial path (host bookkeeping)
    def _install(self, path):
        self.initial_path_ = path
        self.fleet.set_paths([path])
        self.fleet.cur_vel = torch.from_numpy(self.cur_vel_array.astype(np.float32)).to(self.device)[None]

    def set_initial_path(self, path):
        """neupan.py:296-303 / initial_path.py:128-142: path = list of (4,1) x, y, theta, gear."""
        self._generated = False
        self._install(path)

    def _path_from(self, waypoints):
        path = generate_curve(self.curve_style, waypoints, self.interval, self.min_radius)
        if self.curve_style == "line":
            _consistent_angles(path)
        self._generated = True
        self._install(path)
        self.fleet.intervals = [self.interval]             # a generated path keeps the configured interval
        self.fleet._upload()

    def set_initial_path_from_state(self, state):
        """neupan.py:305-312 -> init_check (initial_path.py:345-361)."""
        if self.initial_path_ is None:
            st = np.asarray(state, dtype=np.float64).reshape(-1, 1)[0:3]
            assert len(self.waypoints_) > 0, "Error: waypoints are not set"
            wps = [st] + list(self.waypoints_)
            if self.loop:
                wps = wps + [wps[0]]
            self.waypoints_ = wps
            self._path_from(wps)

    def update_initial_path_from_goal(self, start, goal):
        wps = [start, goal, start] if self.loop else [start, goal]
        self._path_from(wps)
        self.waypoints_ = wps

    def update_initial_path_from_waypoints(self, waypoints):
        self._path_from(waypoints)
        self.waypoints_ = waypoints

    def set_reference_speed(self, speed):
        self.ref_speed = self.fleet.ref_speed = float(speed)

    def update_adjust_parameters(self, **kwargs):
        self.pan.nrmp_layer.update_adjust_parameters_value(**kwargs)

    def reset(self):
        """neupan.py:287-294"""
        self.info["stop"] = self.info["arrive"] = False
        self.cur_vel_array = np.zeros_like(self.cur_vel_array)
        if self.initial_path_ is not None:
            self._install(self.initial_path_)
            if getattr(self, "_generated", False):
                # only set_initial_path averages the interval (initial_path.py:138); a path generated from waypoints keeps
                # the configured one across reset(), as the reference's InitialPath does
                self.fleet.intervals = [self.interval]
                self.fleet._upload()

    def train_dune(self):
        self.pan.dune_layer.train_dune(self.dune_train_kwargs)

    # ------------------------------------------------------------------ one control cycle
    def forward(self, state, points, velocities=None):
        """state (3,1)+; points (2,N)|None; velocities (2,N)|None  ->  (action (2,1) numpy, info)."""
        state = np.asarray(state, dtype=np.float64)
        assert state.shape[0] >= 3
        self.set_initial_path_from_state(state)
        dev = self.device
        pts = vel = None
        if points is not None and np.asarray(points).size:
            pts = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float32), device=dev)[None]
            if velocities is not None:
                vel = torch.as_tensor(np.ascontiguousarray(velocities, dtype=np.float32), device=dev)[None]
        act, fi = self.fleet.forward(state.reshape(-1)[None, :3], pts, vel)
        if bool(fi["arrive"][0]):
            self.info["arrive"] = True
            return np.zeros((2, 1)), self.info
        opt_s, opt_u = fi["opt_s"][0], fi["opt_u"][0]
        opt_s_np, opt_u_np = opt_s.detach().cpu().numpy(), opt_u.detach().cpu().numpy()       # tensor_to_np
        self.cur_vel_array = opt_u_np
        ref_s_np = fi["ref_s"][0].cpu().numpy()
        self.info["state_tensor"], self.info["vel_tensor"] = opt_s, opt_u
        self.info["distance_tensor"] = None if fi["opt_d"] is None else fi["opt_d"][0]
        self.info["ref_state_tensor"], self.info["ref_speed_tensor"] = fi["ref_s"][0], fi["ref_us"][0]
        self.info["ref_state_list"] = [c[:, np.newaxis] for c in ref_s_np.T]
        self.info["opt_state_list"] = [c[:, np.newaxis] for c in opt_s_np.T]
        if self.check_stop():
            self.info["stop"] = True
            return np.zeros((2, 1)), self.info
        self.info["stop"] = False
        action = opt_u_np[:, 0:1]
        if self.robot.kinematics == "omni":                 # neupan.py:158-164
            v, th = float(action[0, 0]), float(action[1, 0])
            self.info["omni_linear_speed"], self.info["omni_orientation"] = v, th
            action = np.array([[v * cos(th)], [v * sin(th)]])
        return action, self.info

    def check_stop(self):
        return bool(self.min_distance < self.collision_threshold)

    # ------------------------------------------------------------------ lidar scan -> points
    def scan_to_point(self, state, scan, scan_offset=(0, 0, 0), angle_range=(-np.pi, np.pi), down_sample=1):
        """neupan.py:173-222: scan = dict(ranges, angle_min, angle_max, range_min, range_max) -> (2,n) numpy | None."""
        st = np.asarray(state, dtype=np.float64).reshape(-1)[None, :3]
        r = np.asarray(scan["ranges"], dtype=np.float64)[None]
        pts, n = scan_to_point_batch(st, r, scan["angle_min"], scan["angle_max"], scan["range_min"], scan["range_max"],
                                     scan_offset=scan_offset, angle_range=angle_range, down_sample=down_sample,
                                     device=self.device)
        n = int(n[0])
        return None if n == 0 else pts[0, :, :n].cpu().numpy()

    def scan_to_point_velocity(self, state, scan, scan_offset=(0, 0, 0), angle_range=(-np.pi, np.pi), down_sample=1):
        """neupan.py:224-281 (scan additionally carries `velocity` (2, n_beams)) -> (points, velocities) | (None, None)."""
        st = np.asarray(state, dtype=np.float64).reshape(-1)[None, :3]
        r = np.asarray(scan["ranges"], dtype=np.float64)[None]
        v = scan.get("velocity")
        v = None if v is None else np.asarray(v, dtype=np.float64)[None]
        pts, vel, n = scan_to_point_velocity_batch(st, r, scan["angle_min"], scan["angle_max"], scan["range_min"],
                                                   scan["range_max"], v, scan_offset=scan_offset, angle_range=angle_range,
                                                   down_sample=down_sample, device=self.device)
        n = int(n[0])
        if n == 0:
            return None, None
        return pts[0, :, :n].cpu().numpy(), vel[0, :, :n].cpu().numpy()

    # ------------------------------------------------------------------ properties other code reads
    @property
    def min_distance(self):
        return self.pan.min_distance

    @property
    def dune_points(self):
        return self.pan.dune_points

    @property
    def nrmp_points(self):
        return self.pan.nrmp_points

    @property
    def initial_path(self):
        return self.initial_path_

    @property
    def adjust_parameters(self):
        return self.pan.nrmp_layer.adjust_parameters

    @property
    def waypoints(self):
        return self.waypoints_

    @property
    def opt_trajectory(self):
        return self.info["opt_state_list"]

    @property
    def ref_trajectory(self):
        return self.info["ref_state_list"]
