"""Batched front end of a control cycle on the GPU: what the reference does per robot in Python
before it calls PAN.forward (neupan/neupan.py:104-131).

* `scan_to_point_batch` / `scan_to_point_velocity_batch`  -- `neupan.scan_to_point` (neupan.py:173-222)
  and `neupan.scan_to_point_velocity` (:224-281) for B lidar scans at once; the result is the
  `(points [B,2,N], n_points [B])` pair `PAN.forward_batch` takes.
* `NominalBatch.generate_nom_ref_state`  -- `InitialPath.generate_nom_ref_state`
  (neupan/blocks/initial_path.py:68-126) for B robots, each on its own path curve; the result is
  the `(nom_s, nom_u, ref_s, ref_us)` quadruple of `PAN.forward_batch`.

Same argument meaning as the reference; arrays gain a leading scene axis.  There is no CPU
fallback: these call the HIP kernels in libneupan_amd.so (csrc/frontend.hip) through the C ABI.
"""
from __future__ import annotations

import ctypes as C
from math import pi

import numpy as np
import torch

from . import _lib
from ._lib import check

KIN = {"diff": 0, "acker": 1, "omni": 2}


class NpaScanParams(C.Structure):
    _fields_ = [("angle_min", C.c_double), ("angle_max", C.c_double), ("range_min", C.c_double),
                ("range_max", C.c_double), ("state", C.c_double * 3), ("offset", C.c_double * 3),
                ("angle_range", C.c_double * 2), ("down_sample", C.c_int32), ("reserved", C.c_int32)]


_SCAN_DTYPE = np.dtype([("angle_min", "f8"), ("angle_max", "f8"), ("range_min", "f8"), ("range_max", "f8"),
                        ("state", "f8", 3), ("offset", "f8", 3), ("angle_range", "f8", 2), ("down_sample", "i4"),
                        ("reserved", "i4")])
assert _SCAN_DTYPE.itemsize == C.sizeof(NpaScanParams)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _bcast(x, B, n=None):
    a = np.asarray(x, dtype=np.float64)
    shape = (B,) if n is None else (B, n)
    return np.broadcast_to(a, shape).copy()


def _scan(mode, states, ranges, angle_min, angle_max, range_min, range_max, velocities, scan_offset, angle_range,
          down_sample, n_beams, max_points, device):
    lib = _lib.load()
    dev = torch.device(device)
    ranges = torch.as_tensor(np.asarray(ranges, dtype=np.float64) if not isinstance(ranges, torch.Tensor) else ranges)
    ranges = ranges.to(device=dev, dtype=torch.float64).contiguous()
    if ranges.dim() != 2:
        raise ValueError("ranges must be [B, beams]")
    B, R = ranges.shape
    par = np.zeros(B, dtype=_SCAN_DTYPE)
    par["angle_min"], par["angle_max"] = _bcast(angle_min, B), _bcast(angle_max, B)
    par["range_min"], par["range_max"] = _bcast(range_min, B), _bcast(range_max, B)
    par["state"] = _bcast(np.asarray(states, dtype=np.float64).reshape(-1, 3), B, 3)
    par["offset"] = _bcast(scan_offset, B, 3)
    par["angle_range"] = _bcast(angle_range, B, 2)
    par["down_sample"] = np.broadcast_to(np.asarray(down_sample, dtype=np.int32), (B,))
    if (par["down_sample"] < 1).any():
        raise ValueError("down_sample must be >= 1")
    par_d = torch.from_numpy(par.view(np.uint8).reshape(B, -1).copy()).to(dev)
    nb = None
    if n_beams is not None:
        nb = torch.as_tensor(n_beams).to(device=dev, dtype=torch.int32).contiguous()
    vel = None
    if velocities is not None:
        vel = torch.as_tensor(np.asarray(velocities, dtype=np.float64) if not isinstance(velocities, torch.Tensor)
                              else velocities).to(device=dev, dtype=torch.float64).contiguous()
        if tuple(vel.shape) != (B, 2, R):
            raise ValueError("velocities must be [B, 2, beams]")
    N = int(max_points) if max_points else R
    pts = torch.zeros((B, 2, N), dtype=torch.float32, device=dev)
    out_v = torch.zeros((B, 2, N), dtype=torch.float32, device=dev) if mode == 1 else None
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.npa_scan_to_points(B, R, _ptr(ranges), _ptr(vel), _ptr(nb), _ptr(par_d), mode, N, _ptr(pts),
                                     _ptr(out_v), _ptr(cnt), _stream(dev)), "npa_scan_to_points")
    if mode == 1:
        return pts, out_v, cnt
    return pts, cnt


def scan_to_point_batch(states, ranges, angle_min, angle_max, range_min, range_max, scan_offset=(0.0, 0.0, 0.0),
                        angle_range=(-pi, pi), down_sample=1, n_beams=None, max_points=None, device="cuda"):
    """B x `neupan.scan_to_point` (neupan.py:173-222).  states [B,3]; ranges [B,beams] (float64 like
    `np.array(scan["ranges"])`); the scalar scan fields may be scalars or [B] arrays.
    Returns (points [B,2,N] float32, n_points [B] int32): scene b's cloud is points[b, :, :n_points[b]]
    (n_points == 0 where the reference returns None)."""
    return _scan(0, states, ranges, angle_min, angle_max, range_min, range_max, None, scan_offset, angle_range,
                 down_sample, n_beams, max_points, device)


def scan_to_point_velocity_batch(states, ranges, angle_min, angle_max, range_min, range_max, velocities=None,
                                 scan_offset=(0.0, 0.0, 0.0), angle_range=(-pi, pi), down_sample=1, n_beams=None,
                                 max_points=None, device="cuda"):
    """B x `neupan.scan_to_point_velocity` (neupan.py:224-281).  velocities [B,2,beams] or None (zeros,
    :251).  Returns (points, point_velocities, n_points)."""
    return _scan(1, states, ranges, angle_min, angle_max, range_min, range_max, velocities, scan_offset, angle_range,
                 down_sample, n_beams, max_points, device)


class NominalBatch:
    """B x the reference `InitialPath` as far as `generate_nom_ref_state` needs it: every scene's
    CURRENT curve (rows x, y, theta, gear: `InitialPath.cur_curve`, initial_path.py:446-448), its
    `point_index` and `interval`.  Path generation, gear splitting and `check_arrive` stay where
    they are in the reference (host, per robot); `set_curves` uploads their result."""

    def __init__(self, receding, step_time, kinematics, wheelbase=0.0, device="cuda"):
        if kinematics not in KIN:
            raise ValueError("kinematics must be one of diff, acker, omni")
        self.T, self.dt, self.kin, self.L = int(receding), float(step_time), kinematics, float(wheelbase or 0.0)
        self.device = torch.device(device)
        self._lib = _lib.load()
        self.B = 0

    def set_curves(self, curves, intervals, point_index=None):
        """curves: list of B arrays [P_b, 4] (or (4,1) point lists like the reference's curve_list
        entries); intervals: scalar or [B]; point_index: [B] (default 0)."""
        rows, off, ln = [], [], []
        o = 0
        for c in curves:
            a = np.hstack(c).T if isinstance(c, (list, tuple)) else np.asarray(c)
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 4))
            if a.shape[0] < 1:
                raise ValueError("empty curve")
            rows.append(a); off.append(o); ln.append(a.shape[0]); o += a.shape[0]
        self.B = len(rows)
        dev = self.device
        self._path = torch.from_numpy(np.ascontiguousarray(np.concatenate(rows, axis=0))).to(dev).contiguous()
        self._off = torch.tensor(off, dtype=torch.int32, device=dev)
        self._len = torch.tensor(ln, dtype=torch.int32, device=dev)
        self._lens = np.asarray(ln)
        self._interval = torch.from_numpy(_bcast(intervals, self.B)).to(dev)
        self.set_point_index(np.zeros(self.B, dtype=np.int32) if point_index is None else point_index)

    def set_point_index(self, point_index):
        pi_ = np.asarray(point_index, dtype=np.int64).reshape(self.B)
        if (pi_ < 0).any() or (pi_ >= self._lens).any():
            raise IndexError("point_index outside its curve")
        self._pidx = torch.from_numpy(pi_.astype(np.int32)).to(self.device)

    def progress(self, state, close_threshold=0.1, ind_range=10, arrive_threshold=0.1, arrive_index_threshold=1):
        """B x `InitialPath.closest_point` + `check_curve_arrive` (initial_path.py:160-181, :279-287): advances the
        device copy of point_index and returns (point_index [B] int32, min_distance [B] float32, arrived [B] int32)
        device tensors.  Defaults are the reference's (:57-60)."""
        B, dev = self.B, self.device
        st = torch.as_tensor(np.asarray(state, dtype=np.float64) if not isinstance(state, torch.Tensor) else state)
        st = st.to(device=dev, dtype=torch.float64).reshape(B, -1)[:, :3].contiguous()
        md = torch.empty((B,), dtype=torch.float32, device=dev)
        arr = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(self._lib.npa_path_progress(B, _ptr(st), _ptr(self._path), _ptr(self._off), _ptr(self._len), _ptr(self._pidx),
                                              float(close_threshold), int(ind_range), float(arrive_threshold),
                                              int(arrive_index_threshold), _ptr(md), _ptr(arr), _stream(dev)),
                  "npa_path_progress")
        self._hold_p = st
        return self._pidx, md, arr

    def generate_nom_ref_state(self, state, cur_vel_array, ref_speed):
        """state [B,3] (float64); cur_vel_array [B,2,T] float32 tensor/array (PAN's previous opt_u) or
        None for the first call (zeros, neupan.py:73); ref_speed scalar or [B].
        Returns nom_s [B,3,T+1], nom_u [B,2,T], ref_s [B,3,T+1], ref_us [B,T] float32 device tensors."""
        if self.B == 0:
            raise RuntimeError("set_curves first")
        B, T, dev = self.B, self.T, self.device
        st = torch.as_tensor(np.asarray(state, dtype=np.float64) if not isinstance(state, torch.Tensor) else state)
        st = st.to(device=dev, dtype=torch.float64).reshape(B, -1)[:, :3].contiguous()
        vel = None
        if cur_vel_array is not None:
            vel = torch.as_tensor(cur_vel_array).to(device=dev, dtype=torch.float32).contiguous()
            if tuple(vel.shape) != (B, 2, T):
                raise ValueError(f"cur_vel_array must be [{B}, 2, {T}]")
        spd = torch.from_numpy(_bcast(ref_speed, B)).to(dev)
        nom_s = torch.empty((B, 3, T + 1), dtype=torch.float32, device=dev)
        ref_s = torch.empty((B, 3, T + 1), dtype=torch.float32, device=dev)
        nom_u = torch.empty((B, 2, T), dtype=torch.float32, device=dev)
        ref_us = torch.empty((B, T), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(self._lib.npa_nominal_ref_states(B, T, KIN[self.kin], self.dt, self.L, _ptr(st), _ptr(vel), _ptr(spd),
                                                   _ptr(self._path), _ptr(self._off), _ptr(self._len), _ptr(self._pidx),
                                                   _ptr(self._interval), _ptr(nom_s), _ptr(nom_u), _ptr(ref_s),
                                                   _ptr(ref_us), _stream(dev)), "npa_nominal_ref_states")
        self._hold = (st, vel, spd)
        return nom_s, nom_u, ref_s, ref_us
