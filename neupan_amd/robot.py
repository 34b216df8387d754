"""Robot geometry/limits holder with the attribute set the reference's `robot` class exposes
to the PAN path (neupan/robot/robot.py:30-71): G, h, kinematics, L, dt, T, speed_bound,
acce_bound, name, vertices.  The cvxpy problem-building half of the reference class
(robot.py:73-236) has no counterpart here: the QP lives in the HIP kernel.

`PAN(robot=...)` accepts either this class or the reference's own robot instance.
"""
from __future__ import annotations

from math import inf

import numpy as np


def _rect_vertices(length, width, wheelbase):
    wb = 0.0 if wheelbase is None else wheelbase
    x0, y0 = -(length - wb) / 2.0, -width / 2.0
    return np.array([[x0, x0 + length, x0 + length, x0], [y0, y0, y0 + width, y0 + width]], dtype=float)


def halfplanes_from_vertices(vertices):
    """G x <= h of a convex polygon given as (2,N) vertices, rows un-normalised, in the
    reference's convention (neupan/util/__init__.py:161-206: edge normal (dy,-dx), CW input
    re-ordered to CCW keeping the first vertex first).  Raises on non-convex input (the
    reference prints and returns None)."""
    v = np.asarray(vertices, dtype=float)
    n = v.shape[1]
    if n < 3:
        raise ValueError("a polygon needs at least 3 vertices")
    nxt, nn = np.roll(v, -1, axis=1), np.roll(v, -2, axis=1)
    turn = (nxt[0] - v[0]) * (nn[1] - nxt[1]) - (nxt[1] - v[1]) * (nn[0] - nxt[0])
    nz = turn[turn != 0]
    if nz.size and not (np.all(nz > 0) or np.all(nz < 0)):
        raise ValueError("robot vertices do not form a convex polygon")
    if not nz.size or nz[0] < 0:
        v = np.concatenate([v[:, :1], v[:, :0:-1]], axis=1)
    e = np.roll(v, -1, axis=1) - v
    G = np.stack([e[1], -e[0]], axis=1)
    h = np.sum(G * v.T, axis=1, keepdims=True)
    return G, h


class Robot:
    def __init__(self, receding=10, step_time=0.1, kinematics=None, vertices=None, max_speed=(inf, inf),
                 max_acce=(inf, inf), wheelbase=None, length=None, width=None, **kwargs):
        if kinematics is None:
            raise ValueError("kinematics is required")
        if kinematics not in ("diff", "acker", "omni"):
            raise ValueError("kinematics currently only supports acker, diff or omni")
        if vertices is not None:
            v = np.array(vertices, dtype=float)
            self.vertices = v.T if isinstance(vertices, list) else v
        else:
            self.vertices = _rect_vertices(length, width, wheelbase)
        self.G, self.h = halfplanes_from_vertices(self.vertices)
        self.T, self.dt, self.L, self.kinematics = receding, step_time, wheelbase, kinematics
        self.max_speed = np.array(max_speed, dtype=float).reshape(2, 1)
        self.max_acce = np.array(max_acce, dtype=float).reshape(2, 1)
        if kinematics == "acker" and self.max_speed[1, 0] >= 1.57:
            self.max_speed[1, 0] = 1.57                      # robot.py:63-66
        self.speed_bound = self.max_speed
        self.acce_bound = self.max_acce * self.dt
        self.name = kwargs.get("name", kinematics + "_robot_default")
