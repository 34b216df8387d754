"""CPU restatement of the two steps in front of PAN.forward.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py / tests/tools/frontend_bench.py -- never by the product path (neupan_amd/).

Pinned against outputs of the UNMODIFIED reference code run in the build container
(tests/golden/make_golden_frontend.py -> tests/golden/frontend_*.npz, see tests/test_frontend.py).

* generate_nom_ref_state   neupan/blocks/initial_path.py:68-126  (+ motion models :388-444,
                           find_interaction_point :183-207, range_cir_seg :209-245,
                           WrapToPi util/__init__.py:98-119)
* path_progress            closest_point + check_curve_arrive, initial_path.py:160-181, :279-287
* scan_to_point            neupan/neupan.py:173-222
* scan_to_point_velocity   neupan/neupan.py:224-281

dtype notes (they decide the low bits and are part of the contract):
  - the reference runs these in numpy float64, EXCEPT that the velocity array handed to the
    rollout is the float32 output of PAN (neupan.py:133-137): with NumPy >= 2 (NEP 50) a float32
    scalar times a Python float stays float32, so `v * cos(phi)` etc. round to float32 before
    the float64 state update (initial_path.py:398-431).  `vel_dtype` selects that behaviour.
  - the reference stores theta corrections back into its path arrays (ref_state is a VIEW of
    the curve point, initial_path.py:110-113, :190-192).  The stored value only ever changes by
    multiples of 2*pi up to rounding, and every later use wraps it again, so this restatement
    keeps the path immutable.
"""
from math import cos, pi, sin, sqrt, tan

import numpy as np


def wrap_to_pi(rad):
    """util/__init__.py:98-119 (loops, not fmod: the low bits depend on it)."""
    while rad > pi:
        rad = rad - 2 * pi
    while rad < -pi:
        rad = rad + 2 * pi
    return rad


def motion_step(kin, state, vel, L, dt):
    """initial_path.py:388-444.  state (3,) float64; vel (2,) float32 or float64."""
    f = type(vel[0])                                  # np.float32 / np.float64: the NEP-50 result type
    v, w = vel[0], vel[1]
    phi = float(state[2])
    if kin == "acker":                                # :398-414
        ds = np.array([v * f(cos(phi)), v * f(sin(phi)), v * f(tan(float(w))) / f(L)], dtype=f)
    elif kin == "diff":                               # :416-432
        ds = np.array([v * f(cos(phi)), v * f(sin(phi)), w], dtype=f)
    elif kin == "omni":                               # :434-444: the literal 0 in the list makes the
        # array float64, so only the two products are rounded to `f`; the dt product is float64
        ds = np.array([float(v * f(cos(float(w)))), float(v * f(sin(float(w)))), 0.0])
        return state + dt * ds
    else:
        raise ValueError(kin)
    return state + (ds * f(dt)).astype(np.float64)


def range_cir_seg(circle, r, sp, ep):
    """initial_path.py:209-245: far intersection of circle (centre, r) with segment sp->ep."""
    d = ep - sp
    if np.linalg.norm(d) == 0:
        return None
    f = sp - circle
    a = d @ d
    b = 2 * f @ d
    c = f @ f - r ** 2
    disc = b ** 2 - 4 * a * c
    if disc < 0:
        return None
    t2 = (-b + sqrt(disc)) / (2 * a)
    if 0 <= t2 <= 1:
        return sp + t2 * d
    return None


def find_interaction_point(curve, ref_xy, ref_index, length):
    """initial_path.py:183-207.  curve: (P,4) float64 rows (x, y, theta, gear)."""
    n = curve.shape[0]
    while True:
        if ref_index > n - 2:
            end = curve[-1]
            return np.array([end[0], end[1], wrap_to_pi(end[2])]), ref_index
        cur, nxt = curve[ref_index], curve[ref_index + 1]
        ip = range_cir_seg(ref_xy, length, cur[0:2], nxt[0:2])
        if ip is not None:
            diff = wrap_to_pi(nxt[2] - cur[2])
            theta = wrap_to_pi(cur[2] + diff / 2)
            return np.array([ip[0], ip[1], theta]), ref_index
        ref_index += 1


def generate_nom_ref_state(curve, point_index, interval, state, cur_vel, ref_speed, T, dt, kin, L):
    """initial_path.py:68-126.  curve (P,4); state (3,); cur_vel (2,T) float32|float64.
    Returns nom_s (3,T+1), nom_u (2,T), ref_s (3,T+1), ref_us (T,) in float64 like the reference
    (the caller casts to float32: neupan.py:121, util np_to_tensor)."""
    curve = np.asarray(curve, dtype=np.float64)
    n = curve.shape[0]
    pre = np.asarray(state, dtype=np.float64).reshape(-1)[:3].copy()
    ref = curve[point_index, 0:3].copy()
    ref_index = int(point_index)
    pre_l, ref_l = [pre.copy()], [ref.copy()]
    gear = [curve[point_index, 3]] * T
    fwd = ref_speed * dt
    for t in range(T):
        pre = motion_step(kin, pre, cur_vel[:, t], L, dt)
        pre_l.append(pre.copy())
        if fwd >= interval:                                        # :93-101
            ref_index = ref_index + int(fwd / interval)
            if ref_index > n - 1:
                ref_index = n - 1
                gear[t] = 0
            ref = curve[ref_index, 0:3].copy()
        else:                                                      # :103-109
            ref, ref_index = find_interaction_point(curve, ref[0:2].copy(), ref_index, fwd)
            if ref_index > n - 1:
                gear[t] = 0
        ref[2] = pre[2] + wrap_to_pi(ref[2] - pre[2])              # :111-112
        ref_l.append(ref.copy())
    nom_s = np.stack(pre_l, axis=1)
    ref_s = np.stack(ref_l, axis=1)
    ref_us = np.array(gear, dtype=np.float64) * ref_speed
    return nom_s, np.asarray(cur_vel), ref_s, ref_us


def path_progress(curve, point_index, state, close_threshold=0.1, ind_range=10, arrive_threshold=0.1,
                  arrive_index_threshold=1):
    """InitialPath.closest_point (initial_path.py:160-181) followed by check_curve_arrive (:279-287), as
    check_arrive runs them (:247-252).  Returns (new point_index, min distance, curve arrived)."""
    curve = np.asarray(curve, dtype=np.float64)
    st = np.asarray(state, dtype=np.float64).reshape(-1)
    n = curve.shape[0]
    min_dis = np.inf
    start, end = max(int(point_index), 0), min(int(point_index) + int(ind_range), n)
    pidx = int(point_index)
    for index in range(start, end):
        dis = sqrt((st[0] - curve[index, 0]) ** 2 + (st[1] - curve[index, 1]) ** 2)      # util distance :133
        if dis < min_dis:
            min_dis = dis
            pidx = index
            if dis < close_threshold:
                break
    arrive_distance = float(np.linalg.norm(st[0:2] - curve[-1, 0:2]))
    arrived = arrive_distance < arrive_threshold and pidx >= (n - arrive_index_threshold - 2)
    return pidx, min_dis, bool(arrived)


def _linspace(a, b, n):
    """numpy.linspace(a, b, n) as it is evaluated for scalars: i*step + a, last element = b."""
    if n == 1:
        return np.array([float(a)])
    step = (b - a) / (n - 1)
    y = np.arange(0, n, dtype=np.float64) * step + a
    y[-1] = b
    return y


def _rot(theta):
    return np.array([[cos(theta), -sin(theta)], [sin(theta), cos(theta)]])


def scan_to_point(state, ranges, angle_min, angle_max, range_min, range_max, scan_offset=(0.0, 0.0, 0.0),
                  angle_range=(-pi, pi), down_sample=1):
    """neupan.py:173-222.  Returns (2,n) float64 or None."""
    ranges = np.asarray(ranges, dtype=np.float64)
    angles = _linspace(angle_min, angle_max, len(ranges))
    keep = (ranges < (range_max - 0.02)) & (ranges > range_min) & (angles > angle_range[0]) & (angles < angle_range[1])
    if not keep.any():
        return None
    r, a = ranges[keep], angles[keep]
    pts = np.vstack([r * np.array([cos(x) for x in a]), r * np.array([sin(x) for x in a])])
    off = np.asarray(scan_offset, dtype=np.float64)
    temp = _rot(off[2]) @ pts + off[0:2, None]                     # :214-215
    st = np.asarray(state, dtype=np.float64).reshape(-1)
    return (_rot(st[2]) @ temp + st[0:2, None])[:, ::down_sample]  # :217-218


def scan_to_point_velocity(state, ranges, angle_min, angle_max, range_min, range_max, velocity=None,
                           scan_offset=(0.0, 0.0, 0.0), angle_range=(-pi, pi), down_sample=1):
    """neupan.py:224-281.  Differences from scan_to_point that are the reference's own:
    `>= range_min` (:256) and the sensor offset applied as R^T (p - t) (:268-270)."""
    ranges = np.asarray(ranges, dtype=np.float64)
    angles = _linspace(angle_min, angle_max, len(ranges))
    vel = np.zeros((2, len(ranges))) if velocity is None else np.asarray(velocity, dtype=np.float64)
    keep = (ranges < (range_max - 0.02)) & (ranges >= range_min) & (angles > angle_range[0]) & (angles < angle_range[1])
    if not keep.any():
        return None, None
    r, a = ranges[keep], angles[keep]
    pts = np.vstack([r * np.array([cos(x) for x in a]), r * np.array([sin(x) for x in a])])
    off = np.asarray(scan_offset, dtype=np.float64)
    temp = _rot(off[2]).T @ (pts - off[0:2, None])
    st = np.asarray(state, dtype=np.float64).reshape(-1)
    return (_rot(st[2]) @ temp + st[0:2, None])[:, ::down_sample], vel[:, keep][:, ::down_sample]
