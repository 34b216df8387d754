"""Generate tests/golden/frontend_*.npz from the UNMODIFIED reference code (build container only).

    python tests/golden/make_golden_frontend.py

* frontend_nominal.npz  outputs of the reference's `InitialPath.generate_nom_ref_state`
                        (neupan/blocks/initial_path.py:68-126) for hand-made paths installed with
                        its own `set_initial_path` (:128-141; gctl's curve generator is not needed)
* frontend_scan.npz     outputs of the reference's `neupan.scan_to_point` / `scan_to_point_velocity`
                        (neupan/neupan.py:173-281), called unbound (they do not touch `self`)

* frontend_progress.npz `InitialPath.closest_point` + `check_curve_arrive` (initial_path.py:160-181, :279-287)
* dune_train_losses.npz the loss terms of the reference's `DUNETrain.train_one_epoch` (dune_train.py:302-366)

The reference functions execute unmodified (numpy 2.2 in this container: NEP-50 scalar promotion).
"""
import copy
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ref_stub_loader import import_reference  # noqa: E402

ref = import_reference()
from neupan.blocks.initial_path import InitialPath  # noqa: E402  (reference)
from neupan.neupan import neupan as RefNeupan  # noqa: E402


# ------------------------------------------------------------------------------ paths
def line_path(n, step, theta=0.0, gear=1.0, x0=0.0, y0=0.0):
    return [np.array([[x0 + i * step * np.cos(theta)], [y0 + i * step * np.sin(theta)], [theta], [gear]]) for i in range(n)]


def arc_path(n, radius, a0, da, gear=1.0):
    """counter-clockwise arc; heading = tangent, deliberately NOT wrapped (crosses +-pi)"""
    pts = []
    for i in range(n):
        a = a0 + i * da
        pts.append(np.array([[radius * np.cos(a)], [radius * np.sin(a)], [a + np.pi / 2], [gear]]))
    return pts


def corner_path(step):
    pts = [np.array([[i * step], [0.0], [0.0], [1.0]]) for i in range(6)]
    pts += [np.array([[5 * step], [(j + 1) * step], [np.pi / 2], [1.0]]) for j in range(8)]
    return pts


def reverse_path(n, step):
    return [np.array([[-i * step], [0.3], [0.0], [-1.0]]) for i in range(n)]


def two_gear_path():
    return line_path(12, 0.5) + reverse_path(10, 0.5)


NOMINAL = []
rng = np.random.default_rng(7)


def add(name, kin, L, T, dt, ref_speed, path, point_index, state, vel, interval=None):
    NOMINAL.append(dict(name=name, kin=kin, L=L, T=T, dt=dt, ref_speed=ref_speed, path=path, point_index=point_index,
                        state=np.asarray(state, dtype=np.float64).reshape(3, 1), vel=vel, interval=interval))


def vel32(T, v, w, jitter=0.2):
    a = np.stack([v + jitter * rng.standard_normal(T), w + 0.5 * jitter * rng.standard_normal(T)])
    return a.astype(np.float32)


add("line_diff_index_mode", "diff", 0.0, 10, 0.1, 4.0, line_path(60, 0.4), 0, [0.1, -0.2, 0.05], vel32(10, 3.5, 0.1))
add("line_diff_first_call_zero_vel", "diff", 0.0, 10, 0.1, 4.0, line_path(60, 0.4), 3, [1.0, 0.3, -0.1], np.zeros((2, 10)))
add("line_diff_end_clamp", "diff", 0.0, 10, 0.1, 4.0, line_path(8, 0.4), 2, [0.9, 0.0, 0.0], vel32(10, 4.0, 0.0))
add("line_diff_fast_ref_2_per_step", "diff", 0.0, 10, 0.1, 8.0, line_path(60, 0.4), 1, [0.2, 0.1, 0.0], vel32(10, 6.0, 0.0))
add("corner_diff_circle_mode", "diff", 0.0, 10, 0.1, 4.0, corner_path(1.0), 0, [0.2, -0.1, 0.1], vel32(10, 3.0, 0.3))
add("corner_diff_circle_mode_end", "diff", 0.0, 12, 0.1, 6.0, corner_path(1.0), 8, [5.1, 2.7, 1.4], vel32(12, 5.0, 0.1))
add("arc_diff_wrap", "diff", 0.0, 10, 0.1, 4.0, arc_path(80, 6.0, 1.0, 0.4 / 6.0), 5, [6.0 * np.cos(1.35), 6.0 * np.sin(1.35), 3.0],
    vel32(10, 4.0, 0.6))
add("arc_diff_circle_mode_wrap", "diff", 0.0, 10, 0.1, 3.0, arc_path(40, 6.0, 1.2, 1.0 / 6.0), 2, [6.0 * np.cos(1.5), 6.0 * np.sin(1.5), -3.1],
    vel32(10, 3.0, 0.5))
add("line_acker_T20", "acker", 3.0, 20, 0.1, 4.0, line_path(80, 0.4, theta=0.3), 4, [1.5, 0.6, 0.25], vel32(20, 4.0, 0.1))
add("reverse_acker", "acker", 3.0, 20, 0.1, 4.0, reverse_path(60, 0.4), 0, [0.0, 0.2, 0.05], vel32(20, -3.0, -0.2))
add("line_omni", "omni", 0.0, 10, 0.1, 4.0, line_path(60, 0.4, theta=-0.4), 0, [0.0, 0.1, 0.0], vel32(10, 3.0, -0.4))
add("two_gear_first_curve", "diff", 0.0, 10, 0.1, 4.0, two_gear_path(), 6, [3.0, 0.1, 0.0], vel32(10, 4.0, 0.0))
add("explicit_interval", "diff", 0.0, 10, 0.1, 4.0, line_path(60, 0.25), 0, [0.0, 0.0, 0.0], vel32(10, 4.0, 0.0), interval=0.3)


def run_nominal():
    out = {}
    names = []
    for c in NOMINAL:
        robot = types.SimpleNamespace(kinematics=c["kin"], L=c["L"], max_speed=[8.0, 1.0])
        ip = InitialPath(c["T"], c["dt"], c["ref_speed"], robot)
        path = copy.deepcopy(c["path"])
        ip.set_initial_path(path)                                   # reference: interval, split by gear
        if c["interval"] is not None:
            ip.interval = c["interval"]
        ip.point_index = c["point_index"]
        curve0 = np.hstack(copy.deepcopy(ip.cur_curve)).T           # (P,4) BEFORE the call mutates theta
        nom_s, nom_u, ref_s, ref_us = ip.generate_nom_ref_state(c["state"].copy(), c["vel"].copy(), c["ref_speed"])
        n = c["name"]
        names.append(n)
        out[n + "/curve"] = curve0
        out[n + "/meta"] = np.array([c["T"], c["dt"], c["ref_speed"], c["L"], c["point_index"], float(ip.interval),
                                     {"diff": 0, "acker": 1, "omni": 2}[c["kin"]]], dtype=np.float64)
        out[n + "/state"] = c["state"].reshape(3)
        out[n + "/vel"] = c["vel"]
        out[n + "/nom_s"] = np.asarray(nom_s, dtype=np.float64)
        out[n + "/ref_s"] = np.asarray(ref_s, dtype=np.float64)
        out[n + "/ref_us"] = np.asarray(ref_us, dtype=np.float64)
        assert np.array_equal(np.asarray(nom_u), c["vel"])
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "frontend_nominal.npz"), **out)
    print("frontend_nominal.npz:", len(names), "cases")


# ------------------------------------------------------------------------------ scans
def make_scan(n, seed, range_max=10.0, range_min=0.1, frac_max=0.25, frac_min=0.05, amin=-np.pi, amax=np.pi):
    r = np.random.default_rng(seed)
    ranges = r.uniform(0.3, range_max - 0.5, n)
    ranges[r.random(n) < frac_max] = range_max                      # no return
    ranges[r.random(n) < frac_min] = range_min * 0.5                # too close
    ranges[::97] = range_min                                        # exactly on the (>, >=) boundary
    ranges[5::131] = range_max - 0.02                               # exactly on the upper boundary
    vel = r.uniform(-1, 1, (2, n))
    return dict(ranges=ranges.tolist(), angle_min=amin, angle_max=amax, range_max=range_max, range_min=range_min,
                velocity=vel)


SCANS = [
    ("full_360", make_scan(360, 1), [1.0, -2.0, 0.7], [0, 0, 0], [-np.pi, np.pi], 1),
    ("offset_1080_ds3", make_scan(1080, 2), [-3.0, 4.0, -2.2], [0.3, -0.1, 0.4], [-np.pi, np.pi], 3),
    ("narrow_fov_720_ds2", make_scan(720, 3, amin=-2.0, amax=2.0), [10.0, 0.5, 3.0], [0.2, 0.0, 0.0], [-1.0, 1.2], 2),
    ("all_filtered_64", dict(ranges=[10.0] * 64, angle_min=-1.0, angle_max=1.0, range_max=10.0, range_min=0.1,
                             velocity=np.zeros((2, 64))), [0.0, 0.0, 0.0], [0, 0, 0], [-np.pi, np.pi], 1),
    ("single_beam", dict(ranges=[2.5], angle_min=0.3, angle_max=0.3, range_max=10.0, range_min=0.1,
                         velocity=np.array([[0.5], [-0.25]])), [1.0, 1.0, 1.0], [0.1, 0.2, 0.3], [-np.pi, np.pi], 1),
    ("big_2048_ds5", make_scan(2048, 4, range_max=30.0), [100.0, -50.0, 0.1], [0.0, 0.0, 3.14], [-3.0, 3.0], 5),
]


def run_scan():
    out = {}
    names = []
    for name, scan, state, off, arange, ds in SCANS:
        st = np.asarray(state, dtype=np.float64).reshape(3, 1)
        p = RefNeupan.scan_to_point(None, st, scan, off, arange, ds)
        pv, vv = RefNeupan.scan_to_point_velocity(None, st, scan, off, arange, ds)
        names.append(name)
        out[name + "/ranges"] = np.asarray(scan["ranges"], dtype=np.float64)
        out[name + "/velocity"] = np.asarray(scan["velocity"], dtype=np.float64)
        out[name + "/meta"] = np.array([scan["angle_min"], scan["angle_max"], scan["range_min"], scan["range_max"],
                                        *state, *off, *arange, ds], dtype=np.float64)
        out[name + "/points"] = np.zeros((2, 0)) if p is None else np.asarray(p, dtype=np.float64)
        out[name + "/points_v"] = np.zeros((2, 0)) if pv is None else np.asarray(pv, dtype=np.float64)
        out[name + "/velocity_v"] = np.zeros((2, 0)) if vv is None else np.asarray(vv, dtype=np.float64)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "frontend_scan.npz"), **out)
    print("frontend_scan.npz:", len(names), "cases")


def run_progress():
    """frontend_progress.npz: InitialPath.closest_point + check_curve_arrive of the reference on random poses"""
    rng = np.random.default_rng(9)
    robot = types.SimpleNamespace(kinematics="diff", L=0.0, max_speed=[8.0, 1.0])
    rows = []
    curves = {}
    for ci, path in enumerate((line_path(30, 0.4), arc_path(40, 6.0, 1.0, 0.4 / 6.0), corner_path(1.0), line_path(6, 0.4))):
        ip = InitialPath(10, 0.1, 4.0, robot)
        ip.set_initial_path(copy.deepcopy(path))
        curve = np.hstack(ip.cur_curve).T.copy()
        curves[f"curve{ci}"] = curve
        n = curve.shape[0]
        for _ in range(40):
            k0 = int(rng.integers(0, n))
            near = int(min(n - 1, k0 + rng.integers(0, 6)))
            st = np.array([[curve[near, 0] + rng.normal(0, 0.15)], [curve[near, 1] + rng.normal(0, 0.15)], [0.0]])
            if rng.random() < 0.2:
                st[0:2, 0] = curve[-1, 0:2] + rng.normal(0, 0.03, 2)            # near the end of the curve
            thr, rng_i = float(rng.choice([0.05, 0.1, 0.3])), int(rng.choice([3, 10, 25]))
            a_thr, a_idx = float(rng.choice([0.1, 0.2])), int(rng.choice([1, 2]))
            ip.point_index = k0
            md = ip.closest_point(st, thr, rng_i)
            arr = ip.check_curve_arrive(st, a_thr, a_idx)
            rows.append([ci, k0, st[0, 0], st[1, 0], thr, rng_i, a_thr, a_idx, ip.point_index, md, float(bool(arr))])
    np.savez_compressed(os.path.join(HERE, "frontend_progress.npz"), rows=np.array(rows, dtype=np.float64), **curves)
    print("frontend_progress.npz:", len(rows), "cases")


def run_dune_train_losses():
    """dune_train_losses.npz: the four loss terms of the reference's DUNETrain.train_one_epoch(validate=True)
    (dune_train.py:302-366) for the shipped diff network with perturbed weights, 600 labelled points, batch
    256, np.random.seed(11) for the per-batch rotation angle."""
    import tempfile

    import torch
    from torch.utils.data import DataLoader

    from neupan.blocks.dune_train import DUNETrain, PointDataset
    from neupan.blocks.obs_point_net import ObsPointNet as RefNet
    from oracle import dune_label_oracle as dl
    G = np.array([[0, -1.6], [2, 0], [0, 1.6], [-2, 0]], dtype=np.float32)
    h = np.full((4, 1), 1.6, dtype=np.float32)
    net = RefNet(2, 4)
    net.load_state_dict(torch.load(os.path.join(HERE, "checkpoints", "diff_robot_default_model_5000.pth"), map_location="cpu"))
    torch.manual_seed(3)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))          # so that the losses are not ~0
    tr = DUNETrain(net, torch.tensor(G), torch.tensor(h), tempfile.mkdtemp())
    P = np.random.default_rng(1).uniform(-25, 25, (600, 2))
    mu, dist = dl.labels(G.astype(np.float64), h.reshape(-1).astype(np.float64), P)
    ds = PointDataset([torch.tensor(P[i].reshape(2, 1), dtype=torch.float32) for i in range(600)],
                      [torch.tensor(mu[i].reshape(4, 1), dtype=torch.float32) for i in range(600)],
                      [torch.tensor(dist[i], dtype=torch.float32) for i in range(600)])
    np.random.seed(11)
    net.eval()
    with torch.no_grad():
        out = tr.train_one_epoch(DataLoader(ds, batch_size=256), True)
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "dune_train_losses.npz"), points=P, mu=mu, dist=dist, losses=np.array(out),
                        G=G, h=h, **{"w/" + k: v for k, v in sd.items()})
    print("dune_train_losses.npz:", out)


def run_pathbook():
    """frontend_pathbook.npz: the host bookkeeping around a path -- `set_initial_path` (average interval, split by
    gear: initial_path.py:128-158, :289-315) and `_ensure_consistent_angles` (:476-497)"""
    out, names = {}, []
    robot = types.SimpleNamespace(kinematics="diff", L=0.0, max_speed=[8.0, 1.0])
    three_gear = line_path(7, 0.5) + reverse_path(5, 0.3) + line_path(4, 0.7, theta=0.4, x0=-1.2, y0=0.3)
    for name, path in (("line", line_path(20, 0.4)), ("two_gear", two_gear_path()), ("three_gear", three_gear),
                       ("corner", corner_path(1.0)), ("arc", arc_path(30, 6.0, 1.0, 0.07)), ("single", line_path(1, 0.4))):
        ip = InitialPath(10, 0.1, 4.0, robot)
        ip.set_initial_path(copy.deepcopy(path))
        names.append(name)
        out[name + "/path"] = np.hstack(path).T
        out[name + "/interval"] = np.array(float(ip.interval))
        out[name + "/curve_len"] = np.array([len(c) for c in ip.curve_list])
        out[name + "/curves"] = np.vstack([np.hstack(c).T for c in ip.curve_list])
        ip2 = InitialPath(10, 0.1, 4.0, robot)
        ip2.initial_path = copy.deepcopy(path)
        ip2._ensure_consistent_angles()
        out[name + "/consistent"] = np.hstack(ip2.initial_path).T
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "frontend_pathbook.npz"), **out)
    print("frontend_pathbook.npz:", len(names), "cases")


if __name__ == "__main__":
    run_pathbook()
    if "--only-pathbook" in sys.argv:
        sys.exit(0)
    run_nominal()
    run_scan()
    run_progress()
    run_dune_train_losses()
