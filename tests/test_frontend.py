"""Front-end steps of a control cycle: nominal/reference rollout and
lidar scan -> point cloud.  CPU part: the oracle restatement against vectors produced by the
unmodified reference code (tests/golden/make_golden_frontend.py).  GPU part (-m gpu): the HIP
kernels through the C ABI against the same vectors and against the oracle on larger batches.

Tolerances: the reference computes in float64 and casts to float32 at the PAN boundary
(neupan.py:121).  The oracle must agree to 1e-12 (same operations; only libm/BLAS rounding may
differ); the HIP path emits float32 and must agree to 1 float32 ulp of the value's magnitude."""
import os

import numpy as np
import pytest

from oracle import frontend_oracle as fo

HERE = os.path.dirname(os.path.abspath(__file__))
KIN = {0: "diff", 1: "acker", 2: "omni"}


def _nominal_cases():
    z = np.load(os.path.join(HERE, "golden", "frontend_nominal.npz"))
    for n in z["names"]:
        n = str(n)
        T, dt, ref_speed, L, pidx, interval, kin = z[n + "/meta"]
        yield n, dict(curve=z[n + "/curve"], T=int(T), dt=float(dt), ref_speed=float(ref_speed), L=float(L),
                      point_index=int(pidx), interval=float(interval), kin=KIN[int(kin)], state=z[n + "/state"],
                      vel=z[n + "/vel"], nom_s=z[n + "/nom_s"], ref_s=z[n + "/ref_s"], ref_us=z[n + "/ref_us"])


def _scan_cases():
    z = np.load(os.path.join(HERE, "golden", "frontend_scan.npz"))
    for n in z["names"]:
        n = str(n)
        m = z[n + "/meta"]
        yield n, dict(ranges=z[n + "/ranges"], velocity=z[n + "/velocity"], angle_min=m[0], angle_max=m[1],
                      range_min=m[2], range_max=m[3], state=m[4:7], offset=m[7:10], angle_range=m[10:12],
                      down_sample=int(m[12]), points=z[n + "/points"], points_v=z[n + "/points_v"],
                      velocity_v=z[n + "/velocity_v"])


def _progress_cases():
    z = np.load(os.path.join(HERE, "golden", "frontend_progress.npz"))
    return z["rows"], [z[f"curve{i}"] for i in range(4)]


NOMINAL = list(_nominal_cases())
SCANS = list(_scan_cases())


@pytest.mark.parametrize("name,c", NOMINAL, ids=[n for n, _ in NOMINAL])
def test_oracle_nominal_vs_reference(name, c):
    nom_s, nom_u, ref_s, ref_us = fo.generate_nom_ref_state(c["curve"], c["point_index"], c["interval"], c["state"],
                                                            c["vel"], c["ref_speed"], c["T"], c["dt"], c["kin"], c["L"])
    assert np.abs(nom_s - c["nom_s"]).max() <= 1e-12
    assert np.abs(ref_s - c["ref_s"]).max() <= 1e-12
    assert np.array_equal(ref_us, c["ref_us"])


@pytest.mark.parametrize("name,c", SCANS, ids=[n for n, _ in SCANS])
def test_oracle_scan_vs_reference(name, c):
    p = fo.scan_to_point(c["state"], c["ranges"], c["angle_min"], c["angle_max"], c["range_min"], c["range_max"],
                         c["offset"], c["angle_range"], c["down_sample"])
    pv, vv = fo.scan_to_point_velocity(c["state"], c["ranges"], c["angle_min"], c["angle_max"], c["range_min"],
                                       c["range_max"], c["velocity"], c["offset"], c["angle_range"], c["down_sample"])
    if c["points"].shape[1] == 0:
        assert p is None and pv is None and vv is None
        return
    assert p.shape == c["points"].shape and pv.shape == c["points_v"].shape
    assert np.abs(p - c["points"]).max() <= 1e-12 * max(1.0, np.abs(c["points"]).max())
    assert np.abs(pv - c["points_v"]).max() <= 1e-12 * max(1.0, np.abs(c["points_v"]).max())
    assert np.array_equal(vv, c["velocity_v"])


def test_oracle_path_progress_vs_reference():
    rows, curves = _progress_cases()
    for r in rows:
        ci, k0, sx, sy, thr, rng_i, a_thr, a_idx, want_idx, want_md, want_arr = r
        idx, md, arr = fo.path_progress(curves[int(ci)], int(k0), [sx, sy, 0.0], thr, int(rng_i), a_thr, int(a_idx))
        assert idx == int(want_idx) and bool(arr) == bool(want_arr)
        assert (np.isinf(md) and np.isinf(want_md)) or abs(md - want_md) <= 1e-13


# ------------------------------------------------------------------------------ HIP path (-m gpu)
def _ulp32(ref):
    """1 float32 ulp at the magnitude of the largest value of `ref` (the reference casts float64 ->
    float32 at the PAN boundary; libm may differ from the device's by 1 float64 ulp before that)."""
    return float(np.spacing(np.float32(max(1.0, np.abs(ref).max()))))


@pytest.mark.gpu
def test_hip_nominal_vs_reference_vectors():
    """all golden cases of one kinematics/horizon in ONE batched call each"""
    from neupan_amd.frontend import NominalBatch
    groups = {}
    for name, c in NOMINAL:
        groups.setdefault((c["kin"], c["T"], c["dt"], c["L"]), []).append((name, c))
    for (kin, T, dt, L), cases in groups.items():
        nb = NominalBatch(T, dt, kin, L)
        nb.set_curves([c["curve"] for _, c in cases], [c["interval"] for _, c in cases], [c["point_index"] for _, c in cases])
        vel = np.stack([c["vel"].astype(np.float32) for _, c in cases])
        out = nb.generate_nom_ref_state(np.stack([c["state"] for _, c in cases]), vel, [c["ref_speed"] for _, c in cases])
        nom_s, nom_u, ref_s, ref_us = (o.cpu().numpy() for o in out)
        for j, (name, c) in enumerate(cases):
            assert np.abs(nom_s[j] - c["nom_s"].astype(np.float32)).max() <= _ulp32(c["nom_s"]), name
            assert np.abs(ref_s[j] - c["ref_s"].astype(np.float32)).max() <= _ulp32(c["ref_s"]), name
            assert np.array_equal(ref_us[j], c["ref_us"].astype(np.float32)), name
            assert np.array_equal(nom_u[j], c["vel"].astype(np.float32)), name


@pytest.mark.gpu
def test_hip_nominal_first_call_without_velocities():
    from neupan_amd.frontend import NominalBatch
    name, c = [x for x in NOMINAL if x[0] == "line_diff_first_call_zero_vel"][0]
    nb = NominalBatch(c["T"], c["dt"], c["kin"], c["L"])
    nb.set_curves([c["curve"]], c["interval"], [c["point_index"]])
    nom_s, nom_u, ref_s, ref_us = (o.cpu().numpy() for o in nb.generate_nom_ref_state(c["state"][None], None, c["ref_speed"]))
    assert np.array_equal(nom_s[0], c["nom_s"].astype(np.float32))
    assert np.abs(ref_s[0] - c["ref_s"].astype(np.float32)).max() <= _ulp32(c["ref_s"])
    assert (nom_u == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kin,L,T", [("diff", 0.0, 10), ("acker", 3.0, 20), ("omni", 0.0, 10)])
def test_hip_nominal_large_batch_vs_oracle(kin, L, T):
    """4096 robots on random polylines, both reference-sampling modes, against the oracle"""
    from neupan_amd.frontend import NominalBatch
    rng = np.random.default_rng(11)
    B, dt = 4096, 0.1
    curves, intervals, pidx, states, speeds = [], [], [], [], []
    for b in range(B):
        n = int(rng.integers(6, 60))
        step = float(rng.uniform(0.2, 1.2))
        head = np.cumsum(rng.uniform(-0.25, 0.25, n)) + rng.uniform(-3, 3)
        xy = np.cumsum(np.stack([step * np.cos(head), step * np.sin(head)], axis=1), axis=0)
        curves.append(np.column_stack([xy, head, np.full(n, 1.0 if b % 3 else -1.0)]))
        intervals.append(step)
        k = int(rng.integers(0, n - 1))
        pidx.append(k)
        states.append([xy[k, 0] + rng.normal(0, 0.1), xy[k, 1] + rng.normal(0, 0.1), head[k] + rng.normal(0, 0.2)])
        speeds.append(float(rng.choice([2.0, 4.0, 8.0])))
    vel = np.stack([rng.uniform(-4, 6, (B, T)), rng.uniform(-1, 1, (B, T))], axis=1).astype(np.float32)
    nb = NominalBatch(T, dt, kin, L)
    nb.set_curves(curves, intervals, pidx)
    nom_s, nom_u, ref_s, ref_us = (o.cpu().numpy() for o in nb.generate_nom_ref_state(np.asarray(states), vel, speeds))
    bad = 0
    for b in range(0, B, 7):
        o = fo.generate_nom_ref_state(curves[b], pidx[b], intervals[b], np.asarray(states[b]), vel[b], speeds[b], T, dt, kin, L)
        ok = (np.abs(nom_s[b] - o[0].astype(np.float32)).max() <= _ulp32(o[0]) and
              np.abs(ref_s[b] - o[2].astype(np.float32)).max() <= _ulp32(o[2]) and
              np.array_equal(ref_us[b], o[3].astype(np.float32)))
        bad += 0 if ok else 1
    assert bad == 0


@pytest.mark.gpu
def test_hip_scan_vs_reference_vectors():
    from neupan_amd.frontend import scan_to_point_batch, scan_to_point_velocity_batch
    for name, c in SCANS:
        kw = dict(scan_offset=c["offset"], angle_range=c["angle_range"], down_sample=c["down_sample"])
        a = (c["state"][None], c["ranges"][None], c["angle_min"], c["angle_max"], c["range_min"], c["range_max"])
        pts, cnt = scan_to_point_batch(*a, **kw)
        n = int(cnt.cpu()[0])
        assert n == c["points"].shape[1], name
        if n:
            assert np.abs(pts.cpu().numpy()[0, :, :n] - c["points"].astype(np.float32)).max() <= _ulp32(c["points"]), name
        pts, vel, cnt = scan_to_point_velocity_batch(*a, velocities=c["velocity"][None], **kw)
        n = int(cnt.cpu()[0])
        assert n == c["points_v"].shape[1], name
        if n:
            assert np.abs(pts.cpu().numpy()[0, :, :n] - c["points_v"].astype(np.float32)).max() <= _ulp32(c["points_v"]), name
            assert np.array_equal(vel.cpu().numpy()[0, :, :n], c["velocity_v"].astype(np.float32)), name


@pytest.mark.gpu
def test_hip_scan_velocity_ragged_batch_vs_oracle():
    from neupan_amd.frontend import scan_to_point_velocity_batch
    rng = np.random.default_rng(6)
    B, R = 128, 900
    nb = rng.integers(1, R + 1, B).astype(np.int32)
    ranges = rng.uniform(0.05, 8.0, (B, R))
    ranges[rng.random((B, R)) < 0.15] = 8.0
    ranges[:, ::50] = 0.1                                              # exactly range_min: kept by this variant (>=)
    vel = rng.uniform(-1, 1, (B, 2, R))
    states = np.column_stack([rng.uniform(-9, 9, B), rng.uniform(-9, 9, B), rng.uniform(-3.1, 3.1, B)])
    off = np.column_stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B), rng.uniform(-1, 1, B)])
    ds = rng.integers(1, 4, B).astype(np.int32)
    pts, pv, cnt = scan_to_point_velocity_batch(states, ranges, -3.0, 3.0, 0.1, 8.0, velocities=vel, scan_offset=off,
                                                down_sample=ds, n_beams=nb)
    pts, pv, cnt = pts.cpu().numpy(), pv.cpu().numpy(), cnt.cpu().numpy()
    for b in range(0, B, 2):
        o, ov = fo.scan_to_point_velocity(states[b], ranges[b, :nb[b]], -3.0, 3.0, 0.1, 8.0, vel[b][:, :nb[b]], off[b],
                                          (-np.pi, np.pi), int(ds[b]))
        if o is None:
            assert cnt[b] == 0
            continue
        assert cnt[b] == o.shape[1]
        assert np.abs(pts[b, :, :cnt[b]] - o.astype(np.float32)).max() <= _ulp32(o)
        assert np.array_equal(pv[b, :, :cnt[b]], ov.astype(np.float32))


@pytest.mark.gpu
def test_hip_scan_ragged_batch_vs_oracle():
    """512 scans of different lengths / poses / fields of view in one call; feeds PAN.forward_batch's
    (points, n_points) layout"""
    from neupan_amd.frontend import scan_to_point_batch
    rng = np.random.default_rng(5)
    B, R = 512, 1500
    nb = rng.integers(1, R + 1, B).astype(np.int32)
    nb[0], nb[1] = R, 1
    ranges = rng.uniform(0.05, 12.0, (B, R))
    ranges[rng.random((B, R)) < 0.2] = 12.0
    states = np.column_stack([rng.uniform(-50, 50, B), rng.uniform(-50, 50, B), rng.uniform(-3.1, 3.1, B)])
    amin = rng.uniform(-3.14, -1.0, B); amax = rng.uniform(1.0, 3.14, B)
    ds = rng.integers(1, 5, B).astype(np.int32)
    off = np.column_stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B), rng.uniform(-1, 1, B)])
    arange = np.column_stack([rng.uniform(-3.2, -0.5, B), rng.uniform(0.5, 3.2, B)])
    pts, cnt = scan_to_point_batch(states, ranges, amin, amax, 0.1, 12.0, off, arange, ds, n_beams=nb)
    pts, cnt = pts.cpu().numpy(), cnt.cpu().numpy()
    for b in range(0, B, 3):
        o = fo.scan_to_point(states[b], ranges[b, :nb[b]], amin[b], amax[b], 0.1, 12.0, off[b], arange[b], int(ds[b]))
        if o is None:
            assert cnt[b] == 0
            continue
        assert cnt[b] == o.shape[1]
        assert np.abs(pts[b, :, :cnt[b]] - o.astype(np.float32)).max() <= _ulp32(o)


@pytest.mark.gpu
def test_hip_path_progress_vs_reference_vectors():
    from neupan_amd.frontend import NominalBatch
    rows, curves = _progress_cases()
    # one call per distinct parameter set (the thresholds are per call, like the reference's attributes)
    keys = sorted({tuple(r[4:8]) for r in rows})
    for key in keys:
        sel = [r for r in rows if tuple(r[4:8]) == key]
        nb = NominalBatch(10, 0.1, "diff")
        nb.set_curves([curves[int(r[0])] for r in sel], 0.4, [int(r[1]) for r in sel])
        pidx, md, arr = nb.progress(np.array([[r[2], r[3], 0.0] for r in sel]), key[0], int(key[1]), key[2], int(key[3]))
        pidx, md, arr = pidx.cpu().numpy(), md.cpu().numpy(), arr.cpu().numpy()
        for j, r in enumerate(sel):
            assert pidx[j] == int(r[8]) and bool(arr[j]) == bool(r[10])
            assert abs(md[j] - np.float32(r[9])) <= 1e-6


def test_path_bookkeeping_matches_reference():
    """gear split, average interval (FleetPlanner.set_paths) and heading repair (planner._consistent_angles) against
    the reference's own InitialPath.set_initial_path / _ensure_consistent_angles (frontend_pathbook.npz)"""
    from neupan_amd.fleet import FleetPlanner
    from neupan_amd.planner import _consistent_angles
    g = np.load(os.path.join(HERE, "golden", "frontend_pathbook.npz"))
    for name in g["names"]:
        name = str(name)
        path = [row.reshape(4, 1).copy() for row in g[name + "/path"]]
        curves = FleetPlanner._split_by_gear(path)
        assert [len(c) for c in curves] == list(g[name + "/curve_len"]), name
        assert np.array_equal(np.vstack(curves), g[name + "/curves"]), name
        assert FleetPlanner._average_interval(path) == float(g[name + "/interval"]), name
        _consistent_angles(path)
        assert np.array_equal(np.hstack(path).T, g[name + "/consistent"]), name
