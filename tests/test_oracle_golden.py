"""CPU tests: the oracle restatement vs vectors produced by the UNMODIFIED reference code
(tests/golden/make_golden.py) and vs the reference's own fixtures (results.txt G/h)."""
import numpy as np
import pytest

from helpers import CONFIGS, golden, make_oracle, robot_numbers
from oracle import pan_oracle as po
from oracle.condensed_ipm import solve_condensed
from oracle.nrmp_qp import NrmpProblem, kkt_certificate, solve_nrmp_qp

POLY = dict(kinematics="diff", vertices=[[-0.8, -1.0], [-1.8, 1.0], [1.8, 1.0], [0.8, -1.0]],
            max_speed=[8, 3], max_acce=[8, 3])
OMNI = dict(kinematics="omni", length=1.6, width=2.0, max_speed=[8, 6.28], max_acce=[3, 3])


def test_geometry_matches_reference_results_txt():
    # reference fixtures: example/model/diff_robot_default/results.txt:1-8,
    # example/model/acker_robot_default/results.txt:1-8, example/model/polygon_robot/results.txt:1-8
    expect = {
        "diff": ([[0, -1.6], [2, 0], [0, 1.6], [-2, 0]], [1.6, 1.6, 1.6, 1.6], CONFIGS["diff_1k_T10_K10"].robot),
        "acker": ([[0, -4.6], [1.6, 0], [0, 4.6], [-1.6, 0]], [3.68, 6.08, 3.68, 1.28], CONFIGS["acker_2k_T20_K15"].robot),
        "polygon": ([[0, -1.6], [2, -1], [0, 3.6], [-2, -1]], [1.6, 2.6, 3.6, 2.6], POLY),
    }
    geo = golden("geometry")
    for name, (G, h, kw) in expect.items():
        Go, ho, sp, ac, _ = robot_numbers(kw, 0.1)
        np.testing.assert_allclose(Go, np.array(G, float), atol=1e-12)
        np.testing.assert_allclose(ho.reshape(-1), np.array(h, float), atol=1e-12)
        np.testing.assert_array_equal(Go, geo[name + "_G"])          # vs the reference's robot class
        np.testing.assert_array_equal(ho, geo[name + "_h"])
        np.testing.assert_allclose(sp, geo[name + "_speed_bound"].reshape(-1))
        np.testing.assert_allclose(ac, geo[name + "_acce_bound"].reshape(-1))


def test_downsample_decimation_indices():
    m = np.arange(2000, dtype=np.float32).reshape(2, 1000)
    out = po.downsample_decimation(m, 100)
    assert out.shape == (2, 100) and out[0, 0] == 0 and out[0, 1] == 10 and out[0, -1] == 999
    assert out[0, 98] == 988 and out[0, 97] == 978       # truncation, SURVEY 8a-2
    assert po.downsample_decimation(m, 1000) is m


STAGES = [("diff_n1000", "diff_1k_T10_K10", None, None), ("dyna_n300", "dyna_4k_T10_K10", None, None),
          ("acker_n200", "acker_2k_T20_K15", None, None), ("diff_n7", "diff_1k_T10_K10", None, None),
          ("diff_n1", "diff_1k_T10_K10", None, None), ("decimate_1000_to_100", "diff_1k_T10_K10", None, None),
          ("polygon_n150", "diff_1k_T10_K10", POLY, "polygon_robot"), ("omni_n64", "diff_1k_T10_K10", OMNI, None)]


@pytest.mark.parametrize("case,cfgname,robot_kw,ck", STAGES)
def test_stage_functions_match_reference(case, cfgname, robot_kw, ck):
    g = golden("stage_" + case)
    cfg = CONFIGS[cfgname]
    rk = dict(cfg.robot if robot_kw is None else robot_kw)
    from helpers import ckpt_path
    w = po.ObsPointNetWeights.from_checkpoint(ckpt_path(ck or cfg.checkpoint))
    G, h = g["G"], g["h"]
    T = g["flow"].shape[0] - 1
    vel = g["velocities"] if bool(g["has_vel"]) else None
    flow, Rl, pl = po.generate_point_flow(g["nom_s"], g["points"], vel, T, cfg.dt, int(g["dune_max_num"]))
    np.testing.assert_allclose(np.stack(flow), g["flow"], atol=2e-6)
    np.testing.assert_allclose(np.stack(Rl), g["R"], atol=1e-7)
    np.testing.assert_array_equal(np.stack(pl), g["pts_t"])
    mu_l, lam_l, pt_l, mind = po.dune_forward(w, G, h, flow, Rl, pl)
    M = int(g["nrmp_max_num"])
    # the first M sorted columns are what the planner consumes (nrmp.py:254-255)
    k = min(M, g["mu"].shape[2])
    np.testing.assert_allclose(np.stack(mu_l)[:, :, :k], g["mu"][:, :, :k], atol=2e-5)
    np.testing.assert_allclose(np.stack(lam_l)[:, :, :k], g["lam"][:, :, :k], atol=5e-5)
    np.testing.assert_allclose(np.stack(pt_l)[:, :, :k], g["sorted_pts"][:, :, :k], atol=1e-6)
    assert abs(float(mind) - float(g["min_distance"])) < 2e-5
    fa, fb = po.generate_coefficient_parameter_value(mu_l, lam_l, pt_l, h, T, M)
    np.testing.assert_allclose(fa, g["fa"], atol=5e-5)
    np.testing.assert_allclose(fb.reshape(T, M, 1), g["fb"], atol=2e-4)
    A, B, C = po.generate_state_parameter_value(g["nom_s"], g["nom_u"], T, cfg.dt, rk["kinematics"], rk.get("wheelbase"))
    np.testing.assert_array_equal(A, g["A"])            # bit-exact: same fp32 rounding sequence
    np.testing.assert_array_equal(B, g["B"])
    np.testing.assert_array_equal(C.reshape(T, 3, 1), g["C"])


def test_obs_point_net_matches_torch_module():
    import torch
    from helpers import ckpt_path
    w = po.ObsPointNetWeights.from_checkpoint(ckpt_path("diff_robot_default"))
    net = torch.nn.Sequential(
        torch.nn.Linear(2, 32), torch.nn.LayerNorm(32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
        torch.nn.Linear(32, 32), torch.nn.LayerNorm(32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
        torch.nn.Linear(32, 32), torch.nn.LayerNorm(32), torch.nn.Tanh(), torch.nn.Linear(32, 4), torch.nn.ReLU())
    sd = torch.load(ckpt_path("diff_robot_default"), map_location="cpu")
    assert len(sd) == 18 and sum(v.numel() for v in sd.values()) == 4644
    net.load_state_dict({k.replace("MLP.", ""): v for k, v in sd.items()})
    x = (np.random.default_rng(0).uniform(-25, 25, (4096, 2))).astype(np.float32)
    with torch.no_grad():
        ref = net(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(po.obs_point_net(w, x), ref, atol=2e-5)


PANS = [("diff_n1000_k3", "diff_1k_T10_K10", dict(iter_num=3)),
        ("diff_n200_k10", "diff_1k_T10_K10", dict(iter_num=10, dune_max_num=200)),
        ("dyna_n300_k4", "dyna_4k_T10_K10", dict(iter_num=4, dune_max_num=300)),
        ("acker_n200_k4", "acker_2k_T20_K15", dict(iter_num=4, dune_max_num=200)),
        ("diff_n7_k3", "diff_1k_T10_K10", dict(iter_num=3)),
        ("omni_n64_k3", "diff_1k_T10_K10", dict(iter_num=3, robot_kw=OMNI)),
        ("polygon_n150_k3", "diff_1k_T10_K10", dict(iter_num=3, robot_kw=POLY, checkpoint="polygon_robot")),
        ("nopoints_k3", "diff_1k_T10_K10", dict(iter_num=3)),
        ("noobs_m0_k3", "diff_1k_T10_K10", dict(iter_num=3, nrmp_max_num=0)),
        ("default_thr_3calls", "diff_1k_T10_K10", dict(iter_num=6, iter_threshold=0.1, dune_max_num=100)),
        ("qs_vector_k2", "diff_1k_T10_K10", dict(iter_num=2, adjust=dict(q_s=[1.0, 0.8, 0.3]))),
        # tests/golden/make_golden_more.py
        ("decimate_n500_k3", "diff_1k_T10_K10", dict(iter_num=3, dune_max_num=100)),
        ("omni_dyna_n80_k3", "dyna_4k_T10_K10", dict(iter_num=3, dune_max_num=80, robot_kw=OMNI)),
        ("acker_reverse_n150_k4", "acker_2k_T20_K15", dict(iter_num=4, dune_max_num=150)),
        ("polygon_dyna_n100_k3", "dyna_4k_T10_K10", dict(iter_num=3, dune_max_num=100, robot_kw=POLY, checkpoint="polygon_robot")),
        # tests/golden/make_golden_full.py: BASELINE.json's configurations at their own sizes, two ensemble-well-posed scenes each
        ("diff_n1000_k10_s0", "diff_1k_T10_K10", dict(iter_num=10)), ("diff_n1000_k10_s1", "diff_1k_T10_K10", dict(iter_num=10)),
        ("acker_n2000_T20_k15_s0", "acker_2k_T20_K15", dict(iter_num=15)), ("acker_n2000_T20_k15_s1", "acker_2k_T20_K15", dict(iter_num=15)),
        ("dyna_n4000_k10_s0", "dyna_4k_T10_K10", dict(iter_num=10)), ("dyna_n4000_k10_s1", "dyna_4k_T10_K10", dict(iter_num=10)),
        ("polygon_n5000_k10_s0", "polygon_5k_T10_K10", dict(iter_num=10)), ("polygon_n5000_k10_s1", "polygon_5k_T10_K10", dict(iter_num=10))]


@pytest.mark.parametrize("case,cfgname,over", PANS)
def test_pan_forward_matches_reference_control_flow(case, cfgname, over):
    """The reference's PAN.forward (with the oracle QP substituted for CvxpyLayer) vs the
    pure-oracle loop: checks the loop, the parameter plumbing, the warm start, the stop
    criterion and its cross-call state.  Tolerance 2e-4: both sides run the same QP solver
    but the MLP runs through torch-CPU on one side and numpy on the other."""
    from helpers import ckpt_path
    g = golden("pan_" + case)
    over = dict(over)
    ck = over.pop("checkpoint", None)
    o = make_oracle(CONFIGS[cfgname], robot_kw=over.pop("robot_kw", None),
                    checkpoint=ckpt_path(ck) if ck else None, **over)
    for c in range(int(g["calls"])):
        pts = g[f"c{c}_points"] if f"c{c}_points" in g.files and case != "nopoints_k3" else None
        vel = g[f"c{c}_velocities"] if bool(g[f"c{c}_has_vel"]) else None
        s, u, d = o.forward(g[f"c{c}_nom_s"], g[f"c{c}_nom_u"], g[f"c{c}_ref_s"], g[f"c{c}_ref_us"], pts, vel)
        assert o.iters_run == int(g[f"c{c}_iters"])
        np.testing.assert_allclose(u, g[f"c{c}_opt_u"], atol=2e-4)
        np.testing.assert_allclose(s, g[f"c{c}_opt_s"], atol=2e-4)
        if d is None:
            assert g[f"c{c}_opt_d"].size == 0
        else:
            np.testing.assert_allclose(d, g[f"c{c}_opt_d"], atol=2e-4)
        if pts is not None and not o.no_obs:
            assert abs(float(o.min_distance) - float(g[f"c{c}_min_distance"])) < 2e-5


def _qp_problem(g, i):
    p = f"q{i}_"
    sc = g[p + "scalars"]
    fa = g[p + "fa"] if g[p + "fa"].size else None
    fb = g[p + "fb"] if g[p + "fb"].size else None
    return NrmpProblem(g[p + "nom_s"], g[p + "qref_s"], g[p + "puref"], g[p + "A"], g[p + "B"], g[p + "C"], fa, fb,
                       g[p + "q_s"], sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], g[p + "speed_bound"],
                       g[p + "acce_bound"], str(g[p + "kin"]))


def test_qp_oracle_vs_highs_and_certificate():
    """The fp64 QP oracle against (a) HiGHS' solution of the same problem, computed in the
    build container and stored in qp_cases.npz, (b) the independent KKT certificate,
    (c) the condensed-formulation prototype of the GPU algorithm."""
    g = golden("qp_cases")
    n = int(g["count"])
    assert n >= 20
    for i in range(n):
        pb = _qp_problem(g, i)
        s, u, d = solve_nrmp_qp(pb)
        np.testing.assert_allclose(u, g[f"q{i}_u"], atol=1e-9)           # deterministic re-solve
        # HiGHS terminates at ~1e-7 objective accuracy; along the flat (high-frequency
        # steering) directions of this QP that is a few 1e-6..1e-5 in u.
        np.testing.assert_allclose(u, g[f"q{i}_u_highs"], atol=5e-5)
        assert float(g[f"q{i}_obj_oracle"]) <= float(g[f"q{i}_obj_highs"]) + 1e-9
        cert = kkt_certificate(pb, s, u, d)
        assert cert["dyn"] < 1e-10 and cert["feas"] < 1e-9 and cert["stat"] < 1e-6 and cert["comp"] < 1e-8, cert
        s2, u2, d2, info = solve_condensed(pb)
        np.testing.assert_allclose(u2, u, atol=2e-6)
        np.testing.assert_allclose(s2, s, atol=2e-6)


@pytest.mark.parametrize("cfgname,scene,kw", [("diff_1k_T10_K10", 5, {}), ("diff_1k_T10_K10", 63, {}),
                                               ("acker_2k_T20_K15", 16, dict(iter_num=4)), ("dyna_4k_T10_K10", 3, dict(iter_num=4))])
def test_qp_oracle_vs_highs_on_benchmark_problems(cfgname, scene, kw):
    """The oracle's QP solve against HiGHS on EVERY QP of a benchmark scene's PAN loop, solved live (the committed
    qp_cases.npz holds 33 small problems; profiles/r02_qp_highs.json the sweep of tests/tools/qp_highs_sweep.py over
    whole batches: 2560 + 360 + 240 + 64 QPs): HiGHS optimal, the oracle's objective never worse than HiGHS' by more
    than 1e-9 relative, controls within HiGHS' own accuracy along the flat steering directions."""
    from neupan_amd.scenes import make_scene
    from oracle.nrmp_qp import kkt_certificate
    from qp_highs import compare_with_highs
    cfg = CONFIGS[cfgname]
    sc = make_scene(cfg, scene)
    orc = make_oracle(cfg, **kw)
    rows = []
    orig = orc.nrmp

    def hook(*a):
        r = orig(*a)
        pb = orc.last_problem
        s, u, d = solve_nrmp_qp(pb)
        c = compare_with_highs(pb, s, u, d)
        c["cert"] = kkt_certificate(pb, s, u, d)
        rows.append(c)
        return r
    orc.nrmp = hook
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
    assert len(rows) == orc.iter_num
    for c in rows:
        assert c["status"] == "Optimal", c
        assert c["obj_diff"] <= 1e-9 * max(1.0, abs(c["obj"])), c
        # HiGHS stops at ~1e-7 .. 1e-6 objective accuracy; along the flat steering directions that is 1e-5 in u for the
        # diff robot and up to 1e-2 for the car-like one (where the oracle's objective is BETTER by 1e-6): the objective is
        # the sharp comparison, the controls a sanity bound
        assert c["du"] <= (5e-2 if cfgname.startswith("acker") else 2e-4), c
        assert c["cert"]["dyn"] < 1e-10 and c["cert"]["feas"] < 1e-9 and c["cert"]["comp"] < 1e-8, c


def test_condensed_solver_warm_start_rules():
    """oracle/condensed_ipm.py carries the kernel's warm start (nrmp_qp.hip): floor QP_WARM_DELTA = 0.003, drop rules at
    iteration 0 / 6, cold retry.  (1) a solve started from its own solution is accepted, takes fewer iterations and
    lands on the same point; (2) a start from an unrelated problem's solution is dropped or converges -- either way the
    point is the cold solve's."""
    g = golden("qp_cases")
    n = int(g["count"])
    fewer = 0
    for i in range(min(n, 12)):
        pb = _qp_problem(g, i)
        s0, u0, d0, info0 = solve_condensed(pb)
        assert info0["warm_code"] == 0
        s1, u1, d1, info1 = solve_condensed(pb, warm=info0["warm"])
        np.testing.assert_allclose(u1, u0, atol=2e-6)
        assert info1["warm_code"] in (1, 2, 3, 4)
        fewer += info1["warm_code"] == 1 and info1["iters_total"] < info0["iters_total"]
        j = (i + 5) % n
        pj = _qp_problem(g, j)
        if pj.T == pb.T and pj.no_obs == pb.no_obs and getattr(pj, "M", None) == getattr(pb, "M", None):
            sj = solve_condensed(pj)[3]["warm"]
            if all(a.shape == b.shape for a, b in zip(sj, info0["warm"])):
                s2, u2, d2, info2 = solve_condensed(pb, warm=sj)
                np.testing.assert_allclose(u2, u0, atol=2e-6)
    assert fewer >= 6, fewer


def test_condensed_solver_repeats_a_jammed_cold_solve():
    """Scene 66 of the acker workload: from the centred cold start (multipliers 3 / slack) the interior-point method jams --
    a step lands on the boundary too early, three non-improving iterations end the solve at 3e-6 -- while unit multipliers
    go through.  oracle/condensed_ipm.py carries the kernel's rule (nrmp_qp.hip, QP_RETRY_MERIT): a cold solve that ends
    above 1e-9 is repeated once from unit multipliers (warm_code 5).  The solution is the uncondensed oracle's.  (Round 6's
    centrality safeguard removes this jam at its root; the retry stays as the backstop and is exercised here under round 5's rules.)"""
    from neupan_amd.scenes import make_scene
    from oracle import condensed_ipm as ci
    cfg = CONFIGS["acker_2k_T20_K15"]
    sc = make_scene(cfg, 66)
    orc = make_oracle(cfg, iter_num=5)
    pbs = []
    orig = orc.nrmp

    def hook(*a):
        r = orig(*a)
        pbs.append(orc.last_problem)
        return r
    orc.nrmp = hook
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
    pb = pbs[-1]
    # round 6: with the centrality safeguard of blocked steps (CENTRAL_*) the centred start goes through by itself ...
    s, u, d, info = solve_condensed(pb)
    assert info["warm_code"] == 0 and info["merit"] <= 1e-12, info
    old = (ci.RETRY_MERIT, ci.STALL_FAR, ci.CENTRAL_GAMMA)
    try:
        # ... under round 5's rules it jams (three non-improving iterations end it at 3e-6) and the retry from unit multipliers saves it
        ci.STALL_FAR, ci.CENTRAL_GAMMA = 3, 0.0
        s5, u5, d5, info5 = solve_condensed(pb)
        assert info5["warm_code"] == 5 and info5["merit"] <= 1e-12, info5
        ci.RETRY_MERIT = float("inf")                     # without the retry rule: the jam
        s1, u1, d1, info1 = solve_condensed(pb)
    finally:
        ci.RETRY_MERIT, ci.STALL_FAR, ci.CENTRAL_GAMMA = old
    assert info1["warm_code"] == 0 and info1["merit"] > 1e-9, info1
    np.testing.assert_allclose(u5, u, atol=1e-7)
    s0, u0, d0 = solve_nrmp_qp(pb)
    np.testing.assert_allclose(u, u0, atol=5e-5)          # (acker QPs are flat in the steering direction: DESIGN.md section 5)


def test_one_step_explanations_flag_what_they_cannot_explain():
    """tests/parity_tools.one_step_consistency(explain=True), the machinery behind verdict D: fed with the ORACLE's own trace it
    finds nothing above the tolerance; fed with controls that are off by 1e-3 for no reason it reports every such step as
    UNEXPLAINED (no rank-M tie: the selection is the oracle's own; no sensitivity: the one-step ensemble under +-1 ulp does
    not move by a third of 1e-3 on these well-posed scenes)."""
    import dataclasses
    from helpers import CONFIGS, make_oracle
    from neupan_amd import scenes as sc_mod
    from parity_tools import one_step_consistency, one_step_report
    name = "corridor_diff_small_k3"
    cfg = dataclasses.replace(CONFIGS["corridor_diff_small"], name=name, iter_num=3, n_points=120)
    sc_mod.CONFIGS[name] = cfg
    try:
        S = 2
        tr_s, tr_u, tr_p = [], [], []
        for b in range(S):
            sc = sc_mod.make_scene(cfg, b)
            orc = make_oracle(cfg)
            ps = []
            orig = orc.nrmp

            def hook(*a, _ps=ps, _orig=orig):
                _ps.append(np.stack([p[:, :cfg.nrmp_max_num].T for p in a[6]]))         # the first M sorted points of every slice
                return _orig(*a)
            orc.nrmp = hook
            orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
            tr_s.append(np.stack([t[0] for t in orc.trace])); tr_u.append(np.stack([t[1] for t in orc.trace])); tr_p.append(np.stack(ps))
        tr_s, tr_u, tr_p = np.stack(tr_s), np.stack(tr_u), np.stack(tr_p)
        dev, why = one_step_consistency(name, range(S), tr_s, tr_u, 1, explain=True, trace_pts=tr_p)
        assert dev.max() <= 1e-6 and why == []
        bad = tr_u.copy()
        bad[1, 2] += np.float32(1e-3 / np.sqrt(bad[1, 2].size))          # one step, control L2 = 1e-3
        dev, why = one_step_consistency(name, range(S), tr_s, bad, 1, explain=True, trace_pts=tr_p)
        rep = one_step_report(dev, why=why)
        assert len(why) == 1 and why[0]["scene"] == 1 and why[0]["iteration"] == 3 and why[0]["explained"] is None
        assert why[0]["slices_with_other_set"] == 0 and rep["unexplained"] == 1
    finally:
        del sc_mod.CONFIGS[name]
