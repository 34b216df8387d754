"""Gradient of the NRMP solution w.r.t. the adjust parameters.
PARITY UNPINNED against cvxpylayers (not installable here): the oracle's implicit gradient
(oracle/nrmp_backward.py) is pinned by central finite differences of the oracle's forward solve
(CPU test), the HIP kernel by the oracle gradient (-m gpu)."""
import numpy as np
import pytest

from helpers import CONFIGS, make_oracle
from neupan_amd.scenes import make_scene
from oracle import nrmp_backward as nb
from oracle import pan_oracle as po


def _problems(cfgname, scenes, iters=3):
    cfg = CONFIGS[cfgname]
    out = []
    for b in scenes:
        sc = make_scene(cfg, b)
        orc = make_oracle(cfg, iter_num=iters)
        keep = []
        orig = po.solve_nrmp_qp
        po.solve_nrmp_qp = lambda pb, *a, **k: (keep.append(pb), orig(pb, *a, **k))[1]
        try:
            orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], None)
        finally:
            po.solve_nrmp_qp = orig
        out.append((sc, orc, keep))
    return cfg, out


def test_oracle_gradient_vs_finite_differences():
    cfg, data = _problems("diff_1k_T10_K10", range(6))
    rng = np.random.default_rng(0)
    errs = []
    for sc, orc, pbs in data:
        pb = pbs[-1]
        T = pb.T
        gs, gu, gd = rng.standard_normal((3, T + 1)), rng.standard_normal((2, T)), rng.standard_normal((1, T))
        a, f = nb.backward_ipm(pb, gs, gu, gd), nb.backward_fd(pb, gs, gu, gd, 1e-5)
        errs.append(max(np.abs(np.atleast_1d(a[k]) - np.atleast_1d(f[k])).max() / max(1.0, np.abs(np.atleast_1d(f[k])).max())
                        for k in a))
    errs = np.sort(errs)
    # finite differences straddle active-set changes on some problems: the bulk must agree tightly
    assert errs[len(errs) // 2] <= 1e-4 and errs[-1] <= 5e-2, errs


@pytest.mark.gpu
def test_hip_gradient_vs_oracle():
    import torch
    from gpu_helpers import make_gpu_pan
    cfg, data = _problems("diff_1k_T10_K10", range(8), iters=2)
    pan = make_gpu_pan(cfg, iter_num=2)
    rng = np.random.default_rng(1)
    worst = []
    for sc, orc, pbs in data:
        # the oracle's last solve linearises around the result of its first iteration
        pb = pbs[-1]
        T = pb.T
        tr = orc.trace[-1] if hasattr(orc, "trace") and orc.trace else None
        nom_s, nom_u = pb.nom_s.astype(np.float32), None
        gs, gu, gd = (rng.standard_normal((3, T + 1)).astype(np.float32), rng.standard_normal((2, T)).astype(np.float32),
                      rng.standard_normal((1, T)).astype(np.float32))
        # run the stage API on the same nominal trajectory: iteration-0 output of the GPU = nominal of iteration 1
        first = make_gpu_pan(cfg, iter_num=1).forward_batch(sc["nom_s"][None], sc["nom_u"][None], sc["ref_s"][None],
                                                            sc["ref_us"][None], sc["points"][None])
        s1, u1 = first["opt_s"], first["opt_u"]
        stage = pan.dune_stage(s1, sc["points"][None])
        r = pan.nrmp_backward(s1, u1, sc["ref_s"][None], sc["ref_us"][None], stage, gs[None], gu[None], gd[None])
        g = r["grad"].cpu().numpy()[0]
        assert g[7] == 0
        ref = nb.backward_ipm(pb, gs.astype(np.float64), gu.astype(np.float64), gd.astype(np.float64))
        want = np.array([*ref["q_s"], ref["p_u"], ref["eta"], ref["d_max"], ref["d_min"]])
        worst.append(np.abs(g[:7] - want).max() / max(1.0, np.abs(want).max()))
        gn = r["grad_nom_s"].cpu().numpy()[0]
        assert np.all(gn[:, 0] == 0)
        worst.append(np.abs(gn - ref["nom_s"]).max() / max(1.0, np.abs(ref["nom_s"]).max()))
    worst = np.sort(worst)
    # the GPU nominal differs from the oracle's by fp32 rounding; weakly active rows amplify that in the gradient
    assert worst[len(worst) // 2] <= 1e-3 and worst[-1] <= 5e-2, worst


@pytest.mark.gpu
def test_autograd_fills_adjust_parameter_gradients():
    import torch
    from gpu_helpers import make_gpu_pan
    from neupan_amd.scenes import make_batch
    cfg = CONFIGS["diff_1k_T10_K10"]
    pan = make_gpu_pan(cfg, iter_num=3)
    batch = make_batch(cfg, 0, 4)
    f = pan.nrmp_layer
    for p in (f.p_u, f.eta, f.d_max):
        p.requires_grad_(True)
    s, u, d = pan.forward_batch_grad(batch["nom_s"], batch["nom_u"], batch["ref_s"], batch["ref_us"], batch["points"])
    loss = 50.0 - d.sum()                          # LON_corridor.py:13-14
    loss.backward()
    for p in (f.p_u, f.eta, f.d_max):
        assert p.grad is not None and np.isfinite(float(p.grad))
    # d rises with eta (the reward on d) => d loss/d eta < 0 wherever some d_t is strictly inside its box
    assert float(f.eta.grad) <= 0.0
    # same numbers as the plain forward
    out = make_gpu_pan(cfg, iter_num=3).forward_batch(batch["nom_s"], batch["nom_u"], batch["ref_s"], batch["ref_us"], batch["points"])
    assert np.array_equal(out["opt_u"].cpu().numpy(), u.detach().cpu().numpy())


def test_oracle_recurrent_gradient_vs_finite_differences():
    """the chain through the proximal centres of all K solves (what the reference's autograd graph carries,
    oracle/nrmp_backward.py docstring) against central differences over the same graph"""
    cfg, data = _problems("diff_1k_T10_K10", range(4), iters=3)
    rng = np.random.default_rng(5)
    errs, rec = [], []
    for sc, orc, pbs in data:
        assert len(pbs) == 3
        T = pbs[0].T
        gs, gu, gd = rng.standard_normal((3, T + 1)), rng.standard_normal((2, T)), rng.standard_normal((1, T))
        one = nb.backward_ipm(pbs[-1], gs, gu, gd)
        # single-solve sensitivity to the proximal centre
        f1 = nb.backward_fd(pbs[-1], gs, gu, gd, 1e-5)
        errs.append(np.abs(one["nom_s"] - f1["nom_s"]).max() / max(1.0, np.abs(f1["nom_s"]).max()))
        a, f = nb.pan_backward(pbs, gs, gu, gd), nb.pan_backward_fd(pbs, gs, gu, gd, 1e-5)
        errs.append(max(np.abs(np.atleast_1d(a[k]) - np.atleast_1d(f[k])).max() / max(1.0, np.abs(np.atleast_1d(f[k])).max())
                        for k in a))
        rec.append(max(np.abs(np.atleast_1d(a[k]) - np.atleast_1d(one[k])).max() for k in a))
    errs = np.sort(errs)
    assert errs[len(errs) // 2] <= 1e-4 and errs[-1] <= 5e-2, errs
    assert max(rec) > 1e-3            # the recurrent terms are not negligible: the test exercises them


@pytest.mark.gpu
def test_hip_recurrent_gradient_vs_oracle():
    """autograd through PAN.forward_batch_grad (all K solves chained through the proximal centre) against
    oracle.nrmp_backward.pan_backward on the oracle's own K problems of the same scenes"""
    import torch
    from gpu_helpers import make_gpu_pan
    K = 3
    cfg, data = _problems("diff_1k_T10_K10", range(8), iters=K)
    rng = np.random.default_rng(2)
    worst, rec = [], []
    for sc, orc, pbs in data:
        T = pbs[0].T
        gs, gu, gd = (rng.standard_normal((3, T + 1)).astype(np.float32), rng.standard_normal((2, T)).astype(np.float32),
                      rng.standard_normal((1, T)).astype(np.float32))
        grads = {}
        for recurrent in (True, False):
            pan = make_gpu_pan(cfg, iter_num=K)
            pan.recurrent = recurrent
            f = pan.nrmp_layer
            params = [f.q_s, f.p_u, f.eta, f.d_max, f.d_min]
            for p in params:
                p.requires_grad_(True)
            s, u, d = pan.forward_batch_grad(sc["nom_s"][None], sc["nom_u"][None], sc["ref_s"][None], sc["ref_us"][None],
                                             sc["points"][None])
            dev = s.device
            loss = (s[0] * torch.from_numpy(gs).to(dev)).sum() + (u[0] * torch.from_numpy(gu).to(dev)).sum() + \
                   (d[0] * torch.from_numpy(gd).to(dev)).sum()
            loss.backward()
            grads[recurrent] = np.array([float(p.grad.sum()) for p in params])
        ref = nb.pan_backward(pbs, gs.astype(np.float64), gu.astype(np.float64), gd.astype(np.float64))
        want = np.array([ref["q_s"].sum(), ref["p_u"], ref["eta"], ref["d_max"], ref["d_min"]])
        worst.append(np.abs(grads[True] - want).max() / max(1.0, np.abs(want).max()))
        rec.append(np.abs(grads[True] - grads[False]).max())
    worst = np.sort(worst)
    assert worst[len(worst) // 2] <= 1e-3 and worst[-1] <= 5e-2, worst
    assert max(rec) > 1e-3                       # the chained terms are there


@pytest.mark.gpu
def test_gradient_tests_on_the_generic_instantiation():
    """The backward solve runs on the register-resident QP instantiations (T = 10 / 20, M = 10) since round 3; the generic
    (LDS) instantiation still serves every other shape.  NPA_QP_GENERIC=1 is read once per process, so the two GPU gradient
    tests above are run once more in a child process with it set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NPA_QP_GENERIC="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_nrmp_backward.py"), "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider", "-k", "hip_gradient_vs_oracle or hip_recurrent_gradient_vs_oracle"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "2 passed" in r.stdout, r.stdout[-500:]
