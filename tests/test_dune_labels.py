"""DUNE training labels: closed form vs (1) its optimality certificate,
(2) the losses the reference logged for its shipped, ECOS-trained checkpoints, (3) -m gpu: the
HIP kernel vs the oracle.  PARITY UNPINNED against ECOS itself (not installable here)."""
import numpy as np
import pytest

from helpers import CONFIGS, make_oracle
from oracle import dune_label_oracle as dl
from oracle import pan_oracle as po

POLY_G = None

# final "Validate Mu Loss" / "Validate Distance Loss" of the reference's training logs
# (example/model/<name>/results.txt, last block; uniform points in [-25,25]^2, MSE)
LOGGED = {"diff_1k_T10_K10": (7.71e-06, 6.86e-06), "acker_2k_T20_K15": (2.31e-06, 9.40e-06)}


def _robot(cfgname):
    orc = make_oracle(CONFIGS[cfgname])
    return orc, np.asarray(orc.G, np.float64), np.asarray(orc.h, np.float64).reshape(-1)


@pytest.mark.parametrize("cfgname", list(LOGGED))
def test_closed_form_is_optimal_and_matches_shipped_checkpoint(cfgname):
    orc, G, h = _robot(cfgname)
    rng = np.random.default_rng(0)
    P = rng.uniform(-25, 25, (6000, 2))
    P[:200] = rng.uniform(-3, 3, (200, 2))                   # inside / near the robot
    mu, dist = dl.labels(G, h, P)
    for k in range(0, 6000, 5):
        c = dl.certificate(G, h, P[k], mu[k], dist[k])
        assert max(c.values()) <= 1e-11, (k, c)
    # the reference's networks were fitted to ECOS labels: same mean-square error against ours
    net = po.obs_point_net(orc.w, P[200:].astype(np.float32))
    dnet = np.einsum("ne,ne->n", net, (G @ P[200:].T - h[:, None]).T)
    mse_mu, mse_d = np.mean((net - mu[200:]) ** 2), np.mean((dnet - dist[200:]) ** 2)
    assert 0.3 * LOGGED[cfgname][0] <= mse_mu <= 3 * LOGGED[cfgname][0]
    assert 0.3 * LOGGED[cfgname][1] <= mse_d <= 3 * LOGGED[cfgname][1]


def test_closed_form_octagon_and_trapezoid():
    from neupan_amd.robot import halfplanes_from_vertices
    rng = np.random.default_rng(1)
    ang = np.linspace(0, 2 * np.pi, 9)[:-1] + 0.2
    octa = np.stack([1.5 * np.cos(ang), 0.9 * np.sin(ang)])
    trap = np.array([[-0.8, -1.8, 1.8, 0.8], [-1.0, 1.0, 1.0, -1.0]])
    for verts in (octa, trap):
        G, h = halfplanes_from_vertices(verts)
        G = np.asarray(G, np.float64); h = np.asarray(h, np.float64).reshape(-1)
        P = rng.uniform(-6, 6, (1500, 2))
        mu, dist = dl.labels(G, h, P)
        assert (np.count_nonzero(mu, axis=1) <= 2).all()
        for k in range(0, 1500, 3):
            c = dl.certificate(G, h, P[k], mu[k], dist[k])
            assert max(c.values()) <= 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("cfgname", list(LOGGED))
def test_hip_labels_vs_oracle(cfgname):
    from neupan_amd.dune_labels import dune_labels
    orc, G, h = _robot(cfgname)
    rng = np.random.default_rng(2)
    P = rng.uniform(-25, 25, (100000, 2))
    P[:500] = rng.uniform(-3, 3, (500, 2))
    mu_g, d_g = (t.cpu().numpy() for t in dune_labels(G, h, P))
    idx = np.arange(0, 100000, 9)
    mu, dist = dl.labels(G, h, P[idx])
    # float32 labels of float64 results: 1 ulp (the reference casts the same way, dune_train.py:101-107)
    assert np.abs(mu_g[idx] - mu.astype(np.float32)).max() <= 1.2e-7
    assert np.abs(d_g[idx] - dist.astype(np.float32)).max() <= np.spacing(np.float32(40.0))


@pytest.mark.gpu
def test_hip_labels_octagon():
    from neupan_amd.dune_labels import dune_labels
    from neupan_amd.robot import halfplanes_from_vertices
    ang = np.linspace(0, 2 * np.pi, 9)[:-1] + 0.2
    G, h = halfplanes_from_vertices(np.stack([1.5 * np.cos(ang), 0.9 * np.sin(ang)]))
    G = np.asarray(G, np.float64); h = np.asarray(h, np.float64).reshape(-1)
    P = np.random.default_rng(3).uniform(-6, 6, (20000, 2))
    mu_g, d_g = (t.cpu().numpy() for t in dune_labels(G, h, P))
    mu, dist = dl.labels(G, h, P[::4])
    assert np.abs(mu_g[::4] - mu.astype(np.float32)).max() <= 2.4e-7
    assert np.abs(d_g[::4] - dist.astype(np.float32)).max() <= 1e-6
