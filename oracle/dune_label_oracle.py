"""CPU restatement of the DUNE training-label problem.

TEST INFRASTRUCTURE ONLY (tests/, tests/tools/): never imported by the product path.

Reference: neupan/blocks/dune_train.py:82-99 (problem), :137-140 (solve with ECOS), :101-107 (labels
are cast to float32 tensors):

        max_mu  mu^T (G p - h)    s.t.  || G^T mu ||_2 <= 1,  mu >= 0            (one SOCP per point)

PARITY UNPINNED against the reference's solver: cvxpy / ECOS (unpinned in pyproject.toml) are not
installable here, so no label the reference itself produced could be generated.  What anchors this
restatement instead:
  1. the problem has a closed-form solution (below) and every answer carries a primal-dual
     optimality certificate (`certificate`): a feasible primal point q of min ||p - x|| s.t. G x <= h
     and a feasible dual mu with equal objective values -- by weak duality both are optimal;
  2. the reference's shipped checkpoints were trained on ECOS labels and log their final validation
     losses (example/model/*/results.txt, last block): the closed-form labels reproduce those
     losses against the shipped networks (tests/test_dune_labels.py).

Closed form.  The problem is the Lagrange dual of the distance from p to the polygon {x: G x <= h}
(rows of G are outward edge normals, un-normalised, counter-clockwise: util/__init__.py:161-206).
Inside: value 0 at mu = 0.  Outside, with q the nearest polygon point and n = (p - q)/|p - q|:
G^T mu = n with mu supported on the edges active at q -- one edge (mu_e = 1/|G_e|) when q is interior
to an edge, the two edges meeting at q when q is a vertex (2x2 solve).  The support is unique
because adjacent edge normals are linearly independent.
"""
import numpy as np


def polygon_vertices(G, h):
    """vertex e = intersection of edges e-1 and e (rows are consecutive CCW edges)."""
    G = np.asarray(G, dtype=np.float64); h = np.asarray(h, dtype=np.float64).reshape(-1)
    E = G.shape[0]
    V = np.zeros((E, 2))
    for e in range(E):
        A = np.array([G[(e - 1) % E], G[e]])
        V[e] = np.linalg.solve(A, np.array([h[(e - 1) % E], h[e]]))
    return V


def label_point(G, h, V, p):
    """Returns (mu [E], dist, q) in float64."""
    E = G.shape[0]
    s = G @ p - h
    mu = np.zeros(E)
    if s.max() <= 0:
        return mu, 0.0, p.copy()
    best = (np.inf, 0, 0.0, None)
    for e in range(E):                        # edge e runs from V[e] to V[e+1]
        a, b = V[e], V[(e + 1) % E]
        d = b - a
        t = min(1.0, max(0.0, float((p - a) @ d) / float(d @ d)))
        q = a + t * d
        dd = float((p - q) @ (p - q))
        if dd < best[0]:
            best = (dd, e, t, q)
    dd, e, t, q = best
    dist = np.sqrt(dd)
    if 0.0 < t < 1.0:
        mu[e] = 1.0 / np.linalg.norm(G[e])
        return mu, float(s[e] * mu[e]), q
    i, j = ((e - 1) % E, e) if t == 0.0 else (e, (e + 1) % E)       # the two edges meeting at the vertex
    n = (p - q) / dist
    m = np.linalg.solve(np.array([G[i], G[j]]).T, n)
    mu[i], mu[j] = max(m[0], 0.0), max(m[1], 0.0)
    return mu, dist, q


def labels(G, h, points):
    """points (n,2) float64 -> mu (n,E), dist (n,) float64 (the reference casts to float32)."""
    G = np.asarray(G, dtype=np.float64); h = np.asarray(h, dtype=np.float64).reshape(-1)
    V = polygon_vertices(G, h)
    P = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    mu = np.zeros((P.shape[0], G.shape[0])); dist = np.zeros(P.shape[0])
    for k in range(P.shape[0]):
        mu[k], dist[k], _ = label_point(G, h, V, P[k])
    return mu, dist


def certificate(G, h, p, mu, dist):
    """Primal-dual optimality residuals for one point: dict of non-negative numbers, all ~1e-12
    for an optimal (mu, dist)."""
    G = np.asarray(G, dtype=np.float64); h = np.asarray(h, dtype=np.float64).reshape(-1)
    V = polygon_vertices(G, h)
    _, _, q = label_point(G, h, V, np.asarray(p, dtype=np.float64))
    primal = float(np.linalg.norm(p - q))
    return dict(primal_feas=float(max(0.0, (G @ q - h).max())),
                dual_cone=float(max(0.0, np.linalg.norm(G.T @ mu) - 1.0)),
                dual_sign=float(max(0.0, -mu.min())),
                gap=abs(primal - float(mu @ (G @ p - h))),
                value=abs(primal - dist))
