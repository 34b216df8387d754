"""The rule of the selection's second-stage filter (csrc/select_geo_body.inc, TABF), restated in numpy on the
reference's own checkpoint and checked for what it must guarantee -- CPU only, nothing of the library runs here:

    key(p)  = g(p) + bilinear f_table(p),   f = network distance - geometric distance at the table's nodes (fp16 cells)
    m[b]    = safety x the largest |key - exact| over the calibration samples whose KEY falls into band b (+- one band)
    U'      = max over the M smallest keys of a list of  key + m[band(key)]
    survive = key - m[band(key)] <= U'

Whatever the list, every one of its true M nearest points (by EXACT network distance) must survive -- provided the residual of
every point is within its margin, which is what the kernel audits at run time; and the rule must be worth something (few
survivors beyond M).  The margins are calibrated here the way npa_create does it (a grid that is not aligned with the cells)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
f32 = np.float32
NB = 88


def band(x):                                    # pan_common.h npa_geo_band: 8 bands per octave of x + 0.25
    v = (np.asarray(x, dtype=np.float32) + f32(0.25)).view(np.uint32)
    return np.clip((v >> 20).astype(np.int64) - (0x3E800000 >> 20), 0, NB - 1)


def test_table_key_rule_keeps_every_true_member_and_prunes():
    from helpers import CONFIGS, make_oracle
    from oracle.pan_oracle import obs_point_net
    cfg = CONFIGS["diff_1k_T10_K10"]
    orc = make_oracle(cfg)
    G, h = np.asarray(orc.G, dtype=np.float64), np.asarray(orc.h, dtype=np.float64).ravel()
    w = orc.w if hasattr(orc, "w") else orc.weights
    # the robot is an axis-aligned box: centre / half extents from its half planes (rows +-x, +-y)
    V = []
    for e in range(4):
        p = (e - 1) % 4
        a, b, c, d = G[p][0], G[p][1], G[e][0], G[e][1]
        det = a * d - b * c
        V.append(((h[p] * d - b * h[e]) / det, (a * h[e] - h[p] * c) / det))
    V = np.array(V)
    cx, cy = 0.5 * (V[:, 0].min() + V[:, 0].max()), 0.5 * (V[:, 1].min() + V[:, 1].max())
    hx, hy = 0.5 * (V[:, 0].max() - V[:, 0].min()), 0.5 * (V[:, 1].max() - V[:, 1].min())

    def geo(x, y):
        dx, dy = np.maximum(np.abs(x - cx) - hx, 0), np.maximum(np.abs(y - cy) - hy, 0)
        return np.sqrt(dx * dx + dy * dy)

    def net(x, y):
        P = np.stack([x.ravel(), y.ravel()], 1).astype(np.float32)
        mu = obs_point_net(w, P).astype(np.float64)
        return np.einsum("ne,ne->n", mu, P.astype(np.float64) @ G.T - h).reshape(x.shape)

    # level 0 of the table at the kernel's cell size (7.8 mm) over a window of it that holds the robot: +-2 m x [-1.5, 0] m
    half, N = 2.0, 512
    rows = np.arange(0, 193)                                      # y from -2 m to -0.5 m
    xs = np.linspace(-half, half, N + 1)
    X, Y = np.meshgrid(xs, xs[rows], indexing="xy")
    F = (net(X, Y) - geo(X, Y)).astype(np.float16).astype(np.float64)

    def key(x, y):
        fx, fy = (x + half) * (N / (2 * half)), (y + half) * (N / (2 * half))
        ix, iy = np.clip(fx.astype(int), 0, N - 1), np.clip(fy.astype(int), 0, rows[-1] - 1)
        tx, ty = fx - ix, fy - iy
        r0 = F[iy, ix] + tx * (F[iy, ix + 1] - F[iy, ix])
        r1 = F[iy + 1, ix] + tx * (F[iy + 1, ix + 1] - F[iy + 1, ix])
        return geo(x, y) + r0 + ty * (r1 - r0)
    # calibration: a grid that drifts through the cells, margins per band of the KEY over the band and its neighbours, x 2
    gs = -half + 2 * half * (np.arange(1601) + 0.37) / 1601
    gy = gs[(gs > -1.98) & (gs < -0.52)]
    SX, SY = np.meshgrid(gs, gy, indexing="xy")
    de, kc = net(SX, SY), key(SX, SY)
    res, kb = np.abs(kc - de), band(kc)
    raw = np.zeros(NB)
    np.maximum.at(raw, kb.ravel(), res.ravel())
    seen = np.bincount(kb.ravel(), minlength=NB) > 0
    m = np.full(NB, np.inf)
    for b in range(NB):
        q = [r for r in (b - 1, b, b + 1) if 0 <= r < NB and seen[r]]
        if q:
            m[b] = max(2.0 * max(raw[r] for r in q), 1e-4)
    assert raw[seen].max() < 0.02 and np.median(res) < 1e-4           # millimetres at worst, far below the 75 mm of g alone
    # random lists inside the window, M = 10
    rng = np.random.default_rng(5)
    M, kept, viol = 10, [], 0
    for trial in range(300):
        n = int(rng.integers(40, 400))
        x, y = rng.uniform(-1.9, 1.9, n), rng.uniform(-1.95, -0.55, n)
        if trial % 3 == 0:                                            # a wall at constant distance: many near-ties
            y = np.full(n, -1.2) + rng.normal(0, 1e-3, n)
        ex, k = net(x, y), key(x, y)
        mk = m[band(k)]
        ok = np.abs(k - ex) <= mk                                     # (what the kernel's audit checks on every survivor)
        viol += int((~ok).sum())
        order = np.argsort(k, kind="stable")[:M]
        U = np.max(k[order] + mk[order])
        surv = k - mk <= U
        true_members = np.argsort(ex, kind="stable")[:M]
        assert surv[true_members].all(), (trial, n)
        kept.append(int(surv.sum()))
    assert viol == 0
    assert np.median(kept) <= 3 * M and max(kept) < 400, (np.median(kept), max(kept))


def test_shortened_search_for_the_nominating_bound_is_an_upper_bound():
    """Stage 1 nominates through the msel-th smallest of the 64 lane minima (select_geo_body.inc: a bit-by-bit search on ballots).
    Round 6 stops the search twelve bits short and fills them with ones.  Restated in numpy on the bit patterns of float keys: the
    result is >= the exact msel-th smallest (so every lane the exact bound admits still nominates: the candidates stay a superset),
    at most 2^12 - 1 key units above it (5e-4 of the key), and 0xFFFFFFFF exactly when fewer than msel lanes hold a point."""
    rng = np.random.default_rng(3)

    def search(lmin, msel, stop):
        bound = 0
        for bit in range(31, stop - 1, -1):
            trial = bound | ((1 << bit) - 1)
            if int((lmin <= trial).sum()) < msel:
                bound |= 1 << bit
        return bound | ((1 << stop) - 1)
    for trial_no in range(400):
        msel = int(rng.integers(1, 11))
        nvalid = int(rng.integers(0, 65))
        g = np.abs(rng.normal(0.0, rng.choice([0.01, 0.5, 3.0, 40.0]), 64)).astype(np.float32)
        if trial_no % 7 == 0:
            g[: nvalid // 2] = g[0]                             # ties
        lmin = g.view(np.uint32).astype(np.uint64)
        lmin[nvalid:] = 0xFFFFFFFF                              # lanes without a point
        exact = search(lmin, msel, 0)
        assert exact == int(np.sort(lmin)[msel - 1])            # the full search IS the msel-th smallest, with multiplicity
        short = search(lmin, msel, 12)
        assert short >= exact and short - exact <= 0xFFF and (short >> 12) == (exact >> 12)
        assert (short == 0xFFFFFFFF) == (nvalid < msel)
        if nvalid >= msel and exact > 0x00800000:               # (a normal float: the slack is below 2^-11 of the key)
            gs, ge = np.uint32(short).view(np.float32), np.uint32(exact).view(np.float32)
            assert 0.0 <= float(gs - ge) <= 5e-4 * float(ge) + 1e-30
