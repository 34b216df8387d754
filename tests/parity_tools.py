"""Parity evidence for the PAN loop: the HIP path against the oracle, judged by an ENSEMBLE of oracle runs.

TEST INFRASTRUCTURE (imports oracle/): used by tests/, tests/tools/ and the cpu_baseline leg of bench.py only.

Why an ensemble.  The PAN loop (reference neupan/blocks/pan.py:109-147) is a fixed-point iteration: iteration k+1
linearises around iteration k's solution and re-selects the M nearest points.  On most scenes the map contracts and
two fp32 evaluations that differ in the last bit stay together (control L2 ~1e-6).  On some it does not, within K
iterations: a last-bit difference is amplified by a constant factor per iteration and the final controls of two
equally valid evaluations of the REFERENCE ALGORITHM differ by 1e-3 .. 1e-1.  There the question "does the HIP path
match the reference" only has an answer up to that spread.  The ensemble measures it per scene and per iteration:

  * `ulp` members: the oracle on inputs whose obstacle coordinates are each moved by +1 or -1 float32 ulp
    (independent random signs per member) -- the smallest change of the input a caller can make;
  * `perm` members: the oracle with the hidden units of the DUNE network permuted (weights, biases and LayerNorm
    vectors permuted consistently: the SAME function in exact arithmetic, a different fp32 summation order) --
    what a different BLAS / MFMA accumulation order does to the same checkpoint.

Per scene and iteration k: spread_k = max pairwise control L2 over {base, members}; hip_k = L2(HIP, base).
Verdict (`judge`):
  A. scenes with spread_K <= 1e-4 (the reference answer is defined to the north-star tolerance): hip_K <= 1e-4;
  B. the others: hip_K <= spread_K (the HIP path is inside the envelope of the reference's own evaluations);
  C. every scene, every iteration k before the ensemble first disagrees by > 1e-5: hip_k <= 1e-5.
"""
from __future__ import annotations

import os
import time

import numpy as np


def _weights_np(cfg):
    from helpers import ckpt_path
    from oracle.pan_oracle import ObsPointNetWeights
    w = ObsPointNetWeights.from_checkpoint(ckpt_path(cfg.checkpoint))
    return dict(W=w.W, b=w.b, gamma=w.gamma, beta=w.beta)


class _W:
    """ObsPointNetWeights look-alike built from arrays (workers do not import torch)."""

    def __init__(self, d):
        self.W, self.b, self.gamma, self.beta = d["W"], d["b"], d["gamma"], d["beta"]


def permuted_weights(wd, rng):
    """The same network with its five hidden layers' units permuted: identical in exact arithmetic (Linear and the
    element-wise layers commute with a permutation, LayerNorm's mean / variance are symmetric), different fp32
    summation order."""
    W = [a.copy() for a in wd["W"]]
    b = [a.copy() for a in wd["b"]]
    g = [a.copy() for a in wd["gamma"]]
    be = [a.copy() for a in wd["beta"]]
    prev = None
    for layer in range(5):                       # Linear 0..4 have 32 outputs; Linear 5 keeps its E outputs
        p = rng.permutation(32)
        if prev is not None:
            W[layer] = W[layer][:, prev]
        W[layer] = W[layer][p]
        b[layer] = b[layer][p]
        if layer in (0, 2, 4):                   # followed by LayerNorm 0, 1, 2
            g[layer // 2] = g[layer // 2][p]
            be[layer // 2] = be[layer // 2][p]
        prev = p
    W[5] = W[5][:, prev]
    return dict(W=[np.ascontiguousarray(a) for a in W], b=b, gamma=g, beta=be)


def _make_oracle(cfg, wd):
    from helpers import robot_numbers
    from oracle.pan_oracle import PanOracle
    G, h, sp, ac, L = robot_numbers(cfg.robot, cfg.dt)
    return PanOracle(cfg.T, cfg.dt, G, h, _W(wd), cfg.robot["kinematics"], L, speed_bound=sp, acce_bound=ac,
                     iter_num=cfg.iter_num, dune_max_num=cfg.n_points, nrmp_max_num=cfg.nrmp_max_num,
                     iter_threshold=0.0, **dict(cfg.adjust))


def _trace_u(orc, sc, points=None):
    orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"] if points is None else points,
                sc["velocities"])
    return np.stack([t[1] for t in orc.trace]).astype(np.float32)          # (K, 2, T)


def ensemble_worker(job):
    """One worker process: for each of its scenes the timed base run (the CPU baseline) and the untimed ensemble
    members.  Returns ([(scene, base (K,2,T), members (n,K,2,T))], seconds of base runs)."""
    workload, scenes, wd, n_ulp, n_perm = job
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = "1"
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        pass
    from neupan_amd.scenes import CONFIGS, make_scene
    cfg = CONFIGS[workload]
    _make_oracle(cfg, wd)
    out, spent = [], 0.0
    for b in scenes:
        sc = make_scene(cfg, b)
        t0 = time.perf_counter()
        base = _trace_u(_make_oracle(cfg, wd), sc)
        spent += time.perf_counter() - t0
        members = []
        for m in range(n_ulp):
            rng = np.random.default_rng(7_000_003 * (b + 1) + m)
            up = rng.random(sc["points"].shape) < 0.5
            pts = np.where(up, np.nextafter(sc["points"], np.float32(np.inf)), np.nextafter(sc["points"], np.float32(-np.inf)))
            members.append(_trace_u(_make_oracle(cfg, wd), sc, pts.astype(np.float32)))
        for m in range(n_perm):
            rng = np.random.default_rng(9_000_011 * (b + 1) + m)
            members.append(_trace_u(_make_oracle(cfg, permuted_weights(wd, rng)), sc))
        out.append((b, base, np.stack(members) if members else np.zeros((0,) + base.shape, np.float32)))
    return out, spent


def run_ensemble(workload, scenes, cores, n_ulp=8, n_perm=4):
    """Returns (base [S,K,2,T], members [S,n,K,2,T], cpu plans/s of the base runs, cores used).
    Rate = scenes / (slowest worker's base-run time): `cores` worker processes x 1 thread over independent scenes."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    from neupan_amd.scenes import CONFIGS
    scenes = list(scenes)
    cores = max(1, min(cores, len(scenes)))
    wd = _weights_np(CONFIGS[workload])
    jobs = [(workload, scenes[w::cores], wd, n_ulp, n_perm) for w in range(cores)]
    if cores == 1:
        res = [ensemble_worker(jobs[0])]
    else:
        with ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as ex:
            res = list(ex.map(ensemble_worker, jobs))
    wall = max(r[1] for r in res)
    got = {}
    for part, _ in res:
        for b, base, mem in part:
            got[b] = (base, mem)
    base = np.stack([got[b][0] for b in scenes])
    members = np.stack([got[b][1] for b in scenes])
    return base, members, len(scenes) / wall, cores


def _l2(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return np.sqrt((d * d).sum(axis=(-1, -2)))


def spreads(base, members):
    """max pairwise control L2 over {base, members} per scene and iteration: [S, K]."""
    allm = np.concatenate([base[:, None], members], axis=1)           # [S, n+1, K, 2, T]
    n = allm.shape[1]
    sp = np.zeros(base.shape[:2])
    for i in range(n):
        for j in range(i + 1, n):
            sp = np.maximum(sp, _l2(allm[:, i], allm[:, j]))
    return sp


def judge(hip_trace, base, members, tol=1e-4, early=1e-5):
    """hip_trace [S,K,2,T] (controls after each PAN iteration of the HIP path).  Returns the report dict with the
    three verdicts A, B, C of the module docstring, the distributions, and the worst scenes side by side."""
    sp = spreads(base, members)                                        # [S, K]
    hip = _l2(hip_trace, base)                                         # [S, K]
    S, K = hip.shape
    diverged = sp > early
    first = np.where(diverged.any(axis=1), diverged.argmax(axis=1), K)  # first iteration with ensemble spread > early
    well = sp[:, -1] <= tol
    a_ok = bool((hip[well, -1] <= tol).all())
    b_ok = bool((hip[~well, -1] <= sp[~well, -1]).all())
    c_viol = []
    for s in range(S):
        k_end = int(first[s])
        if k_end > 0 and hip[s, :k_end].max() > early:
            c_viol.append(int(s))
    worst = np.argsort(-hip[:, -1])[:8]
    rep = {
        "scenes": int(S), "ensemble_members": int(members.shape[1]),
        "ctrl_l2_vs_oracle_median": float(np.median(hip[:, -1])), "max": float(hip[:, -1].max()),
        "frac_le_1e-4": float((hip[:, -1] <= tol).mean()),
        "scenes_well_posed": int(well.sum()),
        "max_over_well_posed": float(hip[well, -1].max()) if well.any() else None,
        "A_well_posed_all_le_tol": a_ok,
        "B_others_inside_envelope": b_ok,
        "C_le_1e-5_until_ensemble_diverges": len(c_viol) == 0,
        "C_violations": c_viol,
        "max_hip_before_divergence": float(max([hip[s, :int(first[s])].max() for s in range(S) if first[s] > 0], default=0.0)),
        "worst_scenes": [{"scene": int(s), "ctrl_l2": float(hip[s, -1]), "ensemble_spread": float(sp[s, -1]),
                          "ensemble_diverges_at_iter": int(first[s]) + 1 if first[s] < K else None,
                          "hip_l2_at_that_iter": float(hip[s, min(int(first[s]), K - 1)]),
                          "ensemble_spread_at_that_iter": float(sp[s, min(int(first[s]), K - 1)])}
                         for s in worst],
    }
    return rep, hip, sp
