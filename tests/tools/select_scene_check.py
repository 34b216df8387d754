"""The scene-wide selection (csrc/select_scene.h, NPA_SELECT_SCENE=1) against select_geo_kernel: rows of the DUNE stage bitwise,
how many slices took the per-slice body, a forward call bitwise, and the stage's launch time (one launch alone on the chip,
wall clock around a loop of synchronised calls).

    python tests/tools/select_scene_check.py [quick]            # on the GPU box
"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from gpu_helpers import make_gpu_pan, wall_batch
from helpers import CONFIGS
from neupan_amd.scenes import make_batch


def pans(cfg, **over):
    a = make_gpu_pan(cfg, **over)
    os.environ["NPA_SELECT_SCENE"] = "1"
    try:
        b = make_gpu_pan(cfg, **over)
    finally:
        del os.environ["NPA_SELECT_SCENE"]
    return a, b


def stats(p):
    out = (C.c_uint * 4)()
    f = p._lib.npa_dbg_select_stats
    f.restype, f.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_uint)]
    f(p._h, out)
    return list(out)


def stage(p, batch, **kw):
    return {k: v.cpu().numpy() for k, v in p.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"), **kw).items()}


quick = "quick" in sys.argv
cases = [("diff_1k_T10_K10", 256), ("dyna_4k_T10_K10", 32), ("acker_2k_T20_K15", 32), ("poly8_5k_T10_K10", 16)]
for name, B in (cases[:1] if quick else cases):
    cfg = CONFIGS[name]
    old, new = pans(cfg)
    batch = make_batch(cfg, 3000, B)
    r, e = stage(new, batch), stage(old, batch)
    st = stats(new)
    print(name, "B", B, {k: bool(np.array_equal(r[k], e[k])) for k in ("mu", "lam", "pts", "dist", "count")},
          "slices per-slice body / fast path:", st[1], "/", st[2], flush=True)
    if not np.array_equal(r["dist"], e["dist"]):
        bad = np.argwhere(r["dist"] != e["dist"])
        print("   first mismatches (scene, slice, rank):", bad[:6].tolist(), "of", len(bad), flush=True)
    if name == "diff_1k_T10_K10":
        wb = wall_batch(cfg, 32)
        r, e = stage(new, wb, n_points=wb["n_points"]), stage(old, wb, n_points=wb["n_points"])
        print("  walls / blobs:", {k: bool(np.array_equal(r[k], e[k])) for k in ("mu", "lam", "pts", "dist", "count")}, "stats", stats(new)[1:3], flush=True)
        rb = make_batch(cfg, 3100, 9)
        n_pts = np.array([0, 1, 5, 63, 64, 65, 255, 256, 257], dtype=np.int32)
        r, e = stage(new, rb, n_points=n_pts), stage(old, rb, n_points=n_pts)
        print("  ragged:", bool(np.array_equal(r["count"], e["count"])), all(np.array_equal(r[k][1:], e[k][1:]) for k in ("mu", "lam", "pts", "dist")), flush=True)
        args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
        a, b = old.forward_batch(*args), new.forward_batch(*args)
        print("  forward call:", {k: bool(np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy(), equal_nan=True)) for k in ("opt_s", "opt_u", "opt_d", "min_distance", "iters", "nrmp_points")},
              "stats", stats(new)[1:3], flush=True)
    # the stage's launch alone on the chip
    ns, pts = torch.from_numpy(batch["nom_s"]).cuda(), torch.from_numpy(batch["points"]).cuda()
    vel = torch.from_numpy(batch["velocities"]).cuda() if batch.get("velocities") is not None else None
    for tag, p in (("select_geo_kernel", old), ("select_scene_kernel", new)):
        p.dune_stage(ns, pts, vel)
        t0 = time.perf_counter()
        for _ in range(20):
            p.dune_stage(ns, pts, vel)
        print("   %-20s %.1f us per synchronised stage call" % (tag, 1e6 * (time.perf_counter() - t0) / 20), flush=True)
print("done", flush=True)
