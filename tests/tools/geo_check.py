"""GPU check of the geometric-key selection: key mode per checkpoint, bitwise equality of the DUNE stage with the
exact-key build on bench / wall / far-cloud scenes, candidate-count statistics (NPA_SEL_DEBUG=1) and stage timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_helpers import make_gpu_pan, wall_batch
from helpers import CONFIGS
from neupan_amd.scenes import make_batch


def mk(cfg, env=None, **kw):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    try:
        t0 = time.perf_counter()
        p = make_gpu_pan(cfg, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v
    return p, dt


def same(a, b):
    return all(np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy()) for k in ("mu", "lam", "pts", "dist", "count"))


for name, B in (("diff_1k_T10_K10", 256), ("acker_2k_T20_K15", 32), ("dyna_4k_T10_K10", 32), ("poly8_5k_T10_K10", 16)):
    cfg = CONFIGS[name]
    pan, dt = mk(cfg)
    exact, _ = mk(cfg, {"NPA_DUNE_FP32KEYS": "1"})
    print(name, "key_mode", pan.key_mode(), "create %.3f s" % dt, flush=True)
    gr = pan.geo_report()
    print("   geometric key: error %.4f margin %.4f m | table-corrected key: error %.2e margin %.2e m" %
          (gr["measured_error"], gr["margin"], gr["table_key_error"], gr["table_key_margin"]), flush=True)
    notab, _ = mk(cfg, {"NPA_GEO_TABLE": "0"})
    batch = make_batch(cfg, 1000, B)
    a = pan.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"))
    b = exact.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"))
    print("   bench scenes bitwise equal to exact keys:", same(a, b), "| to the handle without the table:",
          same(a, notab.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"))))
    if pan.key_mode()["key_terms"] != 4:
        geo, _ = mk(cfg, {"NPA_KEY_TERMS": "4"})
        print("   forced geometric:", geo.key_mode(), "bitwise:", same(geo.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities")), b))
    # far clouds: the same scenes pushed out / blown up
    for scale, shift in ((1.0, 40.0), (6.0, 0.0), (20.0, 0.0), (1.0, 300.0)):
        pts = (batch["points"] * np.float32(scale)).copy(); pts[:, 1] += np.float32(shift)
        a = pan.dune_stage(batch["nom_s"], pts, batch.get("velocities"))
        b = exact.dune_stage(batch["nom_s"], pts, batch.get("velocities"))
        print("   far cloud scale %g shift %g: bitwise %s" % (scale, shift, same(a, b)))
    if name == "diff_1k_T10_K10":
        wb = wall_batch(cfg, 64)
        a = pan.dune_stage(wb["nom_s"], wb["points"], None, wb["n_points"])
        b = exact.dune_stage(wb["nom_s"], wb["points"], None, wb["n_points"])
        print("   wall / blob scenes bitwise:", same(a, b))
    # candidate statistics
    dbg, _ = mk(cfg, {"NPA_SEL_DEBUG": "1"})
    os.environ["NPA_SEL_DEBUG"] = "1"
    c = dbg.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"))["count"].cpu().numpy()
    del os.environ["NPA_SEL_DEBUG"]
    nc, fb = (c >> 8) & 0xFF, c >> 16
    print("   candidates per slice: median %d mean %.1f p90 %d max %d; >32: %.3f; decided by the table filter: %.4f; shortened by it, then "
          "exact keys: %.4f; exact keys for the whole list: %.4f" %
          (np.median(nc), nc.mean(), np.quantile(nc, 0.9), nc.max(), (nc > 32).mean(), (fb == 3).mean(), (fb == 2).mean(), (fb == 1).mean()))
    print("   audit after all of the above:", pan.audit(), flush=True)
    # stage timing
    for p, tag in ((pan, "default"), (notab, "NPA_GEO_TABLE=0"), (exact, "exact keys")):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            p.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"))
        torch.cuda.synchronize()
        print("   dune_stage (%s): %.3f ms per call incl. wrapper" % (tag, (time.perf_counter() - t0) / 20 * 1e3))
