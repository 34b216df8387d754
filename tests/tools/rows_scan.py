"""The DUNE stage's rows (mu, lam, points, distances of the M nearest per slice) on many scenes, default selection against the exact-key
build (NPA_DUNE_FP32KEYS=1: the encoder on every point, ranking on exact keys): bitwise equality scene by scene, at a scale the test
suite does not reach.  The key mode is read when a handle is created: two child processes.

    python tests/tools/rows_scan.py <workload> <scenes> [first scene]
"""
import hashlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(wl, n, s0, out):
    import torch
    from helpers import CONFIGS
    from gpu_helpers import make_gpu_pan
    from neupan_amd.scenes import make_batch
    cfg = CONFIGS[wl]
    pan = make_gpu_pan(cfg)
    digs, viol = [], 0
    for c0 in range(0, n, 256):
        b = make_batch(cfg, s0 + c0, min(256, n - c0))
        st = pan.dune_stage(b["nom_s"], b["points"], b.get("velocities"))
        arrs = [st[k].cpu().numpy() for k in ("mu", "lam", "pts", "dist", "count")]
        for i in range(arrs[0].shape[0]):
            h = hashlib.sha256()
            for a in arrs:
                h.update(np.ascontiguousarray(a[i]).tobytes())
            digs.append(h.hexdigest()[:16])
    a = pan.audit() if hasattr(pan, "audit") else {}
    np.savez(out, digs=np.array(digs), violations=int(a.get("violations", 0)), points=int(a.get("points", 0)))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    wl, n = sys.argv[1], int(sys.argv[2])
    s0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    res = {}
    for tag, env in (("default", {}), ("exact", {"NPA_DUNE_FP32KEYS": "1"})):
        f = f"/tmp/rows_{wl}_{tag}.npz"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wl, str(n), str(s0), f], env=dict(os.environ, **env),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-1500:]
        res[tag] = np.load(f)
    d, e = res["default"]["digs"], res["exact"]["digs"]
    bad = np.nonzero(d != e)[0]
    print(f"{wl}: {n} scenes from {s0}: rows differ from the exact-key build on {len(bad)} scenes {list(s0 + bad[:8])}; "
          f"audit of the default path: {int(res['default']['points'])} points, {int(res['default']['violations'])} violations")
    sys.exit(1 if len(bad) else 0)
