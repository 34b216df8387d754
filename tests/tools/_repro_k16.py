import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_helpers import make_gpu_pan
from helpers import CONFIGS
from neupan_amd.scenes import make_batch
def say(*a):
    print(*a, flush=True)
def mk(cfg, env):
    old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
    try:
        p = make_gpu_pan(cfg); torch.cuda.synchronize(); return p
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v
for name in sys.argv[1:] or ["poly8_5k_T10_K10"]:
    cfg = CONFIGS[name]
    say(name, "create default"); ref = mk(cfg, {})
    say("create k16"); k16 = mk(cfg, {"NPA_KEYS_PRECISION": "bf16"})
    say("report", k16.geo_report())
    batch = make_batch(cfg, 21000, 24)
    for rep in range(3):
        say("create dbg", rep); dbg = mk(cfg, {"NPA_KEYS_PRECISION": "bf16", "NPA_SEL_DEBUG": "1"})
        say("stage dbg"); c = dbg.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"))["count"].cpu().numpy() >> 16
        say("shares", [float((c == v).mean()) for v in range(4)])
    say("stage ref"); a = ref.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities")); torch.cuda.synchronize()
    say("stage k16"); b = k16.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities")); torch.cuda.synchronize()
    say("equal", all(np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy()) for k in ("mu", "lam", "pts", "dist", "count")))
    say("audit", k16.audit(), ref.audit())
