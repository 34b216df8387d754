"""How the QP kernel's aggregate throughput scales with the number of launches in flight (one stream each), QP alone:
tells whether co-resident QP waves overlap (latency bound) or share a saturated resource."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_helpers import make_gpu_pan
from helpers import CONFIGS
from neupan_amd.scenes import make_batch
from neupan_amd.pan import _ptr
from neupan_amd._lib import check

name = sys.argv[1] if len(sys.argv) > 1 else "diff_1k_T10_K10"
cfg = CONFIGS[name]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pan = make_gpu_pan(cfg)
batch = make_batch(cfg, 0, B)
st = pan.dune_stage(batch["nom_s"], batch["points"], batch.get("velocities"))
dev = pan.device
T = pan.T
ins = [pan._dev(batch[k]) for k in ("nom_s", "nom_u", "ref_s", "ref_us")]
NMAX = 16
outs = [(torch.empty((B, 3, T + 1), device=dev), torch.empty((B, 2, T), device=dev), torch.empty((B, 1, T), device=dev),
         torch.zeros((B, 16), dtype=torch.float64, device=dev)) for _ in range(NMAX)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NMAX)]
lib = pan._lib


def launch(j):
    o = outs[j]
    check(lib.npa_nrmp_stage(pan._h, B, _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]), _ptr(ins[3]), _ptr(st["mu"]), _ptr(st["lam"]),
                             _ptr(st["pts"]), _ptr(st["count"]), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(o[3]),
                             None, C.c_void_p(streams[j].cuda_stream)), "nrmp_stage")


for n in (1, 2, 3, 4, 6, 8, 12, 16):
    R = 20
    for j in range(n):
        launch(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(R):
        for j in range(n):
            launch(j)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s B=%d streams %2d: %.3f ms per launch-round, %.2f M scene-solves/s (x%.2f of one stream)" %
          (name, B, n, dt / R * 1e3, n * R * B / dt / 1e6, 0.0 if n == 1 else (n * R * B / dt) / base), flush=True)
    if n == 1:
        base = R * B / dt
