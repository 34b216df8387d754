"""Aggregate throughput of the DUNE stage (stage_kernel + select_kernel) alone with 1 ... 16 launches in flight, one
stream each: the counterpart of qp_scaling.py for the other kernel of a PAN iteration."""
import os, sys, time, ctypes as C
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
NMAX = 16
pans = [make_gpu_pan(cfg) for _ in range(NMAX)]            # one handle (workspace) per launch in flight
pan = pans[0]
batch = make_batch(cfg, 0, B)
dev = pan.device
T, M, E = pan.T, pan.nrmp_max_num, pan.E
nom_s, points = pan._dev(batch["nom_s"]), pan._dev(batch["points"])
N = points.shape[2]
outs = [(torch.zeros((B, T + 1, M, E), device=dev), torch.zeros((B, T + 1, M, 2), device=dev), torch.zeros((B, T + 1, M, 2), device=dev),
         torch.zeros((B, T + 1, M), device=dev), torch.zeros((B, T + 1), dtype=torch.int32, device=dev)) for _ in range(NMAX)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NMAX)]


def launch(j):
    o = outs[j]
    check(pans[j]._lib.npa_dune_stage(pans[j]._h, B, N, _ptr(nom_s), _ptr(points), None, None, _ptr(o[0]), _ptr(o[1]), _ptr(o[2]),
                                      _ptr(o[3]), _ptr(o[4]), C.c_void_p(streams[j].cuda_stream)), "npa_dune_stage")


for n in (1, 2, 4, 8, 12, 16):
    R = 40
    for j in range(n):
        launch(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(R):
        for j in range(n):
            launch(j)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s B=%d streams %2d: %.3f ms per launch-round, %.1f us per stage launch aggregated, %.2f M scene-stages/s" %
          (name, B, n, dt / R * 1e3, dt / R / n * 1e6, n * R * B / dt / 1e6), flush=True)
