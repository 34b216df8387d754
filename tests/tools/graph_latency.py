"""Latency of a forward call issued launch by launch vs recorded into a HIP graph (torch.cuda.CUDAGraph around
PAN.forward_batch on static tensors) and replayed as one graph launch.  Measured rejection, DESIGN.md section 3.4: the
~10 us between two dependent launches of a stream is the dependency itself (completion -> next dispatch), not the
host: a graph recovers 0.04 ms of a 1.6 ms single-scene call."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_helpers import make_gpu_pan
from helpers import CONFIGS
from neupan_amd.scenes import make_batch

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "diff_1k_T10_K10"]
for B in (1, 256):
    pan = make_gpu_pan(cfg); pan.printed = True
    batch = make_batch(cfg, 0, B)
    a = [torch.from_numpy(batch[k]).cuda() for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]

    def plain():
        o = pan.forward_batch(*a, reset_state=True)
        torch.cuda.synchronize()
        return o
    ref = plain()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); plain(); ts.append(time.perf_counter() - t0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            pan.forward_batch(*a, reset_state=True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = pan.forward_batch(*a, reset_state=True)
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(out[k], ref[k]) for k in ("opt_s", "opt_u", "min_distance", "iters"))
    tg = []
    for _ in range(50):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); tg.append(time.perf_counter() - t0)
    print("B=%d: launch by launch %.4f ms, graph replay %.4f ms (median of 50, host call -> results synchronised); outputs bitwise equal: %s"
          % (B, 1e3 * np.median(ts), 1e3 * np.median(tg), same))
