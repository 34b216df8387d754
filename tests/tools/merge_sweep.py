"""Throughput of bench.py's loop over the launch schedule: batches in flight x merged chains (npa_forward_batch_group merges the
steps of a chain into one launch per stage), for the driver's 20-step region and for whole rounds.  On the GPU box:

    python tests/tools/merge_sweep.py [out.txt]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None


def say(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    if out:
        out.write(line + "\n"); out.flush()


dev = torch.device("cuda", 0)
bench.Loop.BURST = True
combos = [(20, 0), (20, 2), (20, 4), (20, 5), (20, 10), (24, 3), (24, 4), (24, 6), (32, 4), (32, 8), (40, 5), (40, 8), (16, 2), (16, 4)]
if os.environ.get("SWEEP_COMBOS"):
    combos = [tuple(int(v) for v in c.split(":")) for c in os.environ["SWEEP_COMBOS"].split(",")]
say("inflight chains | 20 steps (w5): plans/s median [min max] | whole rounds (6 x inflight, w 2 x inflight): median | sel ms  qp ms")
for nfl, ch in combos:
    lp = bench.Loop(bench.WORKLOAD, bench.BATCH, nfl, dev, issue_threads=4, chains=ch)
    a, b = [], []
    for _ in range(5):
        r = lp.timed(20, 5)
        a.append(bench.BATCH * 20 / r["elapsed"])
    for _ in range(3):
        r = lp.timed(6 * nfl, 2 * nfl)
        b.append(bench.BATCH * 6 * nfl / r["elapsed"])
    say(f"{nfl:3d} {lp.chains:3d} | {np.median(a):9.0f} [{min(a):9.0f} {max(a):9.0f}] | {np.median(b):9.0f} [{min(b):9.0f} {max(b):9.0f}] | "
        f"{r['prof']['select_ms']:.4f} {r['prof']['nrmp_ms']:.4f}  audit {lp.audit()['violations']}")
    lp.close()
