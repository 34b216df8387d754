"""Issue order of a burst of steps: call by call (one npa_forward_batch_flags per step) against breadth-first
(npa_forward_batch_group, StepLoop(burst=True)), for the run lengths the round driver and the default bench use, with 0 / 1 /
2 / 4 issuing threads.  ONE process, one set of 20 planners; every (order, threads) pair gets `reps` timed regions of each
length, bracketed by synchronize like bench.py's.

    python tests/tools/burst_sweep.py [reps]            # on the GPU box
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
import torch
import bench
from neupan_amd.serve import StepLoop

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
lp = bench.Loop(bench.WORKLOAD, bench.BATCH, 20, dev, issue_threads=4)
ref = None
rows = []
for burst in (False, True):
    for threads in (4, 2, 1, 0):
        lp.loop.close()
        lp.loop = StepLoop(lp.steps, lp.streams, lp.gatherer, lp.cur, threads=threads, burst=burst)
        assert (lp.loop.groups is not None) == burst
        for steps, warm in ((20, 5), (40, 5), (128, 32)):
            vals, iss = [], []
            for _ in range(reps if steps <= 40 else 2):
                r = lp.timed(steps, warm)
                vals.append(bench.BATCH * steps / r["elapsed"]); iss.append(1e3 * r["t_issue"] / steps)
                u = r["last"][0][0]["opt_u"].cpu().numpy()
                if ref is None:
                    ref = u.copy()
                assert np.array_equal(u, ref), "the issue order changed a result"
            rows.append({"burst": burst, "threads": threads, "steps": steps, "plans_per_s_median": round(float(np.median(vals))),
                         "plans_per_s_all": [round(v) for v in vals], "host_issue_ms_per_step": round(float(np.median(iss)), 4)})
            print(json.dumps(rows[-1]), flush=True)
lp.close()
