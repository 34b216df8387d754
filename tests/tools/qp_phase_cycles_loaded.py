"""Which phases of a QP solve stretch when the chip is full: the -DNPA_QP_PROF=1 build (s_memtime stamps between the phases of
an interior-point iteration, tests/tools/qp_phase_cycles.py) driven by bench.py's own loop with 1, 8 and 20 chains in flight.
Read-out: the last QP launch of chain 0 and of the middle chain (their scenes' accumulators in qp_info), i.e. a launch that
ran while the other chains were executing the end of their last step too.  VALU-heavy phases that stretch point at issue
contention, LDS-heavy ones (factorisation, substitutions) at the shared LDS.

    python tests/tools/qp_phase_cycles.py --mode=1 --build-only      # here (hipcc), the variant library travels with gpurun
    python tests/tools/qp_phase_cycles_loaded.py                      # on the GPU box
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ["NPA_SKIP_SELFTEST"] = "1"
import numpy as np
import torch
import qp_phase_cycles as qpc
import neupan_amd._lib as L
qpc.build_prof(1)
L.LIB_PATH = qpc.prof_lib(1)
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
names = qpc.NAMES[1]
base = None
for nfl in (1, 8, 20):
    lp = bench.Loop(bench.WORKLOAD, bench.BATCH, nfl, dev, issue_threads=min(4, nfl))
    lp.run(3 * nfl)
    torch.cuda.synchronize(dev)
    lp.run(3 * nfl)
    torch.cuda.synchronize(dev)
    rows = []
    for j in sorted({0, nfl // 2}):
        info = lp.pans[j].last_qp_info()
        rows.append(info)
    info = np.concatenate(rows, 0)
    cyc, its = info[:, 5:15], info[:, 4] + 1
    tot = cyc.sum(1).mean()
    print("%d chains in flight: last QP launch of chain(s) %s, %d scenes, iterations mean %.2f; s_memtime cycles per scene (mean)"
          % (nfl, sorted({0, nfl // 2}), info.shape[0], info[:, 4].mean()))
    per = {}
    for i, n in enumerate(names):
        per[i] = cyc[:, i].mean() if i in (0, 9) else (cyc[:, i] / its).mean()
        rel = "" if base is None else "   x%.2f of alone" % (per[i] / base[i])
        print("  %-62s %9.0f %s%s" % (n, per[i], "per solve    " if i in (0, 9) else "per iteration", rel))
    print("  total %.0f cycles per solve%s" % (tot, "" if base is None else "   x%.2f of alone" % (tot / base["tot"])))
    if base is None:
        base = dict(per, tot=tot)
    lp.close()
