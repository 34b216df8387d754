"""Where a single-scene forward call spends its time on the GPU: kernel durations and the gaps between dependent launches.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/lat -o t -- python tests/tools/latency_trace.py run
    python tests/tools/latency_trace.py parse /tmp/lat

`run [n]` issues 40 forward calls of ONE scene (or 14 calls of each of n scenes, one scene per call) for each of the two latency configurations of bench.py (K = 10, N = 1000 and the
shipped K = 2, N -> 100), synchronising after each; `parse` groups the traced kernels by forward call (each starts with the
staging kernel) and prints, per configuration, the medians of: GPU span first start -> last end, time inside kernels by
kernel, time between kernels."""
import csv
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(n_scenes=1):
    import torch
    from gpu_helpers import make_gpu_pan
    from helpers import CONFIGS
    from neupan_amd.scenes import make_batch
    cfg = CONFIGS["diff_1k_T10_K10"]
    batch = make_batch(cfg, 0, n_scenes)
    full = [torch.from_numpy(batch[k]).cuda() for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
    for tag, over in (("K10_N1000", {}), ("shipped_K2_N100", dict(iter_num=2, dune_max_num=100, iter_threshold=0.1))):
        pan = make_gpu_pan(cfg, **over); pan.printed = True
        med = []
        for b in range(n_scenes):
            a = [x[b:b + 1].contiguous() for x in full]
            ts = []
            for _ in range(40 if n_scenes == 1 else 14):
                pan.reset_stop_state()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pan.forward_batch(*a)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            med.append(1e3 * np.median(ts[4:]))
        prof = pan.profile_read() if hasattr(pan, "profile_read") and n_scenes > 1 and False else None
        print("%s host call -> synchronised, %d scene(s) one at a time: mean of the per-scene medians %.4f ms, median %.4f, min %.4f, max %.4f  (NPA_QP_ASET=%r FROM=%r)"
              % (tag, n_scenes, np.mean(med), np.median(med), np.min(med), np.max(med), os.environ.get("NPA_QP_ASET"), os.environ.get("NPA_QP_ASET_FROM")), flush=True)
        del pan


def parse(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    calls, cur = [], None
    for s, e, n in rows:
        short = n.split("(")[0].replace("void ", "")
        short = short.split("<")[0]
        if short == "stage_kernel":
            cur = []; calls.append(cur)
        if cur is not None and short in ("stage_kernel", "select_geo_kernel", "nrmp_qp_kernel", "pan_fused_kernel", "select_kernel", "dune_kernel"):
            cur.append((s, e, short))
    by_len = {}
    for c in calls:
        by_len.setdefault(len(c), []).append(c)
    for n, cs in sorted(by_len.items()):
        if len(cs) < 8:
            continue
        cs = cs[4:]
        span = np.median([c[-1][1] - c[0][0] for c in cs]) / 1e3
        inside = {}
        for c in cs:
            acc = {}
            for s, e, k in c:
                acc[k] = acc.get(k, 0) + (e - s)
            for k, v in acc.items():
                inside.setdefault(k, []).append(v)
        gaps = np.median([sum(c[i + 1][0] - c[i][1] for i in range(len(c) - 1)) for c in cs]) / 1e3
        first_qp = np.median([[e - s for s, e, k in c if k == "nrmp_qp_kernel"][:1] or [0] for c in cs]) / 1e3
        print("forward calls of %d launches (%d traced): GPU span %.1f us; between kernels %.1f us (%.1f per gap); inside: %s; first QP launch %.1f us"
              % (n, len(cs), span, gaps, gaps / max(n - 1, 1),
                 ", ".join("%s %.1f us (%.1f per launch)" % (k, np.median(v) / 1e3, np.median(v) / 1e3 / sum(1 for x in cs[0] if x[2] == k)) for k, v in inside.items()),
                 first_qp))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    else:
        parse(sys.argv[2])
