"""Round 6: the GPU timeline of the driver's 20-step region (bench.py --gpus 1 --steps 20 --warmup 5), plain process against one rank
under torch.distributed.run: rocprofv3 --kernel-trace of each, then -- over the region's launches (the last 40 nrmp_qp_group_kernel
launches and everything between the first selection launch before them and the last kernel of the process) -- span, busy union,
idle gaps, and what runs after the last QP launch (the gathers' tail).

    gpurun --timeout 600 -- 'python tests/tools/region_trace.py'      -> stdout (copied to profiles/r06_region_trace.txt)
"""
import csv, glob, os, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu", "--no-latency", "--no-extras"]


def trace(tag, cmd):
    d = "/tmp/rt_" + tag
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--"] + cmd, cwd="/tmp", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    line = [l for l in r.stdout.split("\n") if l.startswith("{")]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for x in csv.DictReader(open(f)):
            rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"], x.get("Queue_Id", "")))
    rows.sort()
    return line[-1] if line else "", rows


def short(n):
    n = n.split("(")[0]
    return n[:60]


def report(tag, line, rows):
    import json
    qp = [i for i, r in enumerate(rows) if "nrmp_qp_group_kernel" in r[2]]
    sel = [i for i, r in enumerate(rows) if "select_geo_group_kernel" in r[2]]
    if len(qp) < 40 or len(sel) < 40:
        print(tag, "too few group launches:", len(qp), len(sel)); return
    i0 = sel[-40]
    while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 100_000 and "stage_group" in rows[i0 - 1][2]:      # (the chains' staging launches)
        i0 -= 1
    i1 = qp[-1]
    while i1 + 1 < len(rows) and rows[i1 + 1][0] - max(r[1] for r in rows[i0:i1 + 1]) < 150_000:        # (the gathers' tail: copies, the all-gather)
        i1 += 1
    reg = rows[i0:i1 + 1]
    t0, t1 = reg[0][0], max(r[1] for r in reg)
    last_qp_end = max(r[1] for r in reg if "nrmp_qp_group_kernel" in r[2])
    cur_e, gaps = t0, []
    for s, e, n, q in reg:
        if s > cur_e:
            gaps.append((cur_e - t0, s - cur_e))
        cur_e = max(cur_e, e)
    val = json.loads(line)["value"] if line else None
    rm = json.loads(line).get("region_ms") if line else None
    print(f"== {tag}: bench line {val} plans/s, region_ms {rm}")
    print(f"   {len(reg)} kernels; GPU span {1e-6 * (t1 - t0):.3f} ms; idle inside {1e-6 * sum(g for _, g in gaps):.3f} ms in {len(gaps)} gaps (largest: " +
          ", ".join(f"{1e-6 * g:.3f} ms at +{1e-6 * a:.3f}" for a, g in sorted(gaps, key=lambda x: -x[1])[:4]) + ")")
    print(f"   last QP launch ends at +{1e-6 * (last_qp_end - t0):.3f} ms; after it: " +
          "; ".join(f"+{1e-6 * (s - t0):.3f}..+{1e-6 * (e - t0):.3f} {short(n)[:34]}" for s, e, n, q in reg if s >= last_qp_end - 1000))
    # the four chains: their launches in time order (a chain = one queue)
    chains = {}
    for s, e, n, q in reg:
        if "group_kernel" in n and "stage" not in n:
            chains.setdefault(q, []).append((s, e, "S" if "select" in n else "Q"))
    for q, ls in sorted(chains.items(), key=lambda x: x[1][0][0]):
        print(f"   queue {q}: starts +{1e-6 * (ls[0][0] - t0):.3f}, ends +{1e-6 * (ls[-1][1] - t0):.3f}; S/Q durations (ms): " +
              " ".join(f"{k}{1e-6 * (e - s):.2f}" for s, e, k in ls))
    by = {}
    for s, e, n, q in reg:
        k = short(n); by.setdefault(k, [0, 0.0]); by[k][0] += 1; by[k][1] += 1e-6 * (e - s)
    for k, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:6]:
        print(f"      {c:4d} x {k:60s} {t:8.3f} ms summed, {t / c:.4f} each")


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for r in range(reps):
        line, rows = trace("plain", [sys.executable] + BENCH)
        report("plain process", line, rows)
        line, rows = trace("tr", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                                  "--master-port", str(29700 + r)] + BENCH)
        report("one rank under torch.distributed.run", line, rows)
