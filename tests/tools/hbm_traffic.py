"""HBM traffic per launch of the PAN kernels from rocprofv3 PMC counters (run on the GPU box):

    python tests/tools/hbm_traffic.py            # writes gpurun_out/traffic.json

Two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950, see
/opt/skills/guides/MI355X_MICROARCH.md, "rocprofv3 PMC slots"), --kernel-trace only.
Units: FETCH_SIZE / WRITE_SIZE are reported in KiB.  gfx950 correction (same guide, "HBM"):
FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads, so the read side is
reported both raw and doubled; our loads are 4-B-per-lane coalesced (256 B per wave instruction),
for which the doubling is the conservative (upper) figure.
"""
import collections, csv, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_pass(counter):
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu", "--inflight", "1"]
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    agg, n = collections.defaultdict(float), collections.Counter()
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(root, f))):
                    if r["Counter_Name"] != counter:
                        continue
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    agg[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: agg[k] / n[k] for k in agg}


def main():
    fetch = run_pass("FETCH_SIZE")
    write = run_pass("WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not any(s in k for s in ("dune_kernel", "select_kernel", "nrmp_qp_kernel")):
            continue
        f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        out[k] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
