"""PMC evidence for bench.py's roofline object, per kernel, tracked: writes gpurun_out/r06/pmc_<workload>.json (copy it to
profiles/r06_pmc.json; other workloads: profiles/r06_pmc_<tag>.json).

    python tests/tools/pmc_collect.py [workload]            # on the GPU box

Separate rocprofv3 passes (--kernel-trace + ONE --pmc group each; FETCH_SIZE and WRITE_SIZE cannot share a pass, see
/opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots") of `bench.py --inflight 1 --no-cpu --no-latency`, i.e. every
kernel alone on the chip, 256 scenes per launch.  Per kernel: average of every counter over its dispatches, the average
duration from the kernel trace of the same pass, and derived figures:
  valu_issue_frac  = SQ_ACTIVE_INST_VALU [quad-cycles] / (GRBM_GUI_ACTIVE per XCD [cycles] x 1024 SIMDs / 4)
  fp64_flops       = 64 lanes x (ADD_F64 + MUL_F64 + 2 FMA_F64) wave-instructions (an upper bound: lanes may be masked off)
  hbm_bytes        = FETCH_SIZE [KiB] x 1024 (x2 only where the kernel streams 16 B/lane -- none of ours do: the reads are
                     4-B/lane gathers, counted at face value, see the guide's HBM section) + WRITE_SIZE [KiB] x 1024
Every kernel's record carries `isa_hash` = a fingerprint of the machine code of the instantiation that was measured
(bench.kernel_isa_hash) and `source_hash` = the hash of the source files it is built from (bench.kernel_hash): bench.py marks a
record current while the built code is the measured code (the source hash decides where the LLVM tools are missing)."""
import collections, csv, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GROUPS = [
    ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
     "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS"],
    ["SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES",
     "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_ACTIVE_INST_LDS",
     "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]
KERNELS = ("dune_kernel", "select_geo_kernel", "select_geo_group_kernel", "select_kernel", "nrmp_qp_kernel", "nrmp_qp_group_kernel",
           "stage_kernel", "stage_group_kernel")


def short(name):
    n = name.replace("void ", "")
    for k in KERNELS:
        if n.startswith(k):
            return k
    return None


BENCH_ARGS = ["--steps", "4", "--warmup", "1", "--no-cpu", "--no-latency", "--no-extras", "--inflight", "1"]
# the merged-launch kernels (round 5): ONE chain of five steps, i.e. every stage one launch over 5 x 256 scenes, alone on the chip
MERGED_ARGS = ["--steps", "5", "--warmup", "5", "--no-cpu", "--no-latency", "--no-extras", "--inflight", "5", "--chains", "1"]
MERGED_SCENES = 5 * 256
# (no create-time self-test in the profiled process: its 2-scene launches of the same kernels would be averaged in with the
# 256-scene ones -- round 3's record has them: 9 of 79 QP launches, the per-launch averages of that file are ~10 % low)
BASE_ENV = {"TMPDIR": "/tmp", "NPA_SKIP_SELFTEST": "1"}


def bench_iterations(workload, extra_env=None):
    """Interior-point iterations per QP launch of the SAME command (bench.py measures them on the device: roofline.ipm_iterations_per_launch)."""
    env = dict(os.environ, **BASE_ENV, **(extra_env or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *BENCH_ARGS, "--workload", workload], cwd="/tmp", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    line = json.loads(r.stdout.strip().split("\n")[-1])
    return float(line["roofline"]["ipm_iterations_per_launch"])


def run_pass(counters, workload, extra_env=None, bench_args=None):
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    env = dict(os.environ, **BASE_ENV, **(extra_env or {}))
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), *(bench_args or BENCH_ARGS), "--workload", workload]
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    agg, n = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    dur, nd = collections.defaultdict(float), collections.Counter()
    full = {}
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for row in csv.DictReader(open(os.path.join(root, f))):
                    k = short(row["Kernel_Name"])
                    if k is None:
                        continue
                    full[k] = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
            if f.endswith("kernel_trace.csv"):
                for row in csv.DictReader(open(os.path.join(root, f))):
                    k = short(row["Kernel_Name"])
                    if k is None:
                        continue
                    dur[k] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6; nd[k] += 1
    out = {k: {c: v / n[(k, c)] for c, v in cs.items()} for k, cs in agg.items()}
    return out, {k: dur[k] / nd[k] for k in dur}, full, (r.returncode, r.stdout[-400:])


def derive(e, c):
    """Per-wave picture from the SQ counters (quad-cycle units cancel in the ratios)."""
    if c.get("SQ_WAVE_CYCLES"):
        e["wave_valu_busy_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]     # share of a wave's life issuing VALU
        e["wave_wait_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]                # ... parked on s_waitcnt / barriers
    if c.get("SQ_INSTS_VALU"):
        f64 = sum(c.get("SQ_INSTS_VALU_%s_F64" % k, 0.0) for k in ("ADD", "MUL", "FMA", "TRANS"))
        e["fp64_share_of_valu_insts"] = f64 / c["SQ_INSTS_VALU"]
        e["cycles_per_valu_inst"] = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_INSTS_VALU"]
    if c.get("SQ_WAVES"):
        e["valu_insts_per_wave"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVES"]
        e["lds_insts_per_wave"] = c.get("SQ_INSTS_LDS", 0.0) / c["SQ_WAVES"]


def main():
    from bench import source_hash, kernel_hash, kernel_isa_hash, BATCH
    workload = sys.argv[1] if len(sys.argv) > 1 else "diff_1k_T10_K10"
    kern = collections.defaultdict(dict)
    durs = collections.defaultdict(list)
    names, log = {}, []
    for args_ in (BENCH_ARGS, MERGED_ARGS):
        merged = args_ is MERGED_ARGS
        for g in GROUPS:
            vals, dur, full, (rc, tail) = run_pass(g, workload, bench_args=args_)
            # (a merged run's other kernels -- the priming forward calls of PAN.make_step -- are single-call launches: dropped)
            vals = {k: v for k, v in vals.items() if ("group" in k) == merged}
            log.append({"counters": g, "rc": rc, "merged": merged, "kernels_seen": sorted(vals)})
            if rc != 0 or not vals:          # an unknown counter kills the pass: retry one by one
                for c in g:
                    v1, d1, f1, (rc1, _) = run_pass([c], workload, bench_args=args_)
                    v1 = {k: v for k, v in v1.items() if ("group" in k) == merged}
                    log.append({"counters": [c], "rc": rc1, "merged": merged, "kernels_seen": sorted(v1)})
                    for k in v1:
                        kern[k].update(v1[k]); durs[k].append(d1.get(k, 0.0)); names.update(f1)
                continue
            names.update(full)
            for k in vals:
                kern[k].update(vals[k]); durs[k].append(dur.get(k, 0.0))
    res = {"source_hash": source_hash(), "workload": workload, "scenes_per_launch": BATCH,
           "command": "rocprofv3 --kernel-trace --pmc <group> -- python bench.py " + " ".join(BENCH_ARGS),
           "command_merged": "rocprofv3 --kernel-trace --pmc <group> -- python bench.py " + " ".join(MERGED_ARGS),
           "passes": log, "kernels": {}}
    for k, c in kern.items():
        ms = sum(durs[k]) / max(len(durs[k]), 1)
        e = {"kernel": names.get(k, k), "avg_ms_alone": ms, "counters": c, "source_hash": kernel_hash(k),
             "scenes_per_launch": MERGED_SCENES if "group" in k else BATCH,
             "isa_hash": kernel_isa_hash(names.get(k, k))}          # (machine code of the instantiation that was measured)
        if c.get("SQ_ACTIVE_INST_LDS"):
            e["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_ACTIVE_INST_LDS"]
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui and ms > 0:
            per_xcd = gui / max(1, round(gui / (ms * 1e-3 * 2.3e9)))      # the CSV may hold the sum over the 8 XCDs
            e["gui_active_cycles_per_xcd"] = per_xcd
            if "SQ_ACTIVE_INST_VALU" in c:
                e["valu_issue_frac"] = c["SQ_ACTIVE_INST_VALU"] / (per_xcd * 1024 / 4)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (per_xcd * 1024)
        if "SQ_INSTS_VALU" in c:
            e["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
        e["fp64_flops_per_launch"] = 64.0 * (c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) +
                                             2 * c.get("SQ_INSTS_VALU_FMA_F64", 0))
        derive(e, c)
        f, w = c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
        e["fetch_bytes_raw"], e["write_bytes"], e["hbm_bytes_per_launch"] = f, w, f + w
        res["kernels"][k] = e
    # ---- fp64 flops of the QP kernel per unit of work: a second operating point (NPA_QP_COLD=1: every solve cold, about twice
    # the interior-point iterations) and the iterations per launch of both, measured on the device by the same command:
    #     flops per launch = per_iteration x iterations per launch + per_solve x scenes per launch
    try:
        flop = lambda c: 64.0 * (c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * c.get("SQ_INSTS_VALU_FMA_F64", 0))
        vals_c, _, _, (rc_c, _) = run_pass(GROUPS[0], workload, {"NPA_QP_COLD": "1"})
        it_w, it_c = bench_iterations(workload), bench_iterations(workload, {"NPA_QP_COLD": "1"})
        f_w, f_c = flop(kern["nrmp_qp_kernel"]), flop(vals_c["nrmp_qp_kernel"])
        a = (f_c - f_w) / (it_c - it_w)
        b = (f_w - a * it_w) / BATCH
        res["qp_flops_model"] = {"per_iteration": a, "per_solve": b, "fit": {"warm": {"ipm_iterations_per_launch": it_w, "fp64_flops_per_launch": f_w},
                                                                                  "cold": {"ipm_iterations_per_launch": it_c, "fp64_flops_per_launch": f_c}},
                                 "note": "fp64 flops (64 lanes x (ADD + MUL + 2 FMA) wave-instructions, PMC) of one nrmp_qp_kernel launch at two "
                                         "operating points (warm start on / NPA_QP_COLD=1), iterations per launch measured on the device by the "
                                         "same command; per_iteration = slope, per_solve = intercept / scenes"}
    except Exception as e:          # (the record stays usable without the fit: bench.py then scales the launch average)
        res["qp_flops_model_error"] = repr(e)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r06", f"pmc_{workload}.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in res["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
