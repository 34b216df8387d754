"""The bench line's contract (what the round driver parses) checked on the committed line of the last measured build, and
the tie between the tracked PMC file and the kernel sources it was measured on.  CPU only: nothing is run here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _line(name):
    return json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().split("\n")[-1])


def test_default_bench_line_carries_every_contract_field():
    d = _line("r06_bench_full.json")             # (the full record of the default command; its compact last line is r06_bench.json)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "plans/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 256 * d["n_gpus"] * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]      # whole-job plans / time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    # the CPU baseline is a sane one: a plan inside a worker of the winning level costs at most ~3x a plan on an idle host
    assert c["seconds_per_plan_in_worker"]["median"] <= 3.0 * c["seconds_per_plan_single_worker"] + 0.05, c
    assert len(c["sweep"]) >= 3 and c["value"] == max(r["plans_per_s"] for r in c["sweep"])
    p = d["parity"]
    assert p["A_well_posed_all_le_tol"] and p["C_le_1e-5_until_ensemble_diverges"]
    assert p["one_step"]["frac_le_tol"] >= 0.995 and p["one_step"]["median"] <= 2e-6          # verdict D
    assert 0 < p["well_posed_frac"] <= 1 and p["well_posed_only"]["plans_per_s"] > 0
    assert d["margin_audit"]["violations"] == 0 and d["margin_audit"]["points"] > 0
    assert d["host_issue_ms_per_step"] <= 0.5 * d["ms_per_step"]
    # round 4: the roofline is priced on interior-point iterations counted in the same run, every single-step deviation
    # above the tolerance is explained, and the other paths / configurations ride in the driver's line
    assert r["ipm_iterations_per_launch"] > 0 and "per_iteration" in r["flops_model"] and r["chip_aggregate"]["frac"] > r["frac"]
    assert p["one_step"]["unexplained"] == 0
    x = d["extra"]
    assert {"exact_fp32_keys", "network_keys_1", "network_keys_3"} <= set(x["paths"])
    for k, v in x["paths"].items():
        if k != "note":
            assert v["plans_per_s"] > 0 and 0 < v["dune_executed_mfma"]["frac"] < 1 and v["controls_equal_default_path"] and v["margin_violations"] == 0
    u = x["uniform_cloud"]
    assert u["plans_per_s"] > 0 and u["candidates_per_slice"]["share_overflow_to_exact_keys"] == 0 and u["parity"]["one_step"]["unexplained"] == 0
    # round 5: the same 20 batches in flight under other launch schedules (same scenes per step): the merged default is the best
    sh = {k: v for k, v in x["launch_shapes"].items() if isinstance(v, dict)}
    assert len(sh) >= 3 and all(v["scenes_per_step"] == d["config"]["scenes_per_gpu"] and 0 < v["plans_per_s"] < d["value"] for v in sh.values())
    assert "merged launch" in d["config"]["schedule"] and d["config"]["chains"] == 8 and d["config"]["batches_in_flight"] == 40
    assert d["roofline"]["scenes_per_launch"] == 5 * 256 and d["roofline"]["kernel"].startswith("nrmp_qp_group_kernel")
    oc = {k: v for k, v in x["other_configs"].items() if isinstance(v, dict)}
    assert len(oc) >= 5 and all(v["plans_per_s"] > 0 and v["margin_violations"] == 0 for v in oc.values())
    exact = {k: v for k, v in oc.items() if "bf16_rows" not in k}             # the bf16 ROWS tier is labelled as NOT meeting parity
    assert all(v["parity"]["one_step"]["unexplained"] == 0 and v["parity"]["one_step"]["stalled"] == 0 and
               v["parity"]["A_well_posed_all_le_tol"] and v["parity"]["C_le_1e-5_until_ensemble_diverges"] for v in exact.values())
    assert any("bf16_rows" in k and not v["parity"]["A_well_posed_all_le_tol"] for k, v in oc.items())
    assert oc["poly8_5k_T10_K10_bf16_keys"]["controls_equal_exact_path"] is True          # the bf16 KEY tier: bitwise the exact path
    # the second-stage key filter (the table-corrected geometric key): same controls, faster on the dense clouds
    for k in ("dyna_4k_T10_K10_batch1024", "poly8_5k_T10_K10_exact_fp32_rows"):
        w = oc[k]["without_key_table"]
        assert w["controls_equal"] is True and oc[k]["plans_per_s"] > 1.15 * w["plans_per_s"], (k, w)
    assert p["one_step"]["stalled"] == 0 and p["one_step"]["frac_le_tol"] == 1.0
    ee, hh = x["early_exit"], x["h2d_inclusive"]                              # SURVEY 8(d)'s second run; the PCIe-inclusive rate
    assert ee["iter_threshold"] == 0.1 and 2 <= ee["iterations_mean"] < d["config"]["K"] and ee["plans_per_s"] > d["value"]
    assert 0 < hh["plans_per_s"] < d["value"] and hh["bytes_per_step"] > 256 * 8 * 1000
    # the compact last line of the same run: what the driver parses
    c = _line("r06_bench.json")
    assert len(json.dumps(c)) <= 6144 and c["value"] == d["value"] and c["roofline"]["frac"] == d["roofline"]["frac"]
    assert c["cpu_baseline"]["value"] == d["cpu_baseline"]["value"] and c["parity"]["A"] is True and c["parity"]["D_stalled"] == 0
    # round 6: the closed-loop cycle rides in the line; the metric's own schedule (one launch chain per
    # 256-scene step) sits in the part of the compact line that is never dropped, next to the scenes per merged launch
    fc = x["fleet_cycle"]
    assert fc["robots"] == 256 and fc["shipped"]["K"] == 2 and fc["k10"]["K"] == d["config"]["K"]
    assert fc["shipped"]["robot_cycles_per_s"] > fc["k10"]["robot_cycles_per_s"] > 0 and 0 < fc["k10"]["front_end_share"] < fc["shipped"]["front_end_share"] < 1
    assert c["extra"]["fleet_cycle"]["k10"]["robot_cycles_per_s"] == fc["k10"]["robot_cycles_per_s"]
    assert c["config"]["scenes_per_launch"] == 1280 and 0 < c["config"]["plans_per_s_one_chain_per_step"] < c["value"]


def test_driver_flag_line_is_the_same_contract():
    """What the round driver runs (--steps 20 --warmup 5): one wave of chains, lower by its issue ramp."""
    d, full = _line("r06_bench_driver_flags.json"), _line("r06_bench.json")
    assert len(open(os.path.join(ROOT, "profiles", "r06_bench_driver_flags.json")).read().strip()) <= 6144      # the line the driver parses
    assert d["steps"] == 20 and d["warmup"] == 5 and d["metric"] == full["metric"] and d["config"]["workload"] == full["config"]["workload"]
    assert 0.7 * full["value"] <= d["value"] <= full["value"]
    for k in ("roofline", "cpu_baseline", "parity", "extra"):
        assert k in d, k
    assert d["roofline"]["pmc"]["current"]["nrmp_qp_group_kernel"] is True and d["roofline"]["frac"] > 0.03
    t = _line("r06_bench_torchrun1.json")
    assert t["n_gpus"] == 1 and t["value"] >= 0.93 * full["value"]         # one rank with a live RCCL communicator and its gathers
    # round 6: the same under the driver's flags -- the per-GPU value an N-GPU run multiplies.  -8 % against the plain process on
    # alternating runs (profiles/r06_torchrun_overhead.txt, r06_region_trace.txt: medians 735 k / 795 k over 24 + 8 runs); the two tracked
    # lines are single runs of a region that varies +-4 % each, hence the margin
    t20 = _line("r06_bench_torchrun1_driver_flags.json")
    assert t20["n_gpus"] == 1 and t20["steps"] == 20 and t20["value"] >= 0.85 * d["value"]


def test_tracked_pmc_file_matches_the_built_kernels():
    """bench.py prices the roofline with a kernel's record of profiles/r06_pmc.json only while the code the tree builds IS
    the code that was measured: by the fingerprint of the kernel's machine code (bench.kernel_isa_hash: function bytes +
    kernel descriptor of the measured instantiation, read from the built library), or -- where the ROCm LLVM tools are not
    installed -- by the hash of the source files the kernel is built from.  An edit of the dominant kernel that changes its
    code without a new counter run fails here; a comment, or another kernel added next to it, does not."""
    import bench
    from neupan_amd import build
    build.build(force=False, verbose=False)
    pj = bench.load_pmc(bench.WORKLOAD)
    assert pj["_file"] == "profiles/r06_pmc.json"              # the newest tracked record is the one the tree is priced with
    for name in ("nrmp_qp_kernel", "nrmp_qp_group_kernel", "select_geo_kernel", "select_geo_group_kernel"):
        k = pj["kernels"][name]
        isa = bench.kernel_isa_hash(k["kernel"])
        if isa is not None and k.get("isa_hash"):
            assert k["isa_hash"] == isa, name
        else:
            assert k["source_hash"] == bench.kernel_hash(name), name
        assert bench.record_is_current(name, k), name
    assert pj["kernels"]["nrmp_qp_group_kernel"]["scenes_per_launch"] == 1280 and pj["kernels"]["nrmp_qp_kernel"]["scenes_per_launch"] == 256
    k = pj["kernels"]["nrmp_qp_kernel"]
    isa = bench.kernel_isa_hash(k["kernel"])
    assert k["fp64_flops_per_launch"] > 0 and k["hbm_bytes_per_launch"] > 0 and 0 < k["valu_issue_frac"] < 1
    # the lookup itself: a template instantiation, a plain kernel, an unknown name
    if isa is not None:
        assert bench.kernel_isa_hash("stage_kernel") and bench.kernel_isa_hash("void nrmp_qp_kernel<20, 10, false, true, 2, false>")
        assert bench.kernel_isa_hash("nrmp_qp_kernel<20, 10, false, true, 2, false>") != isa
        assert bench.kernel_isa_hash("no_such_kernel<1>") is None


def test_compact_line_stays_under_the_drivers_limit():
    """The LAST stdout line of bench.py is what the round driver parses; round 4's grew to 21 KB and was not parsed.  compact()
    of the largest full record on file (every extra leg, parity listings, PMC dump, CPU sweep) must stay <= 6 KB and keep
    every contract key, the roofline and cpu_baseline objects and one figure per extra leg."""
    import bench
    full = _line("r04_bench_driver_flags.json")
    assert len(json.dumps(full)) > 20000
    c = bench.compact(full, "bench_full.json")
    txt = json.dumps(c)
    assert len(txt) <= bench.COMPACT_LIMIT == 6144, len(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    assert c["value"] == full["value"] and c["roofline"]["frac"] == full["roofline"]["frac"]
    assert "workload" in c["config"] and "model" not in c["config"]
    assert c["parity"]["A"] is True and c["parity"]["D_unexplained"] == 0
    assert set(c["extra"]["other_configs"]) == {k for k, v in full["extra"]["other_configs"].items() if isinstance(v, dict)}
    assert all(isinstance(v["plans_per_s"], float) for v in c["extra"]["other_configs"].values())
    # a record that would not fit sheds its least essential parts instead of growing
    fat = json.loads(json.dumps(full))
    fat["extra"]["other_configs"].update({f"pad{i}": fat["extra"]["other_configs"]["acker_2k_T20_K15"] for i in range(40)})
    assert len(json.dumps(bench.compact(fat, "x"))) <= bench.COMPACT_LIMIT


def test_bare_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command with another N) must
    not die in argument handling: it re-executes itself under torch.distributed.run, one rank per GPU.  Driven here without a
    GPU (--dry-run: gloo, a numpy stand-in for the step): rendezvous on 127.0.0.1, barrier + K steps + barrier, MAX over the
    ranks, exactly ONE JSON line on stdout, from rank 0, with n_gpus = 2."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["dry_run"] is True
    assert d["unit"] == "plans/s" and d["scaling"] == "weak" and d["higher_is_better"] is True and d["value"] > 0
    assert abs(d["value"] - 256 * 2 * 1e3 / d["ms_per_step"]) <= 2e-2 * d["value"]
    # one rank, launched the way the driver launches N > 1
    r1 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--dry-run"],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert json.loads([ln for ln in r1.stdout.strip().split("\n") if ln.startswith("{")][-1])["n_gpus"] == 1
