"""Round 6: copy what tests/tools/record_round6.sh wrote under gpurun_out/r06final/ (and, after --pmc, gpurun_out/r06/pmc_*.json) to the
tracked names under profiles/, and print the figures the documents quote.

    python tests/tools/adopt_records.py [--pmc]
"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r06final"), os.path.join(ROOT, "profiles")
MAP = {"bench.json": "r06_bench.json", "bench_full.json": "r06_bench_full.json", "bench_driver_flags.json": "r06_bench_driver_flags.json",
       "bench_driver_flags_full.json": "r06_bench_driver_flags_full.json", "bench_inflight1.json": "r06_bench_inflight1.json",
       "bench_torchrun1.json": "r06_bench_torchrun1.json", "gpu_tests.txt": "r06_gpu_tests.txt", "kernel_stats.csv": "r06_kernel_stats.csv",
       "kernel_stats_driver_flags.csv": "r06_kernel_stats_driver_flags.csv", "qp_phase_cycles.txt": "r06_qp_phase_cycles.txt",
       "select_phase_cycles.txt": "r06_select_phase_cycles.txt"}


def line(path):
    return json.loads(open(path).read().strip().split("\n")[-1])


if __name__ == "__main__":
    if "--pmc" in sys.argv:
        for w, n in (("diff_1k_T10_K10", "r06_pmc.json"), ("acker_2k_T20_K15", "r06_pmc_acker.json"), ("poly8_5k_T10_K10", "r06_pmc_poly8.json")):
            shutil.copy(os.path.join(ROOT, "gpurun_out", "r06", f"pmc_{w}.json"), os.path.join(DST, n))
    for a, b in MAP.items():
        shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
    open(os.path.join(DST, "r06_parity_wide.json"), "w").write(open(os.path.join(SRC, "parity_wide.json")).read().strip().split("\n")[-1] + "\n")
    tr = sorted((line(os.path.join(SRC, f"bench_torchrun1_driver_flags_{i}.json"))["value"], i) for i in (1, 2, 3))
    shutil.copy(os.path.join(SRC, f"bench_torchrun1_driver_flags_{tr[1][1]}.json"), os.path.join(DST, "r06_bench_torchrun1_driver_flags.json"))
    d, f = line(os.path.join(SRC, "bench.json")), line(os.path.join(SRC, "bench_driver_flags.json"))
    full = json.load(open(os.path.join(SRC, "bench_full.json")))
    e = full["extra"]
    print("default      ", d["value"], d["ms_per_step"], "select", d["roofline"]["select_launch_ms"], "qp", d["roofline"]["launch_ms"], "achieved", d["roofline"]["achieved"],
          "frac", d["roofline"]["frac"], "pipe", d["roofline"]["vector_pipe"]["frac"], "region", d.get("region_ms"))
    print("driver flags ", f["value"], f["ms_per_step"], "select", f["roofline"]["select_launch_ms"], "qp", f["roofline"]["launch_ms"], "frac", f["roofline"]["frac"],
          "pipe", f["roofline"]["vector_pipe"]["frac"], "region", f.get("region_ms"))
    print("  repeats    ", [round(line(os.path.join(SRC, f"drv_rep{i}.json"))["value"]) for i in (2, 3, 4, 5)])
    print("torchrun     ", line(os.path.join(SRC, "bench_torchrun1.json"))["value"], "20 steps:", [round(v) for v, _ in tr], "-> tracked:", round(tr[1][0]))
    i1 = line(os.path.join(SRC, "bench_inflight1.json"))
    print("inflight 1   ", i1["value"], i1["ms_per_step"], "select", i1["roofline"]["select_launch_ms"], "qp", i1["roofline"]["launch_ms"])
    print("early exit   ", e["early_exit"]["plans_per_s"], "h2d", e["h2d_inclusive"]["plans_per_s"], "fleet", e["fleet_cycle"]["shipped"]["robot_cycles_per_s"], e["fleet_cycle"]["k10"]["robot_cycles_per_s"])
    print("cpu baseline ", full["cpu_baseline"]["value"], full["cpu_baseline"]["cores"], "parity A/C", full["parity"]["A_well_posed_all_le_tol"], full["parity"]["C_le_1e-5_until_ensemble_diverges"],
          "frac_le_1e-4", full["parity"]["frac_le_1e-4"])
    for k, v in e["other_configs"].items():
        if isinstance(v, dict) and "plans_per_s" in v:
            print("  ", k, v["plans_per_s"], v["parity"].get("A_well_posed_all_le_tol"), v["parity"].get("C_le_1e-5_until_ensemble_diverges"))
    for w in ("acker_2k_T20_K15", "dyna_4k_T10_K10", "poly8_5k_T10_K10"):
        print("   GPU only", w, line(os.path.join(SRC, f"bench_{w}.json"))["value"])
    print(open(os.path.join(SRC, "gpu_tests.txt")).read().strip().split("\n")[-2:], open(os.path.join(SRC, "smoke.log")).read().strip().split("\n")[-2:])
