"""LDS-side counters of the two hot kernels, alone on the chip (the passes of tests/tools/pmc_collect.py carry the VALU side):
how many LDS-array cycles a QP solve needs against its lifetime -- with eight QP waves per CU (two per SIMD) sharing ONE LDS,
is the saturated kernel bound by VALU issue or by the LDS?  Writes gpurun_out/r04/pmc_lds.json.

    python tests/tools/pmc_lds.py            # on the GPU box
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import pmc_collect as pc

GROUPS = [["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES",
           "SQ_WAVES", "GRBM_GUI_ACTIVE"],
          ["SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_MEM_VIOLATIONS", "SQ_LDS_ATOMIC_RETURN", "SQ_ACTIVE_INST_SCA",
           "SQ_INSTS_SMEM", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_MISC"]]

if __name__ == "__main__":
    wl = "diff_1k_T10_K10"
    out = {"workload": wl, "command": "rocprofv3 --kernel-trace --pmc <group> -- python bench.py " + " ".join(pc.BENCH_ARGS), "kernels": {}, "passes": []}
    for g in GROUPS:
        vals, dur, full, (rc, tail) = pc.run_pass(g, wl)
        out["passes"].append({"counters": g, "rc": rc, "kernels_seen": sorted(vals)})
        if rc != 0 or not vals:
            for c in g:
                v1, d1, f1, (rc1, _) = pc.run_pass([c], wl)
                out["passes"].append({"counters": [c], "rc": rc1, "kernels_seen": sorted(v1)})
                for k in v1:
                    out["kernels"].setdefault(k, {"counters": {}})["counters"].update(v1[k]); out["kernels"][k]["avg_ms_alone"] = d1.get(k)
            continue
        for k in vals:
            out["kernels"].setdefault(k, {"counters": {}})["counters"].update(vals[k]); out["kernels"][k]["avg_ms_alone"] = dur.get(k)
    for k, e in out["kernels"].items():
        c = e["counters"]
        if c.get("SQ_WAVES") and c.get("SQ_LDS_IDX_ACTIVE") is not None:
            # SQ counters of this family come in units of 4 cycles (the VALU ones do, pmc_collect.py); report both readings
            e["lds_array_cycles_per_wave_if_unit_is_1"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_WAVES"]
            e["wave_cycles_per_wave_x4"] = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / c["SQ_WAVES"]
            e["lds_insts_per_wave"] = c.get("SQ_INSTS_LDS", 0.0) / c["SQ_WAVES"]
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r04"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04", "pmc_lds.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items()} for k, v in out["kernels"].items()}, indent=1))
