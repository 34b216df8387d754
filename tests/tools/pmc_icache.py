"""Instruction-cache counters of the hot kernels, each alone on the chip (the same command as pmc_collect.py's single-call passes):
requests / hits / misses of the SQC instruction cache and the wave-cycles spent waiting for an instruction (SQ_WAIT_INST_ANY).

    python tests/tools/pmc_icache.py [workload]            # on the GPU box; prints a table, writes gpurun_out/r06/pmc_icache_<workload>.json
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools")); sys.path.insert(0, ROOT)
import pmc_collect as pc

if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "diff_1k_T10_K10"
    out = {}
    for group in (["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"],
                  ["SQ_IFETCH", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES"]):
        vals, dur, full, (rc, tail) = pc.run_pass(group, wl)
        for k, v in vals.items():
            out.setdefault(k, {"kernel": full.get(k), "avg_ms": dur.get(k)}).update(v)
    for k, v in out.items():
        if "SQC_ICACHE_REQ" in v and v["SQC_ICACHE_REQ"]:
            v["icache_miss_frac"] = (v.get("SQC_ICACHE_MISSES", 0) + v.get("SQC_ICACHE_MISSES_DUPLICATE", 0)) / v["SQC_ICACHE_REQ"]
        if v.get("SQ_WAVE_CYCLES"):
            v["wait_inst_frac"] = v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"]
        print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()}))
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06", f"pmc_icache_{wl}.json"), "w"), indent=1)
