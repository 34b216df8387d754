#!/usr/bin/env python3
"""bench.py -- MPC plans/s of the MI355X-native PAN inner solver on BASELINE.json configs[1]:
1 GPU (per rank), batch = 256 synthetic scenes, diff robot, 1000 obstacle points, T = 10,
K = 10 PAN iterations (iter_threshold = 0 so that exactly K run), fp32 DUNE + fp64 QP.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (one forward call: K x {selection, QP}) over one batch of 256 scenes per
rank, inputs already resident in HBM; with N > 1 ranks every rank plans its own 256 scenes (weak scaling, no
data-path collective) and the control outputs are all-gathered over RCCL inside the timed region.  Like any
serving loop the bench keeps `--inflight` independent batches in flight, each on its own HIP stream:
consecutive steps are different batches of 256 scenes, and the latency-bound kernels of one batch fill the
SIMDs the others leave idle.  Every step still executes its full K iterations inside the timed region;
`--inflight 1` gives the strictly sequential number (also reported: single-scene latency).  Rank 0 prints ONE
JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# One HIP stream per batch in flight; every stream should own a hardware queue, or two batches serialise behind each
# other.  The runtime's default is 4 queues; 32 covers the 16 batches in flight plus torch's own streams (sweep in
# with fewer queues than streams the throughput falls to what that many concurrent chains give).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOAD = "diff_1k_T10_K10"
BATCH = 256
# /opt/skills/guides/MI355X_MICROARCH.md, chip-level table (256 CUs x 4 SIMDs, 2.4 GHz)
PEAK_FP64_VALU_TFLOPS = 78.6
PEAK_FP32_MFMA_TFLOPS = 157.3       # v_mfma_f32_32x32x2_f32: what the exact encoder of select_geo_kernel runs on
PEAK_F16_MFMA_TFLOPS = 2500.0
N_SIMD = 1024
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc.json")
# which source files a kernel's counters depend on (a kernel's PMC record is used only while these are unchanged)
KERNEL_SOURCES = {"nrmp_qp_kernel": ("nrmp_qp.hip", "pan_common.h"), "select_geo_kernel": ("dune.hip", "pan_common.h"),
                  "select_kernel": ("dune.hip", "pan_common.h"), "dune_kernel": ("dune.hip", "pan_common.h"),
                  "stage_kernel": ("c_api.hip", "pan_common.h")}


def source_hash(files=None):
    """sha256 over kernel sources (all of neupan_amd/csrc, or the named files): ties the tracked counters to the code
    they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "neupan_amd", "csrc")
    for f in sorted(os.listdir(d)) if files is None else files:
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_hash(kernel):
    return source_hash(KERNEL_SOURCES.get(kernel, None))


def main():
    global BATCH
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--cpu-scenes", type=int, default=0,
                    help="scenes of the first batch planned by the CPU oracle + its ensemble (rank 0, N=1); 0 = 256 on a host "
                         "with >= 64 cores, else 96")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the CPU baseline (0 = all host cores)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-scene latency measurement")
    ap.add_argument("--graph", action="store_true", help="HIP-graph replay of a step's 21 launches for the planners that carry no "
                                                        "timing events (measured: SLOWER than eager launches with 20 chains in "
                                                        "flight, 513 k vs 656 k plans/s, DESIGN.md section 3.4; default: eager)")
    ap.add_argument("--inflight", type=int, default=20, help="independent batches (steps) kept in flight, one stream each")
    ap.add_argument("--issue-threads", type=int, default=4, help="host threads issuing the steps' launches (0: the main thread alone)")
    ap.add_argument("--workload", default=WORKLOAD,
                    choices=sorted(k for k in __import__("neupan_amd.scenes", fromlist=["CONFIGS"]).CONFIGS),
                    help="scene configuration (default: the one BASELINE.json's metric is quoted on)")
    ap.add_argument("--batch", type=int, default=BATCH, help="scenes per step and GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "RANK" in os.environ and "MASTER_PORT" in os.environ:      # under torch.distributed.run, also with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("NPA_BENCH_LAZY_PG"):       # diagnostics: no eager communicator
            dist.init_process_group("nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # the measured leg imports the product only; tests/ (and with it oracle/) is touched by the cpu_baseline leg alone
    from neupan_amd.pan import PAN
    from neupan_amd.serve import ControlGatherer, StepLoop, bind_to_gpu_numa_node, run_steps as serve_steps
    numa = bind_to_gpu_numa_node(local_rank)            # launch thread next to its GPU (one process per GPU)
    from neupan_amd.robot import Robot
    from neupan_amd.scenes import CONFIGS, make_batch

    def make_gpu_pan(cfg, device, **over):
        ck = os.path.join(ROOT, "tests", "golden", "checkpoints", f"{cfg.checkpoint}_model_5000.pth")
        if not os.path.exists(ck):              # the 8-edge stand-in: tests/golden/make_poly8_checkpoint.py
            ck = os.path.join(ROOT, "tests", "golden", "checkpoints", f"{cfg.checkpoint}_model_quick.pth")
        kw = dict(iter_num=cfg.iter_num, dune_max_num=cfg.n_points, nrmp_max_num=cfg.nrmp_max_num, iter_threshold=0.0,
                  dune_checkpoint=ck, adjust_kwargs=dict(cfg.adjust), device=device)
        kw.update(over)
        return PAN(cfg.T, cfg.dt, Robot(cfg.T, cfg.dt, **cfg.robot), **kw)

    if os.environ.get("NPA_BENCH_HIPRIO_STREAM"):          # diagnostics: does the mere existence of a high-priority stream cost?
        _hp = torch.cuda.Stream(device=dev, priority=-1)
        with torch.cuda.stream(_hp):
            torch.zeros(16, device=dev).add_(1)
        torch.cuda.synchronize(dev)
    if os.environ.get("NPA_BENCH_RCCL_COMM_ONLY"):        # diagnostics: a raw RCCL communicator (no torch process group)
        import ctypes as _C
        _r = _C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        _comm = _C.c_void_p()
        _devs = (_C.c_int * 1)(local_rank)
        print("ncclCommInitAll rc", _r.ncclCommInitAll(_C.byref(_comm), 1, _devs), file=sys.stderr)
    cfg = CONFIGS[args.workload]
    BATCH = args.batch
    T, K, N = cfg.T, cfg.iter_num, cfg.n_points
    nfl = max(1, args.inflight)
    pans = [make_gpu_pan(cfg, device=dev) for _ in range(nfl)]
    E = pans[0].E
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    args_dev = []
    for j in range(nfl):                    # batch j of this rank: its own 256 scenes
        batch = make_batch(cfg, (rank * nfl + j) * BATCH, BATCH)
        a = [torch.from_numpy(batch[k]).to(dev) for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
        a.append(torch.from_numpy(batch["velocities"]).to(dev) if batch.get("velocities") is not None else None)
        args_dev.append(a)
    torch.cuda.synchronize(dev)
    cur = torch.cuda.current_stream(dev)
    # One prepared step per batch in flight (PAN.make_step: arguments validated once, ONE library call per step, output
    # tensors reused -- a serving loop owns its buffers); fresh stop-criterion state every step, reset inside the staging
    # launch.  The controls of every step are all-gathered over RCCL from ONE communication stream (neupan_amd/serve.py).
    # Planners that carry timing events: every 4th (events ride on the dispatches).  --graph: the others replay a HIP graph
    # of the same launches (measured slower, kept for the record).
    timed_idx = set(range(0, nfl, 4)) if nfl >= 4 else set(range(nfl))
    # (NPA_BENCH_NOGATHER=1, diagnostics only: process group up, no gathers -- isolates what the collectives themselves cost)
    gatherer = ControlGatherer(None if os.environ.get("NPA_BENCH_NOGATHER") else dist, world, device=dev, slots=nfl, shape=(BATCH, 2, T))
    steps = []
    for j in range(nfl):
        with torch.cuda.stream(streams[j]):
            steps.append(pans[j].make_step(*args_dev[j], reset_every_step=True, graph=(args.graph and j not in timed_idx)))
    torch.cuda.synchronize(dev)

    loop = StepLoop(steps, streams, gatherer, cur, threads=args.issue_threads)

    def run_steps(n):
        """n steps (= n forward calls over batches of 256 scenes), `nfl` of them in flight, one stream each: step i is
        planned by planner i % nfl; the launches are issued by --issue-threads host threads (0: this thread alone)."""
        return loop.run(n)

    run_steps(args.warmup)
    torch.cuda.synchronize(dev)
    # HIP events ride on the launches of every 4th batch in flight (all of them with < 4): recording two events per launch
    # doubles the host's time to enqueue a step, and with 20 chains to start that ramp is what a short timed region
    # (the driver's 20 steps) mostly measures.  launch_ms below is the average over the launches that carry events.
    timed_pans = [pans[j] for j in sorted(timed_idx)]
    for p in timed_pans:
        p.profile(True)
    if dist is not None and not os.environ.get("NPA_BENCH_LAZY_PG"):
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    t_issue = time.perf_counter() - t0          # host time to enqueue every step (the GPU is still working)
    torch.cuda.synchronize(dev)
    if dist is not None and not os.environ.get("NPA_BENCH_LAZY_PG"):
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    profs = [p.profile_read() for p in timed_pans]
    for p in timed_pans:
        p.profile(False)
    if dist is not None and not os.environ.get("NPA_BENCH_LAZY_PG"):
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    nl = sum(q["launches"] for q in profs)
    avg = lambda key: sum(q[key] * q["launches"] for q in profs) / max(nl, 1)
    prof = {"launches": nl, "dune_ms": avg("dune_ms"), "select_ms": avg("select_ms"), "nrmp_ms": avg("nrmp_ms")}
    out, gathered = last[0]
    for o, g in (x for x in last if x is not None):
        assert (o["iters"].cpu().numpy() == K).all(), "every scene must run exactly K PAN iterations inside the timed region"
        assert g.numel() == world * BATCH * 2 * T

    plans = BATCH * world * args.steps
    value = plans / elapsed
    km = pans[0].key_mode()

    # the run-time audit of the geometric-key margin over everything this process launched so far (all planners): a
    # violation would mean plans computed from a wrong selection -- the number is reported AND must be zero
    audits = [p.audit() for p in pans]
    audit = {k: sum(a[k] for a in audits) for k in ("tiles", "points", "violations")}
    audit["worst_excess"] = max(a["worst_excess"] for a in audits)
    assert audit["violations"] == 0, f"geometric-key margin violated at run time: {audit}"

    # ---- roofline of the dominant kernel -------------------------------------------------------------------------
    # With geometric keys the encoder runs on ~2 % of the points (the candidates), and the step is the QP: a serial
    # fp64 interior-point chain, one wave per scene -- VALU/latency bound, no MFMA, no HBM stream.  EXECUTED work only:
    # counters from profiles/r03_pmc.json (tests/tools/pmc_collect.py: separate rocprofv3 --pmc passes), used per kernel
    # only while the sources that kernel is built from are unchanged (kernel_hash).  SURVEY 8(d)'s roofline -- algorithmic
    # dense flops of the encoder over the fp32-MFMA peak -- is retired: 943 Mflop/plan x this rate would be 2-4x that peak,
    # the encoder simply does not run on 98 % of the points any more.
    pmc = None
    if os.path.exists(PMC_FILE):
        try:
            pj = json.load(open(PMC_FILE))
            if pj.get("workload") == args.workload:
                pmc = pj
        except Exception:
            pmc = None

    def pmc_kernel(name):
        k = pmc["kernels"].get(name) if pmc else None
        return k if k and k.get("source_hash") == kernel_hash(name) else None
    scenes_per_launch = BATCH
    qp_ms, sel_ms, dune_ms = prof["nrmp_ms"], prof["select_ms"], prof["dune_ms"]
    roof = {"bound": "valu", "kernel": f"nrmp_qp_kernel<{T},{cfg.nrmp_max_num}>", "unit": "TFLOP/s", "peak": PEAK_FP64_VALU_TFLOPS,
            "launch_ms": round(qp_ms, 4), "launches_timed": nl, "select_launch_ms": round(sel_ms, 4),
            "dune_launch_ms": round(dune_ms, 4), "key_mode": km, "achieved": None, "frac": None, "traffic": None}
    kq = pmc_kernel("nrmp_qp_kernel")
    if kq:
        flops = kq["fp64_flops_per_launch"] * scenes_per_launch / pmc["scenes_per_launch"]
        roof["achieved"] = round(flops / (qp_ms * 1e-3) / 1e12, 4) if qp_ms > 0 else None
        roof["frac"] = round(roof["achieved"] / PEAK_FP64_VALU_TFLOPS, 5) if roof["achieved"] else None
        roof["flops_per_launch"] = int(flops)
        roof["traffic"] = int(kq["hbm_bytes_per_launch"] * scenes_per_launch / pmc["scenes_per_launch"])
        roof["valu_issue_frac_alone"] = kq.get("valu_issue_frac")
        roof["frac_alone"] = round(flops / (kq["avg_ms_alone"] * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS, 5) if kq.get("avg_ms_alone") else None
        roof["lds_bank_conflict_frac"] = kq.get("lds_bank_conflict_frac")
    if pmc:
        roof["pmc"] = {"file": "profiles/r03_pmc.json",
                       "per_kernel": {k: dict({kk: v[kk] for kk in ("valu_insts_per_launch", "valu_issue_frac", "hbm_bytes_per_launch",
                                                                      "avg_ms_alone", "mfma_busy_frac", "source_hash") if kk in v},
                                                  current=(v.get("source_hash") == kernel_hash(k)))
                                      for k, v in pmc["kernels"].items()}}
    ks = pmc_kernel("select_geo_kernel")
    if ks and sel_ms > 0 and "SQ_INSTS_MFMA" in ks.get("counters", {}):
        # the exact encoder's matrix work: executed v_mfma_f32_32x32x2_f32 (4096 flop each) over the fp32-MFMA peak
        mf = ks["counters"]["SQ_INSTS_MFMA"] * 4096.0 * scenes_per_launch / pmc["scenes_per_launch"]
        roof["select"] = {"kernel": f"select_geo_kernel<{E}>", "launch_ms": round(sel_ms, 4), "mfma_flops_per_launch": int(mf),
                          "mfma_tflops": round(mf / (sel_ms * 1e-3) / 1e12, 3), "mfma_peak": PEAK_FP32_MFMA_TFLOPS,
                          "mfma_frac": round(mf / (sel_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 5),
                          "valu_issue_frac_alone": ks.get("valu_issue_frac"), "mfma_busy_frac_alone": ks.get("mfma_busy_frac"),
                          "traffic": int(ks["hbm_bytes_per_launch"] * scenes_per_launch / pmc["scenes_per_launch"]),
                          "algorithmic_bytes_per_launch": int(scenes_per_launch * (8 * N * (2 if args_dev[0][5] is not None else 1) + 400))}
    roof["note"] = ("dominant kernel by GPU time = the QP (fp64 Mehrotra IPM, one wave per scene, serial chain: latency / VALU-issue "
                    "bound).  achieved = fp64 flops EXECUTED per launch (PMC: SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 x 64 lanes, FMA x2) / "
                    "launch time measured here with HIP events on the launch's stream (other batches' kernels co-run on the same "
                    "SIMDs; frac_alone: the same over the kernel's duration alone on the chip); nothing is priced above what it "
                    "executes.  DUNE: " +
                    ("geometric distance keys inside select_geo_kernel nominate the candidates, the exact fp32-MFMA encoder runs on "
                     "those only (~1.1 tiles of 32 points per slice instead of N/32); no dune_kernel launch" if km["key_terms"] == 4
                     else f"network keys (mode {km['key_terms']}) from dune_kernel over every point, exact re-encode of the candidates"))
    if km["key_terms"] != 4 and dune_ms > 0:
        slices = (T + 1) + (K - 1) * T
        nf16 = 8 if km["key_terms"] == 1 else 24
        tiles_per_launch = scenes_per_launch * slices / K * ((N + 31) // 32)
        ex = tiles_per_launch * (nf16 * 32768 + 4096) * 2 / 2 / (dune_ms * 1e-3) / 1e12
        roof["dune_executed_mfma"] = {"tflops": round(ex, 2), "peak": PEAK_F16_MFMA_TFLOPS, "frac": round(ex / PEAK_F16_MFMA_TFLOPS, 4)}

    line = {
        "metric": "MPC plans/sec (node), diff robot, 1k pts, T=10, K=10; ctrl L2 vs ref" if args.workload == WORKLOAD
                  else f"MPC plans/sec (node), workload {args.workload}; ctrl L2 vs ref",
        "value": round(value, 1), "unit": "plans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[1]: batch=256 synthetic scenes/GPU, diff robot, 1000 pts, "
                                "T=10, K=10 (iter_threshold=0), M=10, fp32 DUNE (MFMA) + fp64 QP") if args.workload == WORKLOAD
                               else f"{args.workload}: batch={BATCH} synthetic scenes/GPU, {cfg.kinematics} robot, {N} pts, "
                                    f"T={T}, K={K} (iter_threshold=0), M={cfg.nrmp_max_num}, fp32 DUNE (MFMA) + fp64 QP"
                                    + ("; BASELINE's 'bf16 DUNE' tier is NOT built: rows are exact fp32 (bf16 cannot hold 1e-4 "
                                       "through a top-M selection, SURVEY section 7), keys geometric" if args.workload.startswith("poly8") else ""),
                   "scenes_per_gpu": BATCH, "points": N, "T": T, "K": K, "M": cfg.nrmp_max_num,
                   "batches_in_flight": nfl, "schedule": "one HIP stream per batch in flight, one prepared library call per step "
                                                        "(PAN.make_step), " + ("eager launches" if not args.graph else
                                                        "HIP-graph replay except on the planners that carry timing events") +
                                                        ", gathers on one communication stream",
                   "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "numa_node": numa, "issue_threads": loop.threads,
                   "parallelism": f"scene-shard x{world}, RCCL all-gather of controls"
                                  + (f" (process group initialised; {gatherer.collectives} collectives for {gatherer.issued} steps: "
                                     f"one per {nfl} steps, all inside the timed loop)" if dist is not None else "")},
        "roofline": roof,
        # host time to enqueue a step (one library call = 21 launches, one Python thread) next to the step's wall time: when
        # the two are close the step is bound by the host's launch rate, not by the kernels
        "host_issue_ms_per_step": round(1e3 * t_issue / args.steps, 4),
        "margin_audit": audit,
    }

    # ---- single-scene latency: the reference's actual use (neupan/neupan.py:104-166, one robot, README "15 Hz") -------
    if rank == 0 and not args.no_latency:
        lat = {}
        for tag, over, npts in (("K10_N1000", {}, N), ("shipped_K2_N100", dict(iter_num=2, dune_max_num=100, iter_threshold=0.1), N)):
            p1 = make_gpu_pan(cfg, device=dev, **over)
            p1.printed = True                   # the reference prints a decimation notice once (pan.py:172): keep stdout to ONE line
            a1 = [a[:1].contiguous() if a is not None else None for a in args_dev[0]]
            ts = []
            for rep in range(60):
                p1.reset_stop_state()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                p1.forward_batch(*a1)
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t1)
            lat[tag] = round(1e3 * float(np.median(ts[10:])), 4)
            del p1
        line["latency_B1_ms"] = dict(lat, note="one scene per forward call, host call -> results synchronised, median of 50; "
                                               f"same workload ({N} points; the shipped config decimates to 100 and may stop early)")

    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from parity_tools import gpu_last_qp_certificates, judge, run_ensemble
        try:                                    # the CPU baseline gets the whole host, not the GPU's NUMA node
            os.sched_setaffinity(0, range(os.cpu_count() or 1))
        except Exception:
            pass
        host = os.cpu_count() or 1
        n_sc = args.cpu_scenes if args.cpu_scenes > 0 else (BATCH if host >= 64 else min(96, BATCH))
        n_sc = min(n_sc, BATCH)
        ncore = args.cpu_cores if args.cpu_cores > 0 else host
        base, members, cpu_rate, ncore = run_ensemble(args.workload, range(n_sc), ncore)
        # controls after every PAN iteration of the same batch, untimed; its last iteration IS the timed result
        timed_u = out["opt_u"].cpu().numpy().copy()
        pans[0].reset_stop_state()
        tr = pans[0].forward_batch_trace(*args_dev[0])
        trace_u = tr["trace_u"].cpu().numpy()
        assert np.array_equal(trace_u[:, -1], timed_u), "traced run differs from the timed run"
        rep, hip, sp = judge(trace_u[:n_sc], base, members)
        from parity_tools import host_cores, one_step_consistency, one_step_report
        # D: one oracle iteration from the HIP path's own iterate vs the HIP path's next iterate, every scene, every iteration
        rep["one_step"] = one_step_report(one_step_consistency(args.workload, range(n_sc), tr["trace_s"].cpu().numpy()[:n_sc],
                                                               trace_u[:n_sc], args.cpu_cores if args.cpu_cores > 0 else host))
        phys, logical = host_cores()
        quota = None
        try:                                    # cgroup v2 CPU quota of this container ("max" = none): the ceiling of any CPU baseline here
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else round(float(q) / float(per), 2)
        except Exception:
            pass
        line["cpu_baseline"] = {"value": round(cpu_rate, 3), "unit": "plans/s", "cores": ncore, "kind": "port",
                                "sample": f"scenes of the same workload (first {n_sc}, cycled), K={K} each, oracle/pan_oracle.py "
                                          f"(numpy fp32 + fp64 IPM): kind 'port' because /root/reference does not exist on the GPU "
                                          f"box (the reference's own a-2..a-6 cannot be imported there); worker processes x 1 "
                                          f"thread, BLAS/OpenMP pinned to 1 thread in the parent before the spawn; best rate of a "
                                          f"sweep over the number of concurrent workers ({ncore} won; host: {phys} physical cores, "
                                          f"{logical} hardware threads, cgroup CPU quota {quota}), 2-4 plans per worker, workers started and warm",
                                "cgroup_cpu_quota": quota,
                                "sweep": getattr(run_ensemble, "last_sweep", None),
                                "seconds_per_plan_single_worker": round(getattr(run_ensemble, "last_single_seconds", 0.0) or 0.0, 3)}
        js = getattr(run_ensemble, "last_job_seconds", None)
        if js:
            line["cpu_baseline"]["seconds_per_plan_in_worker"] = {"median": round(float(np.median(js)), 3), "max": round(float(max(js)), 3),
                                                                  "jobs_seen": len(js)}
        # conditioning of the WORKLOAD (a property of neupan_amd/scenes.py's generator, not of either implementation): the
        # share of scenes on which 13 equally valid evaluations of the reference algorithm agree to 1e-4 at all; and the
        # headline pair once more on those scenes only
        well = sp[:, -1] <= 1e-4
        rep["well_posed_frac"] = round(float(well.mean()), 4)
        if well.any() and n_sc == BATCH:
            idx = np.flatnonzero(well)
            idx = torch.from_numpy(np.resize(idx, BATCH)).to(dev)
            a_w = [a.index_select(0, idx).contiguous() if a is not None else None for a in args_dev[0]]
            st_w = []
            for j in range(nfl):
                with torch.cuda.stream(streams[j]):
                    st_w.append(pans[j].make_step(*a_w, reset_every_step=True))
            torch.cuda.synchronize(dev)
            nw = args.steps
            serve_steps(args.warmup, st_w, streams, None, cur)
            torch.cuda.synchronize(dev)
            tw = time.perf_counter()
            serve_steps(nw, st_w, streams, None, cur)
            torch.cuda.synchronize(dev)
            tw = time.perf_counter() - tw
            rep["well_posed_only"] = {"scenes": int(well.sum()), "plans_per_s": round(BATCH * nw / tw, 1), "steps": nw,
                                      "ctrl_l2_max": float(hip[well, -1].max()), "ctrl_l2_median": float(np.median(hip[well, -1])),
                                      "note": "the same loop on a batch made of the well-posed scenes only (cycled to 256)"}
        rep["note"] = ("oracle = reference code restated + substituted fp64 QP solver (ECOS unavailable: parity unpinned at that "
                       "boundary).  Ensemble per scene = the oracle itself on inputs moved by +-1 float32 ulp (8 members) and with the "
                       "DUNE hidden units permuted (same function, other fp32 summation order; 4 members).  well posed = ensemble "
                       "spread of the final controls <= 1e-4.  A: HIP <= 1e-4 on every well-posed scene; B: HIP inside the ensemble "
                       "spread elsewhere (reported; a 14th sample of a chaotic scene need not fall inside the hull of 13); C: HIP <= 1e-5 at "
                       "every iteration before the ensemble itself first disagrees by > 1e-5; D (one_step): on every scene and "
                       "iteration ONE oracle iteration from the HIP path's own iterate reproduces the HIP path's next iterate "
                       "(tests/parity_tools.py, DESIGN.md section 5)")
        # the kernel's own last QP, per scene: rebuilt on the host from the parameters the kernel built, fp64 solution certified
        batch0 = make_batch(cfg, rank * nfl * BATCH, BATCH)
        rep["gpu_last_qp"] = dict(gpu_last_qp_certificates(pans[0], cfg, batch0),
                                  note="KKT certificate (NNLS stationarity, complementarity, feasibility) of the kernel's fp64 "
                                       "solution of its last QP and objective gap to the oracle's solve of the same problem, "
                                       "all scenes of the batch; stat_oracle = the same certificate on the oracle's solutions")
        line["parity"] = rep
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
