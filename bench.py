#!/usr/bin/env python3
"""bench.py -- MPC plans/s of the MI355X-native PAN inner solver on BASELINE.json configs[1]:
1 GPU (per rank), batch = 256 synthetic scenes, diff robot, 1000 obstacle points, T = 10,
K = 10 PAN iterations (iter_threshold = 0 so that exactly K run), fp32 DUNE + fp64 QP.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (npa_forward_batch) over one batch of 256 scenes per
rank, inputs already resident in HBM; with N > 1 ranks every rank plans its own 256 scenes
(weak scaling, no data-path collective) and the control outputs are all-gathered over
RCCL inside the timed region.  Like any serving loop the bench keeps `--inflight` (default 5)
independent batches in flight: consecutive steps are different batches of 256 scenes whose PAN
iterations are interleaved on one stream (the latency-bound QP of one batch runs underneath the
DUNE launches of the other).  Every step still executes its full K iterations inside the timed
region; `--inflight 1` gives the strictly sequential number.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The interleaved schedule is sensitive to how HIP maps streams onto hardware queues: with the runtime's
# default of 4 the five helper streams share three queues, which measures best (2/3/4/5/6/8 queues:
# 84/100/123/114/89/87 k plans/s).  Pin the default so an inherited setting cannot change it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOAD = "diff_1k_T10_K10"
BATCH = 256
PEAK_FP32_MFMA_TFLOPS = 157.3           # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table
PEAK_F16_MFMA_TFLOPS = 2500.0           # dense fp16/bf16 MFMA, same table


def _cpu_worker(job):
    """One worker process of the CPU baseline: the oracle, single-threaded, on its share of the scenes.
    Returns (list of (scene, u, self_sensitivity), seconds spent planning).  self_sensitivity: how far the oracle's
    own control output moves when every obstacle coordinate of its input changes by one float32 ulp (a second,
    untimed run) -- where that is large the reference algorithm itself has no well-defined answer to compare with."""
    workload, scenes = job
    os.environ["OMP_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
    import numpy as _np
    import torch as _torch
    _torch.set_num_threads(1)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import CONFIGS, make_oracle
    from neupan_amd.scenes import make_scene
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        pass
    cfg = CONFIGS[workload]
    out = []
    make_oracle(cfg)                                   # imports / checkpoint load outside the timed part
    spent = 0.0
    for b in scenes:
        sc = make_scene(cfg, b)
        t0 = time.perf_counter()
        orc = make_oracle(cfg)
        s, u, d = orc.forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], sc["points"], sc["velocities"])
        spent += time.perf_counter() - t0
        # a PAN iteration that does not contract within K amplifies 1e-7 differences by a constant factor per
        # iteration: measure that on the oracle itself (not part of the timed baseline)
        pts = _np.nextafter(sc["points"], _np.float32(_np.inf))
        s2, u2, d2 = make_oracle(cfg).forward(sc["nom_s"], sc["nom_u"], sc["ref_s"], sc["ref_us"], pts, sc["velocities"])
        out.append((b, u, float(_np.linalg.norm(u - u2))))
    return out, spent


def cpu_baseline(workload, n_scenes, u_gpu, cores):
    """Oracle (CPU restatement, kind='port') timed on this box's host cores: `cores` worker processes x 1
    thread over independent scenes (the fairest CPU throughput), on the first n_scenes
    scenes of the same workload; also returns the control L2 of the GPU result against it (the parity
    half of the metric).  Rate = scenes / (slowest worker's planning time)."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    cores = max(1, min(cores, n_scenes))
    jobs = [(workload, list(range(w, n_scenes, cores))) for w in range(cores)]
    if cores == 1:
        res = [_cpu_worker(jobs[0])]
    else:
        with ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as ex:
            res = list(ex.map(_cpu_worker, jobs))
    wall = max(r[1] for r in res)
    errs, moving = [0.0] * n_scenes, [0.0] * n_scenes
    for part, _ in res:
        for b, u, mv in part:
            errs[b] = float(np.linalg.norm(u_gpu[b].astype(np.float64) - u))
            moving[b] = mv
    return n_scenes / wall, errs, moving, cores


def main():
    global BATCH
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-scenes", type=int, default=96, help="scenes timed through the CPU oracle (rank 0, N=1)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the CPU baseline (0 = min(host cores, 32))")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--schedule", choices=["pipeline", "groups"], default="groups",
                    help="groups: lockstep groups of forward calls (forward_interleaved, default); pipeline: staggered forward "
                         "calls (PanPipeline) -- measured slower with more than 3 batches in flight")
    ap.add_argument("--coalesce", type=int, default=1,
                    help="run this many in-flight batches as one forward call (larger launches); 1 = off (default)")
    ap.add_argument("--lanes", type=int, default=0, help="helper streams shared by the batches in flight (0 = one each)")
    ap.add_argument("--inflight", type=int, default=5, help="independent batches (steps) kept in flight")
    ap.add_argument("--workload", default=WORKLOAD, choices=sorted(k for k in __import__("neupan_amd.scenes", fromlist=["CONFIGS"]).CONFIGS),
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # the measured leg imports the product only; tests/helpers (and with it oracle/) is touched by cpu_baseline alone
    from neupan_amd.dist import gather_controls
    from neupan_amd.pan import PAN, PanPipeline, forward_interleaved
    from neupan_amd.robot import Robot
    from neupan_amd.scenes import CONFIGS, make_batch

    def make_gpu_pan(cfg, device):
        ck = os.path.join(ROOT, "tests", "golden", "checkpoints", f"{cfg.checkpoint}_model_5000.pth")
        if not os.path.exists(ck):              # the 8-edge stand-in: tests/golden/make_poly8_checkpoint.py
            ck = os.path.join(ROOT, "tests", "golden", "checkpoints", f"{cfg.checkpoint}_model_quick.pth")
        return PAN(cfg.T, cfg.dt, Robot(cfg.T, cfg.dt, **cfg.robot), iter_num=cfg.iter_num, dune_max_num=cfg.n_points,
                   nrmp_max_num=cfg.nrmp_max_num, iter_threshold=0.0, dune_checkpoint=ck, adjust_kwargs=dict(cfg.adjust),
                   device=device)

    cfg = CONFIGS[args.workload]
    BATCH = args.batch
    T, K, N, E = cfg.T, cfg.iter_num, cfg.n_points, 4      # E: replaced by the planner's edge count below
    nfl = max(1, args.inflight)
    pans = [make_gpu_pan(cfg, device=dev) for _ in range(nfl)]
    E = pans[0].E
    args_dev = []
    for j in range(nfl):                    # batch j of this rank: its own 256 scenes
        batch = make_batch(cfg, (rank * nfl + j) * BATCH, BATCH)
        a = [torch.from_numpy(batch[k]).to(dev) for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
        if batch.get("velocities") is not None:            # moving points (configs[3])
            a.append(torch.from_numpy(batch["velocities"]).to(dev))
        args_dev.append(a)
    torch.cuda.synchronize(dev)

    pipe = PanPipeline(pans)

    def run_steps(n):
        """n steps (= n batches of 256 scenes), `nfl` of them in flight at a time."""
        out0 = gathered = None
        if args.schedule == "pipeline":
            # continuous: the planners' forward calls are staggered, the DUNE stream never drains
            outs = pipe.run([args_dev[i % nfl] for i in range(n)], reset_state=True)
            for o in outs:
                gathered = gather_controls(o["opt_u"], dist, world, equal_shards=True)   # RCCL all-gather when world > 1
            return outs[0], gathered
        done = 0
        while done < n:                      # groups of forward calls in lockstep
            g = min(nfl, n - done)
            outs = forward_interleaved(pans[:g], args_dev[:g], reset_state=True, lanes=args.lanes or None,
                                       coalesce=args.coalesce)   # fresh planners every step
            for o in outs:
                gathered = gather_controls(o["opt_u"], dist, world, equal_shards=True)
            out0 = outs[0]
            done += g
        return out0, gathered

    run_steps(args.warmup)
    torch.cuda.synchronize(dev)
    for p in pans:
        p.profile(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out, gathered = run_steps(args.steps)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    profs = [p.profile_read() for p in pans]
    for p in pans:
        p.profile(False)
    nl = sum(q["launches"] for q in profs)
    prof = {"launches": nl,
            "dune_ms": sum(q["dune_ms"] * q["launches"] for q in profs) / max(nl, 1),
            "nrmp_ms": sum(q["nrmp_ms"] * q["launches"] for q in profs) / max(nl, 1)}
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    iters = out["iters"].cpu().numpy()
    assert (iters == K).all(), "every scene must run exactly K PAN iterations inside the timed region"
    assert gathered.shape[0] == world * BATCH

    plans = BATCH * world * args.steps
    value = plans / elapsed
    # dense flops per (point x horizon slice) = 8320 + 64 E; a launch covers
    # one sub-batch of scenes (the C API pipelines the batch in sub-batches), all T+1 slices.
    # executed slices per plan: T+1 in the first PAN iteration, T afterwards (slice 0 is invariant
    # within a forward call and is evaluated once) -- only executed flops are counted
    slices = (T + 1) + (K - 1) * T
    total_flops = args.steps * BATCH * slices * N * (8320 + 64 * E)
    flops_per_launch = total_flops / max(prof["launches"], 1)
    dune_s = prof["dune_ms"] * 1e-3
    achieved = flops_per_launch / dune_s / 1e12 if dune_s > 0 else 0.0

    # HBM bytes per dune_kernel launch from the PMC passes of tests/tools/hbm_traffic.py (separate rocprofv3
    # runs: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), committed under profiles/
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tpath) and args.workload == WORKLOAD:      # measured for this workload only
        try:
            tj = json.load(open(tpath))
            key = [k for k in tj if k.startswith("dune_kernel")][0]
            per_launch_256 = tj[key]["hbm_bytes_per_launch"]                 # measured at 256 scenes / launch
            traffic = int(per_launch_256 * (args.steps * K * BATCH / max(prof["launches"], 1)) / BATCH)
        except Exception:
            traffic = None

    km = pans[0].key_mode()
    nf16 = 8 if km["key_terms"] == 1 else 24
    line = {
        "metric": "MPC plans/sec (node), diff robot, 1k pts, T=10, K=10; ctrl L2 vs ref" if args.workload == WORKLOAD
                  else f"MPC plans/sec (node), workload {args.workload}; ctrl L2 vs ref",
        "value": round(value, 1), "unit": "plans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[1]: batch=256 synthetic scenes/GPU, diff robot, 1000 pts, "
                                "T=10, K=10 (iter_threshold=0), M=10, fp32 DUNE (MFMA) + fp64 QP") if args.workload == WORKLOAD
                               else f"{args.workload}: batch={BATCH} synthetic scenes/GPU, {cfg.kinematics} robot, {N} pts, "
                                    f"T={T}, K={K} (iter_threshold=0), M={cfg.nrmp_max_num}, fp32 DUNE (MFMA) + fp64 QP",
                   "scenes_per_gpu": BATCH, "points": N, "T": T, "K": K, "M": cfg.nrmp_max_num,
                   "batches_in_flight": nfl, "coalesced_per_launch": args.coalesce,
                   "parallelism": f"scene-shard x{world}, RCCL all-gather of controls"},
        "roofline": {"bound": "mfma", "kernel": f"dune_kernel<{E},{km['key_terms']}>",
                     "note": ("ALGORITHMIC fp32 flops per launch / launch time against the fp32-input MFMA peak (the arithmetic "
                              "the path is specified in).  The kernel computes distance KEYS with the four 32x32 layers as "
                              + ("single fp16 products (2 v_mfma_f32_32x32x16_f16 per layer)" if km["key_terms"] == 1 else
                                 "fp16x2 split products (6 v_mfma_f32_32x32x16_f16 per layer)") +
                              "; the keys only nominate candidates (margin = 6 x the key error measured for the checkpoint at "
                              "npa_create), the emitted rows are re-encoded with the exact fp32 MFMA, so frac can exceed 1; "
                              "`executed` prices the MFMA flops it really issues against the fp16 peak.  The binding unit is "
                              "the VALU (LayerNorm/tanh/conversions), see DESIGN.md")
                             if km["key_terms"] != 0 else "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                     "key_mode": km,
                     "achieved": round(achieved, 3),
                     "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                     "traffic": traffic, "flops_per_launch": int(flops_per_launch),
                     "executed": ({"mfma": f"v_mfma_f32_32x32x16_f16 x{nf16} + v_mfma_f32_32x32x2_f32 x1 per tile",
                                   "tflops": round(achieved * (nf16 * 32768 + 4096) / (32 * (8320 + 64 * E)), 2),
                                   "peak": PEAK_F16_MFMA_TFLOPS,
                                   "frac": round(achieved * (nf16 * 32768 + 4096) / (32 * (8320 + 64 * E)) / PEAK_F16_MFMA_TFLOPS, 4)}
                                  if km["key_terms"] != 0 else None),
                     "launch_ms": round(prof["dune_ms"], 4), "launches_timed": prof["launches"],
                     "nrmp_qp_launch_ms": round(prof["nrmp_ms"], 4)},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        u_gpu = out["opt_u"].cpu().numpy()
        ncore = args.cpu_cores if args.cpu_cores > 0 else min(os.cpu_count() or 1, 32)
        cpu_rate, errs, moving, ncore = cpu_baseline(args.workload, args.cpu_scenes, u_gpu, ncore)
        errs, moving = np.array(errs), np.array(moving)
        conv = moving <= 1e-4                       # oracle output stable under a 1-ulp change of its input
        worst = np.argsort(-errs)[:5]
        line["cpu_baseline"] = {"value": round(cpu_rate, 3), "unit": "plans/s", "cores": ncore, "kind": "port",
                                "sample": f"first {args.cpu_scenes} scenes of the same workload, K={K} each, "
                                          f"oracle/pan_oracle.py (numpy fp32 + fp64 IPM), {ncore} worker processes x 1 thread "
                                          f"(host has {os.cpu_count()} cores)"}
        line["parity"] = {"ctrl_l2_vs_oracle_median": float(np.median(errs)), "max": float(errs.max()),
                          "frac_le_1e-4": float((errs <= 1e-4).mean()), "scenes": int(len(errs)),
                          "scenes_well_posed": int(conv.sum()),
                          "max_over_well_posed": float(errs[conv].max()) if conv.any() else None,
                          "worst_scenes": [{"scene": int(b), "ctrl_l2": float(errs[b]), "oracle_self_sensitivity": float(moving[b])}
                                           for b in worst],
                          "note": "oracle = reference code restated + substituted fp64 QP solver (ECOS unavailable).  well posed = "
                                  "the oracle's own control output moves <= 1e-4 when every obstacle coordinate of its input "
                                  "changes by one float32 ulp (second, untimed oracle run per scene); elsewhere the PAN "
                                  "iteration does not contract within K and the reference algorithm has no answer that is "
                                  "stable to rounding (DESIGN.md section 5)"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
