#!/usr/bin/env python3
"""bench.py -- MPC plans/s of the MI355X-native PAN inner solver on BASELINE.json configs[1]:
1 GPU (per rank), batch = 256 synthetic scenes, diff robot, 1000 obstacle points, T = 10,
K = 10 PAN iterations (iter_threshold = 0 so that exactly K run), fp32 DUNE + fp64 QP.

    python bench.py [--gpus N --steps K --warmup W]          # (N > 1 without a launcher: re-executes itself as below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path (one forward call: K x {selection, QP}) over one batch of 256 scenes per
rank, inputs already resident in HBM; with N > 1 ranks every rank plans its own 256 scenes (weak scaling, no
data-path collective) and the control outputs are all-gathered over RCCL inside the timed region.  Like any
serving loop the bench keeps `--inflight` independent batches in flight, grouped into `--chains C` launch chains
(default: five batches per chain): the steps of a chain are issued as ONE library call (npa_forward_batch_group) on ONE
stream and every stage of theirs runs as one merged launch over all their scenes (blockIdx.y = the step).
Every step still executes its full K iterations inside the timed region, on its own 256 scenes, its own
buffers and its own planner state; `--chains 0` gives one stream and one launch chain per step (round 4's
schedule), `--inflight 1` the strictly sequential number.

OUTPUT: the LAST stdout line is ONE compact JSON record (<= 6 KB: the contract fields, `roofline`,
`cpu_baseline`, the parity verdicts and one figure per extra leg); the full record (per-scene listings, PMC
dump, CPU sweep, notes) goes to bench_full.json next to this file (NPA_BENCH_FULL=path moves it).

Besides the headline the record carries (rank 0, one GPU, unless --no-extras):
  extra.paths          the SAME loop with the un-pruned / fallback selections: exact fp32 network keys over every point
                       (NPA_DUNE_FP32KEYS=1: SURVEY 8(d)'s literal "encoder on every obstacle point x horizon step"), network
                       keys with single / split fp16 products (NPA_KEY_TERMS=1 / 3), each with the encoder's executed MFMA
                       rate against its peak (per launch and aggregated over the region);
  extra.uniform_cloud  SURVEY 8(d)'s uniform cloud (workload uniform_1k_T10_K10): throughput, candidates per slice, parity;
  extra.other_configs  BASELINE configs[2], [3], [4] (acker 2k T=20 K=15; 4000 moving points, 1024 scenes per step; 8-edge
                       hull 5000 points in exact fp32 AND in the labelled bf16 tier): throughput + a 16-scene parity verdict;
  extra.early_exit     the default loop with the reference's default iter_threshold = 0.1 (pan.py:243): plans/s and the
                       executed PAN iterations (mean / max), SURVEY 8(d)'s second run;
  extra.h2d_inclusive  the default loop with every step's inputs uploaded from pinned host memory on the step's own stream
                       (the reference re-uploads the cloud every step, neupan.py:123-127): the PCIe-inclusive rate -- never `value`.
"""
import argparse
import contextlib
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# One HIP stream per batch in flight; every stream should own a hardware queue, or two batches serialise behind each
# other.  The runtime's default is 4 queues; 32 covers the 20 batches in flight plus torch's own streams (sweep in
# with fewer queues than streams the throughput falls to what that many concurrent chains give).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import re  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOAD = "diff_1k_T10_K10"
BATCH = 256
BURST_DEFAULT = 1
CHAIN_THREADS_DEFAULT = -1  # issuing threads of a chained loop: -1 one per chain, 0 the calling thread alone (one group call per round)
CHAINS_DEFAULT = -1        # -1: batches in flight / 5 (four chains of five steps at 20 in flight, eight at 40)
# /opt/skills/guides/MI355X_MICROARCH.md, chip-level table (256 CUs x 4 SIMDs, 2.4 GHz)
PEAK_FP64_VALU_TFLOPS = 78.6
PEAK_FP32_MFMA_TFLOPS = 157.3       # v_mfma_f32_32x32x2_f32: what the exact encoder runs on
PEAK_F16_MFMA_TFLOPS = 2500.0       # dense fp16 / bf16 MFMA
N_SIMD = 1024
PMC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json")]     # newest first
# which source files a kernel's counters depend on (a record is "current" only while these are unchanged)
_QP_SRC = ("nrmp_qp.hip", "nrmp_qp_device.h", "nrmp_qp_body.inc", "aset_reduce.h", "pan_common.h")
_DUNE_SRC = ("dune.hip", "dune_device.h", "select_geo_carve.inc", "select_geo_body.inc", "pan_common.h")
KERNEL_SOURCES = {"nrmp_qp_kernel": _QP_SRC, "nrmp_qp_group_kernel": _QP_SRC, "select_geo_kernel": _DUNE_SRC,
                  "select_geo_group_kernel": _DUNE_SRC, "select_kernel": _DUNE_SRC, "dune_kernel": _DUNE_SRC,
                  "stage_kernel": ("c_api.hip", "pan_common.h"), "stage_group_kernel": ("c_api.hip", "pan_common.h")}


def source_hash(files=None):
    """sha256 over kernel sources (all of neupan_amd/csrc, or the named files): ties tracked counters to the code they
    were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "neupan_amd", "csrc")
    for f in sorted(os.listdir(d)) if files is None else files:
        if f.endswith((".hip", ".h", ".inc")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_hash(kernel):
    return source_hash(KERNEL_SOURCES.get(kernel, None))


_ISA = {}


def kernel_isa_hash(full_name):
    """Fingerprint of the MACHINE CODE of one kernel instantiation of the built library (tests/tools/kernel_resources.py:
    the bytes of its function symbol + its kernel descriptor), looked up by the demangled name a rocprofv3 trace prints for it
    ("nrmp_qp_kernel<10, 10, false, false, 2, false>").  A counter record stays valid exactly as long as this does -- source
    edits that compile to the same code (comments, another kernel in the file) keep it, another compiler or register
    allocation does not.  None when the ROCm LLVM tools are not installed (the source hash then decides)."""
    if "table" not in _ISA:
        _ISA["table"] = None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
            import kernel_resources as kr
            if kr.tools_available():
                _ISA["table"] = kr.kernel_isa_hashes()
        except Exception:
            _ISA["table"] = None
    t = _ISA["table"]
    m = re.match(r"^(?:void )?(\w+)(?:<(.*)>)?$", full_name.strip())
    if t is None or not m:
        return None
    # Itanium mangling of a kernel template whose arguments are ints and bools: _Z<len><name>I(Li<n>E|Lb<0/1>E)*E...
    name, targs = m.group(1), m.group(2)
    prefix = f"_Z{len(name)}{name}"
    if targs is not None:
        enc = []
        for a in (x.strip() for x in targs.split(",")):
            if a in ("true", "false"):
                enc.append("Lb%dE" % (a == "true"))
            elif re.fullmatch(r"-?\d+", a):
                enc.append("Li%sE" % a.replace("-", "n"))
            else:
                return None
        prefix += "I" + "".join(enc) + "E"
    hits = [h for n, h in t.items() if n.startswith(prefix) and (targs is not None or not n[len(prefix):].startswith("I"))]
    return hits[0] if len(hits) == 1 else None


def record_is_current(short, rec):
    """Was this kernel's counter record (profiles/*_pmc.json) measured on the code the tree builds?  By machine code when the
    record carries an isa_hash and the tools are here; by the hash of the kernel's source files otherwise."""
    isa = kernel_isa_hash(rec.get("kernel", "")) if rec.get("isa_hash") else None
    if isa is not None:
        return isa == rec["isa_hash"]
    return rec.get("source_hash") == kernel_hash(short)


def load_pmc(workload):
    """The newest tracked PMC record of this workload (tests/tools/pmc_collect.py), or None."""
    for f in PMC_FILES:
        if os.path.exists(f):
            try:
                pj = json.load(open(f))
            except Exception:
                continue
            if pj.get("workload") == workload:
                pj["_file"] = os.path.relpath(f, ROOT)
                return pj
    return None


@contextlib.contextmanager
def environ(env):
    """os.environ updated for the duration (the library reads its knobs when a handle is created)."""
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def make_gpu_pan(cfg, device, **over):
    from neupan_amd.pan import PAN
    from neupan_amd.robot import Robot
    ck = os.path.join(ROOT, "tests", "golden", "checkpoints", f"{cfg.checkpoint}_model_5000.pth")
    if not os.path.exists(ck):              # the 8-edge stand-in: tests/golden/make_poly8_checkpoint.py
        ck = os.path.join(ROOT, "tests", "golden", "checkpoints", f"{cfg.checkpoint}_model_quick.pth")
    kw = dict(iter_num=cfg.iter_num, dune_max_num=cfg.n_points, nrmp_max_num=cfg.nrmp_max_num, iter_threshold=0.0,
              dune_checkpoint=ck, adjust_kwargs=dict(cfg.adjust), device=device)
    kw.update(over)
    return PAN(cfg.T, cfg.dt, Robot(cfg.T, cfg.dt, **cfg.robot), **kw)


class Loop:
    """`nfl` planners of one workload, one stream and one prepared step each (PAN.make_step: arguments validated once, ONE
    library call per step, output tensors reused; every step starts from a cleared stop-criterion / warm-start state, reset
    inside the staging launch, so that all steps do the same work), driven by neupan_amd.serve.StepLoop; the controls of
    every step are all-gathered over RCCL from ONE communication stream when a process group exists.  Planners that carry
    timing events: every 4th (the events ride on the dispatches: recording two per launch doubles the host's time to enqueue
    a step)."""

    STREAMS = {}           # device -> the process's chain streams, created once (every Loop of the process reuses them: a second
                           # set of 20 live streams would push the process past the ~24 hardware queues the device schedules
                           # without time-slicing)

    CHAIN_THREADS = -1     # issuing threads of a chained loop (--chain-threads; see __init__)
    BURST = False          # StepLoop(burst=...): set once from the command line, every Loop of the process follows it
    CHAINS = 0             # > 0: the batches in flight form this many launch chains -- the steps of a chain share ONE stream, are
                           # issued as one group call and run every stage as one merged launch (npa_forward_batch_group)

    def __init__(self, workload, batch, nfl, dev, rank=0, world=1, dist=None, env=None, graph=False, issue_threads=4,
                 scene_index=None, chains=None, over=None, h2d=False):
        from neupan_amd.scenes import CONFIGS, make_batch
        from neupan_amd.serve import ControlGatherer, StepLoop
        self.cfg = cfg = CONFIGS[workload]
        self.workload, self.batch, self.nfl, self.dev, self.world = workload, batch, nfl, dev, world
        self.early = bool(over) and over.get("iter_threshold", 0.0) > 0
        with environ(env):
            self.pans = [make_gpu_pan(cfg, device=dev, **(over or {})) for _ in range(nfl)]
        chains = Loop.CHAINS if chains is None else chains
        self.chains = chains = min(chains, nfl) if (chains > 0 and Loop.BURST and not graph) else 0
        user_threads = issue_threads
        if chains:
            # Loop.CHAIN_THREADS < 0: one issuing thread per chain (slot j belongs to thread j % threads); >= 0: that many threads
            # (0: the calling thread issues every chain of a round in ONE breadth-first group call)
            issue_threads = chains if Loop.CHAIN_THREADS < 0 else min(Loop.CHAIN_THREADS, chains)
        pool = Loop.STREAMS.setdefault(str(dev), [])
        while len(pool) < (chains or nfl):
            pool.append(torch.cuda.Stream(device=dev))
        self.streams = [pool[j % chains] for j in range(nfl)] if chains else pool[:nfl]
        self.args = []
        for j in range(nfl):                    # batch j of this rank: its own scenes
            b = make_batch(cfg, (rank * nfl + j) * batch, batch)
            a = [torch.from_numpy(b[k]).to(dev) for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
            a.append(torch.from_numpy(b["velocities"]).to(dev) if b.get("velocities") is not None else None)
            if scene_index is not None:         # (a batch made of selected scenes of batch 0, cycled)
                if j == 0:
                    self._sel = [t.index_select(0, scene_index).contiguous() if t is not None else None for t in a]
                a = self._sel
            if h2d:
                # the step's inputs live in ONE device buffer (views, 256-byte aligned) mirrored by ONE pinned host buffer: a
                # host caller ships them with a single copy per step
                offs, tot = [], 0
                for t in a:
                    offs.append(tot)
                    tot += 0 if t is None else (t.numel() * 4 + 255) // 256 * 256
                blob = torch.empty(tot, dtype=torch.uint8, device=dev)
                views = []
                for t, o in zip(a, offs):
                    if t is None:
                        views.append(None); continue
                    v = blob[o:o + t.numel() * 4].view(torch.float32).view(t.shape)
                    v.copy_(t)
                    views.append(v)
                a = views
                self._blobs = getattr(self, "_blobs", []) + [blob]
            self.args.append(a)
        torch.cuda.synchronize(dev)
        self.cur = torch.cuda.current_stream(dev)
        self.timed_idx = set(range(0, nfl, 4)) if nfl >= 4 else set(range(nfl))
        self.gatherer = ControlGatherer(dist, world, device=dev, slots=nfl, shape=(batch, 2, cfg.T))
        self.steps = []
        for j in range(nfl):
            with torch.cuda.stream(self.streams[j]):
                self.steps.append(self.pans[j].make_step(*self.args[j], reset_every_step=True,
                                                          graph=(graph and j not in self.timed_idx)))
        torch.cuda.synchronize(dev)
        if chains and nfl >= 2:
            # the library decides whether the steps of a chain can share launches (geometric keys, a register-resident QP,
            # one configuration): if not -- network-key paths, other polygon sizes -- every step keeps a stream of its own
            from neupan_amd.pan import StepGroup
            mine = [j for j in range(nfl) if j % chains == 0]
            if len(mine) < 2 or not StepGroup([self.steps[j] for j in mine], [self.streams[j] for j in mine]).merged():
                self.chains = chains = 0
                while len(pool) < nfl:
                    pool.append(torch.cuda.Stream(device=dev))
                self.streams = pool[:nfl]
                issue_threads = user_threads
        self.h2d_bytes = 0
        if h2d:
            # every step's inputs come from pinned host memory, uploaded on the step's own stream in front of the step (what
            # a host caller of the reference does per control cycle, neupan.py:123-127); the device tensors the step reads
            # are the ones it captured
            self.host = [b.cpu().pin_memory() for b in self._blobs]
            for j in range(nfl):
                self.steps[j].pre_issue = (lambda d, h_: (lambda: d.copy_(h_, non_blocking=True)))(self._blobs[j], self.host[j])
            self.h2d_bytes = int(self.host[0].numel())
        self.loop = StepLoop(self.steps, self.streams, self.gatherer, self.cur, threads=issue_threads, burst=Loop.BURST)

    def run(self, n):
        return self.loop.run(n)

    def timed(self, steps, warmup, barrier=None):
        """warmup untimed steps, then EXACTLY `steps` steps between (barrier +) synchronize on both sides.
        Returns dict(elapsed, t_issue, prof, last)."""
        dev = self.dev
        self.run(warmup)
        torch.cuda.synchronize(dev)
        tp = [self.pans[j] for j in sorted(self.timed_idx)]
        for p in tp:
            p.profile(True)
        if barrier:
            # (the barrier is a collective of its own kind -- an all-reduce -- on the communicator: two untimed ones first, so that
            # nothing it sets up on first use is left for the closing barrier inside the region; profiles/r06_region_trace.txt:
            # three closing barriers of ~60 took 2 - 11 ms instead of 0.06)
            barrier()
            barrier()
            barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        last = self.run(steps)
        t_issue = time.perf_counter() - t0          # host time to enqueue every step (the GPU is still working)
        torch.cuda.synchronize(dev)
        t_drain = time.perf_counter() - t0          # ... until the device has finished them (and the gathers behind them)
        if barrier:
            barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        profs = [p.profile_read() for p in tp]
        for p in tp:
            p.profile(False)
        nl = sum(q["launches"] for q in profs)
        avg = lambda key: sum(q[key] * q["launches"] for q in profs) / max(nl, 1)
        K = self.cfg.iter_num
        its = []
        for o, g in (x for x in last if x is not None):
            it = o["iters"].cpu().numpy()
            its.append(it)
            assert self.early or (it == K).all(), "every scene must run exactly K PAN iterations inside the timed region"
            assert g.numel() == self.world * self.batch * 2 * self.cfg.T
        self.last_iters = np.concatenate(its) if its else None
        na = sum(q.get("aset_launches", 0) for q in profs)
        return dict(elapsed=elapsed, t_issue=t_issue, t_drain=t_drain, last=last,
                    prof={"launches": nl, "dune_ms": avg("dune_ms"), "select_ms": avg("select_ms"), "nrmp_ms": avg("nrmp_ms"),
                          "aset_launches": na,
                          "aset_ms": sum(q.get("aset_ms", 0.0) * q.get("aset_launches", 0) for q in profs) / max(na, 1)})

    def audit(self):
        audits = [p.audit() for p in self.pans]
        a = {k: sum(x[k] for x in audits) for k in ("tiles", "points", "violations")}
        a["worst_excess"] = max(x["worst_excess"] for x in audits)
        return a

    def qp_iterations(self):
        """Interior-point iterations the QP launches of ONE step of batch 0 execute, measured on the device (qp_info[14] of
        every scene after every PAN iteration, untimed): list over k of (sum over scenes, max over scenes)."""
        pan, a = self.pans[0], self.args[0]
        pan.forward_begin(*a, reset_state=True)
        out = []
        for k in range(pan.iter_num):
            pan.forward_iter(k)
            it = pan.last_qp_info()[:, 14]
            out.append((float(it.sum()), float(it.max())))
        pan.forward_end()
        return out

    def close(self):
        self.loop.close()
        self.steps = self.pans = self.args = None
        torch.cuda.empty_cache()


def dune_mfma_rate(km, cfg, batch, dune_ms, steps=None, elapsed=None):
    """Executed matrix work of the encoder per second against the peak of the MFMA it runs on (SURVEY 8(d)'s roofline, priced
    on what is EXECUTED).  Network keys: dune_kernel encodes every point of every slice -- per 32-point tile four 32x32x32
    layers (262 144 flop; 3x with split products) on fp16 MFMA, or on fp32 MFMA with NPA_DUNE_FP32KEYS.  `tflops` / `frac`:
    over the duration of ONE launch while the other chains' kernels co-run; `aggregate_*`: every encoder launch of the timed
    region over its wall time (steps x K launches / elapsed) -- the chip's rate."""
    T, K, N = cfg.T, cfg.iter_num, cfg.n_points
    if km["key_terms"] == 4 or dune_ms <= 0:
        return None
    slices = ((T + 1) + (K - 1) * T) / K                     # slices a launch encodes (slice 0 only in the first iteration)
    tiles = batch * slices * ((N + 31) // 32)
    per_tile = {0: 262144 + 4096, 1: 262144 + 4096, 3: 3 * 262144 + 4096}[km["key_terms"]]
    peak = PEAK_FP32_MFMA_TFLOPS if km["key_terms"] == 0 else PEAK_F16_MFMA_TFLOPS
    ex = tiles * per_tile / (dune_ms * 1e-3) / 1e12
    out = {"kernel": "dune_kernel", "launch_ms": round(dune_ms, 4), "tflops": round(ex, 2), "peak": peak,
           "frac": round(ex / peak, 4), "points_encoded_per_launch": int(tiles * 32),
           "mfma": "v_mfma_f32_32x32x2_f32" if km["key_terms"] == 0 else "v_mfma_f32_32x32x16_f16"}
    if steps and elapsed:
        agg = tiles * per_tile * K * steps / elapsed / 1e12
        out["aggregate_tflops"], out["aggregate_frac"] = round(agg, 2), round(agg / peak, 4)
    return out


def short_run(workload, batch, nfl, dev, steps, warmup, env=None, issue_threads=4, **loop_kw):
    """One more timed loop (GPU only) on another workload / another path of the library."""
    lp = Loop(workload, batch, nfl, dev, env=env, issue_threads=issue_threads, **loop_kw)
    r = lp.timed(steps, warmup)
    km = lp.pans[0].key_mode()
    out = {"plans_per_s": round(batch * steps / r["elapsed"], 1), "ms_per_step": round(1e3 * r["elapsed"] / steps, 4),
           "steps": steps, "scenes_per_step": batch, "batches_in_flight": nfl, "chains": lp.chains, "key_terms": km["key_terms"],
           "select_launch_ms": round(r["prof"]["select_ms"], 4), "qp_launch_ms": round(r["prof"]["nrmp_ms"], 4),
           "dune_launch_ms": round(r["prof"]["dune_ms"], 4)}
    mf = dune_mfma_rate(km, lp.cfg, batch, r["prof"]["dune_ms"], steps, r["elapsed"])
    if mf:
        out["dune_executed_mfma"] = mf
    out["margin_violations"] = lp.audit()["violations"]
    return out, lp


_ENSEMBLES = {}


def parity_leg(lp, scenes, cores, n_ulp=8, n_perm=4, sweep=False, explain=True):
    """The ensemble verdicts A / C / D of tests/parity_tools.py for the first `scenes` scenes of the loop's batch 0 (CPU oracle;
    rank 0 only, untimed).  Returns (report, cpu rate, workers, hip deviations, spreads, trace).  The oracle's ensemble of a
    workload is computed once per process (two tiers of the same workload are judged against the same runs)."""
    from parity_tools import judge, one_step_consistency, one_step_report, run_ensemble
    key = (lp.workload, scenes, n_ulp, n_perm)
    if key not in _ENSEMBLES:
        _ENSEMBLES[key] = run_ensemble(lp.workload, range(scenes), cores, n_ulp=n_ulp, n_perm=n_perm, sweep=sweep)
    base, members, cpu_rate, ncore = _ENSEMBLES[key]
    pan = lp.pans[0]
    pan.reset_stop_state()
    tr = pan.forward_batch_trace(*lp.args[0])
    trace_u = tr["trace_u"].cpu().numpy()
    rep, hip, sp = judge(trace_u[:scenes], base, members)
    tp = tr["trace_pts"].cpu().numpy()[:scenes] if tr.get("trace_pts") is not None else None
    if explain:
        dev, why = one_step_consistency(lp.workload, range(scenes), tr["trace_s"].cpu().numpy()[:scenes], trace_u[:scenes], cores,
                                        explain=True, trace_pts=tp, trace_merit=tr["trace_qp_info"].cpu().numpy()[:scenes, :, 1],
                                        trace_rows=None if tr.get("trace_mu") is None else (tr["trace_mu"].cpu().numpy()[:scenes],
                                                                                            tr["trace_lam"].cpu().numpy()[:scenes]))
        rep["one_step"] = one_step_report(dev, why=why)
    else:
        rep["one_step"] = one_step_report(one_step_consistency(lp.workload, range(scenes), tr["trace_s"].cpu().numpy()[:scenes],
                                                               trace_u[:scenes], cores))
    rep["well_posed_frac"] = round(float((sp[:, -1] <= 1e-4).mean()), 4)
    return rep, cpu_rate, ncore, hip, sp, tr


def fleet_cycle_leg(dev, with_cpu=True, robots=256):
    """SURVEY.md 8(f) rows 1-2 in their place: ONE closed-loop control cycle of `robots` robots on the device, in the order of
    neupan.forward (neupan/neupan.py:104-167): scan -> points (neupan.py:173-222), path progress + nominal / reference rollout
    (initial_path.py:68-126, 247-277), the PAN loop, warm start, stop test, action (neupan_amd.fleet.FleetPlanner).  Two settings:
    `shipped` = every shipped planner.yaml (K = 2, dune_max_num = 100, iter_threshold = 0.1, a 100-beam lidar) and `k10` = the
    headline's (K = 10, 1000 beams / points, iter_threshold = 0).  Per setting: robot-cycles/s, and the front end's share of the
    cycle = (scan -> points + progress + rollout alone) / (whole cycle).  CPU beside it: oracle/frontend_oracle.py (the reference's
    algorithm, one thread) per robot for the two front-end steps."""
    import torch
    from neupan_amd.fleet import FleetPlanner
    from neupan_amd.robot import Robot
    from neupan_amd.scenes import CONFIGS
    cfg = CONFIGS[WORKLOAD]
    B = robots
    rng = np.random.default_rng(11)
    robot = Robot(cfg.T, cfg.dt, **cfg.robot)
    ck = os.path.join(ROOT, "tests", "golden", "checkpoints", f"{cfg.checkpoint}_model_5000.pth")
    paths = [[np.array([[i * 0.4], [0.3 * (b % 5) - 0.6], [0.0], [1.0]]) for i in range(80)] for b in range(B)]
    poses = np.column_stack([rng.uniform(0, 2, B), rng.uniform(-0.6, 0.6, B), rng.uniform(-0.1, 0.1, B)])

    def gpu_us(fn, reps):
        fn(); fn(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps * 1e6

    out = {"robots": B}
    for tag, K, beams, thr in (("shipped", 2, 100, 0.1), ("k10", cfg.iter_num, cfg.n_points, 0.0)):
        fleet = FleetPlanner(robot, cfg.T, cfg.dt, cfg.ref_speed, device=dev, dune_checkpoint=ck, iter_num=K, dune_max_num=beams,
                             nrmp_max_num=cfg.nrmp_max_num, iter_threshold=thr, adjust_kwargs=dict(cfg.adjust))
        fleet.set_paths(paths)
        # a corridor seen by the lidar: walls 3 - 4.5 m to either side, a fifth of the beams out of range
        ang = np.linspace(-np.pi, np.pi, beams)
        ranges = np.clip(rng.uniform(3.0, 4.5, (B, 1)) / np.maximum(np.abs(np.sin(ang))[None], 0.3), 0.2, 9.5)
        ranges[rng.random((B, beams)) < 0.2] = 10.0
        r_d = torch.from_numpy(ranges).to(dev)

        def front():
            pts, npts = fleet.scan_to_point(poses, r_d, -np.pi, np.pi, 0.1, 10.0, max_points=beams)
            fleet.nb.progress(poses, fleet.close_threshold, fleet.ind_range, fleet.arrive_threshold, fleet.arrive_index_threshold)
            fleet.nb.generate_nom_ref_state(poses, fleet.cur_vel, fleet.ref_speed)
            return pts, npts

        def cycle():
            pts, npts = fleet.scan_to_point(poses, r_d, -np.pi, np.pi, 0.1, 10.0, max_points=beams)
            fleet.forward(poses, pts, None, npts)

        cyc = gpu_us(cycle, 30)
        fr = gpu_us(front, 30)
        out[tag] = {"K": K, "beams": beams, "iter_threshold": thr, "us_per_cycle": round(cyc, 1),
                    "robot_cycles_per_s": round(B / cyc * 1e6), "front_end_us": round(fr, 1), "front_end_share": round(fr / cyc, 3)}
        if with_cpu:
            from oracle import frontend_oracle as fo
            curves = [np.array([p.reshape(4) for p in paths[b]]) for b in range(8)]
            vel = np.zeros((2, cfg.T), np.float32)
            t0 = time.perf_counter()
            for b in range(8):
                fo.scan_to_point(poses[b], ranges[b], -np.pi, np.pi, 0.1, 10.0)
                fo.generate_nom_ref_state(curves[b], 0, 0.4, poses[b], vel, cfg.ref_speed, cfg.T, cfg.dt, "diff", 0.0)
            out[tag]["cpu_front_end_us_per_robot"] = round((time.perf_counter() - t0) / 8 * 1e6, 1)
        del fleet
    out["note"] = ("one closed-loop cycle of `robots` robots on the device (FleetPlanner: scan -> points, progress, rollout, PAN, warm "
                   "start, stop test, action; one host read of the arrival flags per cycle, wall clock incl. the Python wrapper); "
                   "front_end_share = the front-end steps alone / the whole cycle; cpu_front_end: oracle/frontend_oracle.py, 1 thread")
    return out


def slim(rep):
    """The verdicts of a parity report without the per-scene listings (the other configurations' entries of the line)."""
    keep = ("scenes", "ensemble_members", "ctrl_l2_vs_oracle_median", "max", "frac_le_1e-4", "scenes_well_posed",
            "max_over_well_posed", "A_well_posed_all_le_tol", "C_le_1e-5_until_ensemble_diverges", "max_hip_before_divergence",
            "well_posed_frac")
    out = {k: rep[k] for k in keep if k in rep}
    os_ = rep.get("one_step", {})
    out["one_step"] = {k: os_[k] for k in ("steps_checked", "max", "p99", "median", "frac_le_tol", "unexplained", "stalled", "explained_by",
                                           "largest_explained") if k in os_}
    if os_.get("above_tol"):
        out["one_step"]["worst_above_tol"] = os_["above_tol"][:2]
        out["one_step"]["unexplained_or_stalled"] = [w for w in os_["above_tol"] if w["explained"] is None or w.get("stalled")][:6]
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: the same command again under torch.distributed.run, one rank
    per GPU (what the round driver does itself for N > 1; a bare call should not die in argument handling)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def dry_run(args):
    """--dry-run: the launch / rendezvous / timing / printing protocol of the bench without a GPU (gloo, a numpy stand-in for
    the step): barrier + K steps + barrier, MAX over the ranks, ONE line from rank 0.  tests/test_bench_contract.py drives it
    with two ranks through self_launch -- the N > 1 path the driver's SCALE run takes."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    a = np.ones((64, 64), dtype=np.float32)

    def step():
        return float((a @ a).sum())
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    if rank == 0:
        print(json.dumps({"metric": "MPC plans/sec (node), diff robot, 1k pts, T=10, K=10; ctrl L2 vs ref", "value": round(args.batch * world * args.steps / elapsed, 1),
                          "unit": "plans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "DRY RUN: no GPU work, the launch and timing protocol only"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


COMPACT_LIMIT = 6144


def _verdicts(par):
    """A / C / D of a parity report as scalars."""
    if not par:
        return None
    os_ = par.get("one_step", {})
    return {"scenes": par.get("scenes"), "members": par.get("ensemble_members"), "median": _r(par.get("ctrl_l2_vs_oracle_median")),
            "well_posed": par.get("scenes_well_posed"), "max_well_posed": _r(par.get("max_over_well_posed")),
            "A": par.get("A_well_posed_all_le_tol"), "C": par.get("C_le_1e-5_until_ensemble_diverges"),
            "D_frac_le_tol": os_.get("frac_le_tol"), "D_unexplained": os_.get("unexplained"), "D_stalled": os_.get("stalled"),
            "D_max": _r(os_.get("max"))}


def _r(x, n=4):
    if x is None or isinstance(x, (bool, int, str)):
        return x
    try:
        return float(f"{float(x):.{n}g}")
    except Exception:
        return x


def compact(line, full_path):
    """The record the driver parses: <= COMPACT_LIMIT bytes, every contract key, scalars only below them."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    out = {k: line[k] for k in keep}
    c = line["config"]
    out["config"] = {k: c[k] for k in ("workload", "scenes_per_gpu", "points", "T", "K", "M", "batches_in_flight", "chains",
                                       "issue_threads", "parallelism") if k in c}
    out["config"]["scenes_per_launch"] = line["roofline"].get("scenes_per_launch")
    r = line["roofline"]
    ro = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "scenes_per_launch",
                                "select_launch_ms", "launches_timed", "ipm_iterations_per_launch", "flops_per_launch", "frac_alone",
                                "valu_issue_frac_alone", "lds_bank_conflict_frac", "algorithmic_bytes_per_launch")}
    if r.get("chip_aggregate"):
        ro["chip_aggregate"] = r["chip_aggregate"]
    if r.get("vector_pipe"):
        ro["vector_pipe"] = {k: r["vector_pipe"][k] for k in ("simd_cycles_per_plan", "bound_plans_per_s", "frac")}
    if r.get("pmc"):
        pk = r["pmc"].get("per_kernel", {})
        ro["pmc"] = {"file": r["pmc"].get("file"), "current": {k: v.get("current") for k, v in pk.items()}}
    if r.get("select"):
        ro["select"] = {k: r["select"].get(k) for k in ("kernel", "launch_ms", "mfma_tflops", "mfma_peak", "mfma_frac", "traffic",
                                                        "algorithmic_bytes_per_launch")}
    out["roofline"] = ro
    if "cpu_baseline" in line:
        cb = line["cpu_baseline"]
        out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": cb.get("sample_short", cb["sample"][:160]), "cgroup_cpu_quota": cb.get("cgroup_cpu_quota")}
    if "parity" in line:
        p = line["parity"]
        v = _verdicts(p)
        v["frac_le_1e-4"] = p.get("frac_le_1e-4")
        v["max"] = _r(p.get("max"))
        g = p.get("gpu_last_qp") or {}
        v["gpu_last_qp"] = {k: _r(g[k]) for k in list(g)[:6] if not isinstance(g[k], (dict, list, str))}
        if p.get("well_posed_only"):
            v["well_posed_only_plans_per_s"] = p["well_posed_only"].get("plans_per_s")
        out["parity"] = v
    for k in ("latency_B1_ms",):
        if k in line:
            out[k] = {kk: vv for kk, vv in line[k].items() if kk != "note"}
    out["host_issue_ms_per_step"] = line.get("host_issue_ms_per_step")
    out["region_ms"] = line.get("region_ms")
    if "margin_audit" in line:
        out["margin_audit"] = {k: line["margin_audit"][k] for k in ("points", "violations")}
    x = line.get("extra")
    if x:
        e = {}
        if "paths" in x:
            e["paths"] = {k: {"plans_per_s": v["plans_per_s"], "same_controls": v.get("controls_equal_default_path"),
                              "mfma_tflops_aggregate": (v.get("dune_executed_mfma") or {}).get("aggregate_tflops"),
                              "mfma_frac_aggregate": (v.get("dune_executed_mfma") or {}).get("aggregate_frac")}
                          for k, v in x["paths"].items() if isinstance(v, dict)}
        if "uniform_cloud" in x:
            u = x["uniform_cloud"]
            e["uniform_cloud"] = {"plans_per_s": u["plans_per_s"], "parity": _verdicts(u.get("parity"))}
        if "other_configs" in x:
            e["other_configs"] = {k: dict({"plans_per_s": v["plans_per_s"], "parity": _verdicts(v.get("parity"))},
                                          **({"same_controls_as_exact": v["controls_equal_exact_path"]} if "controls_equal_exact_path" in v else {}),
                                          **({"no_key_table": v["without_key_table"]["plans_per_s"]} if "without_key_table" in v else {}))
                                  for k, v in x["other_configs"].items() if isinstance(v, dict)}
        if "launch_shapes" in x:
            e["launch_shapes"] = {k: v["plans_per_s"] for k, v in x["launch_shapes"].items() if isinstance(v, dict)}
        for k in ("early_exit", "h2d_inclusive"):
            if k in x:
                e[k] = {kk: vv for kk, vv in x[k].items() if kk != "note"}
        if "fleet_cycle" in x:
            f = x["fleet_cycle"]
            e["fleet_cycle"] = {"robots": f["robots"], **{t: {k: f[t][k] for k in ("robot_cycles_per_s", "front_end_share", "us_per_cycle",
                                                                                   "cpu_front_end_us_per_robot") if k in f[t]}
                                                          for t in ("shipped", "k10") if t in f}}
        e["seconds"] = x.get("seconds")
        out["extra"] = e
        # the metric is quoted on 256 scenes per STEP; the default schedule merges the steps of a chain into launches of
        # scenes_per_launch scenes -- the rate of the one-launch-chain-per-step schedule rides in the part that is never dropped
        if isinstance((x.get("launch_shapes") or {}).get("one_chain_per_step"), dict):
            out["config"]["plans_per_s_one_chain_per_step"] = x["launch_shapes"]["one_chain_per_step"]["plans_per_s"]
    out["full_record"] = full_path
    # never above the limit: drop the least essential parts first
    for drop in (("extra", "launch_shapes"), ("extra", "paths"), ("parity", "gpu_last_qp"), ("roofline", "pmc"), ("extra", "other_configs"),
                 ("extra", None)):
        if len(json.dumps(out)) <= COMPACT_LIMIT:
            break
        a, b = drop
        if a in out and (b is None or b in out[a]):
            if b is None:
                del out[a]
            else:
                del out[a][b]
    if len(json.dumps(out)) > COMPACT_LIMIT:
        # last resort: the contract keys, the two required objects at their required keys, nothing else -- a line is ALWAYS printed
        r = out["roofline"]
        out = {k: out[k] for k in keep + ("config",)}
        out["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")}
        if "cpu_baseline" in line:
            cb = line["cpu_baseline"]
            out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:120]}
        out["full_record"] = full_path
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)            # (whole rounds of the 40 batches in flight)
    ap.add_argument("--warmup", type=int, default=80)
    ap.add_argument("--cpu-scenes", type=int, default=0,
                    help="scenes of the first batch planned by the CPU oracle + its ensemble (rank 0, N=1); 0 = 256 on a host "
                         "with >= 64 cores, else 96")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the CPU baseline (0 = all host cores)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-scene latency measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip extra.paths / extra.uniform_cloud / extra.other_configs")
    ap.add_argument("--graph", action="store_true", help="HIP-graph replay of a step's 21 launches for the planners that carry no "
                                                        "timing events (measured: SLOWER than eager launches with 20 chains in "
                                                        "flight, 513 k vs 656 k plans/s, DESIGN.md section 3.4; default: eager)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="independent batches (steps) kept in flight, one stream each (0 = 20; 18 with a process group unless the "
                         "run is a single wave of <= 20 steps)")
    ap.add_argument("--issue-threads", type=int, default=4, help="host threads issuing the steps' launches (0: the main thread alone)")
    ap.add_argument("--burst", type=int, default=BURST_DEFAULT,
                    help="1: the steps a host thread issues in one round of the chains go out as ONE breadth-first library call "
                         "(npa_forward_batch_group: staging of every chain, then PAN iteration 0 of every chain, ...); 0: call by call")
    ap.add_argument("--chain-threads", type=int, default=CHAIN_THREADS_DEFAULT,
                    help="host threads issuing the launch chains (-1: one per chain; 0: the calling thread issues all chains of a round "
                         "in one breadth-first group call)")
    ap.add_argument("--chains", type=int, default=CHAINS_DEFAULT,
                    help="launch chains the batches in flight form: the steps of a chain share one stream and run every stage as ONE "
                         "merged launch (npa_forward_batch_group); -1 (default): batches in flight / 5; 0: one stream and one launch "
                         "chain per step (round 4's schedule)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: the launch / rendezvous / timing protocol only (gloo)")
    ap.add_argument("--workload", default=WORKLOAD,
                    choices=sorted(k for k in __import__("neupan_amd.scenes", fromlist=["CONFIGS"]).CONFIGS),
                    help="scene configuration (default: the one BASELINE.json's metric is quoted on)")
    ap.add_argument("--batch", type=int, default=BATCH, help="scenes per step and GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args.gpus)                  # (does not return)
    if args.dry_run:
        return dry_run(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "RANK" in os.environ and "MASTER_PORT" in os.environ:      # under torch.distributed.run, also with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # the measured leg imports the product only; tests/ (and with it oracle/) is touched by the cpu_baseline / parity leg alone
    from neupan_amd.scenes import CONFIGS, make_batch
    from neupan_amd.serve import bind_to_gpu_numa_node
    numa = bind_to_gpu_numa_node(local_rank)            # launch thread next to its GPU (one process per GPU)

    if os.environ.get("NPA_BENCH_RCCL_COMM_ONLY"):        # diagnostics: a raw RCCL communicator (no torch process group)
        import ctypes as _C
        _r = _C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        _comm = _C.c_void_p()
        _devs = (_C.c_int * 1)(local_rank)
        print("ncclCommInitAll rc", _r.ncclCommInitAll(_C.byref(_comm), 1, _devs), file=sys.stderr)
        if os.environ.get("NPA_BENCH_RCCL_COMM_ONLY") == "destroy":
            print("ncclCommDestroy rc", _r.ncclCommDestroy(_comm), file=sys.stderr)
    cfg = CONFIGS[args.workload]
    B = args.batch
    T, K, N = cfg.T, cfg.iter_num, cfg.n_points
    # Chains in flight.  Every chain owns a HIP stream = a hardware queue; the device schedules ~24 of them per process without
    # time-slicing.  RCCL brings three streams of its own and the gather a fourth: with 20 chains the process sits AT the limit
    # and every few hundred steps one queue is evicted in mid-kernel for ~10 ms (kernel trace: one select_geo_kernel of 10.8 ms on
    # one queue while the other 21 run on -- what rounds 2 and 3 reported as "the communicator's passive cost", 11 - 24 % of a
    # 128-step run, gone with <= 18 chains or once the communicator is destroyed).  A run of <= 20 steps
    # is ONE wave of chains: 20 chains finish it in one chain latency, 18 need two.
    # With merged launches (--chains C > 0) the chains, not the steps, own the streams: C queues whatever --inflight is, so the
    # batches in flight are no longer capped by the queue budget: 40 once the run has rounds for them (>= 3 rounds), 20 below
    # (the driver's 20-step region is one round of 20).  Default chains: 5 steps per chain (measured: gpurun_out/r05a/sweep.txt).
    merged = args.chains != 0 and bool(args.burst) and not args.graph
    if args.inflight > 0:
        nfl = args.inflight
    elif merged:
        nfl = 40 if args.steps >= 120 else 20
    else:
        nfl = 20 if (dist is None or args.steps <= 20) else 18
    Loop.BURST = bool(args.burst)
    Loop.CHAIN_THREADS = args.chain_threads
    Loop.CHAINS = (max(1, nfl // 5) if args.chains < 0 else args.chains) if merged else 0
    lp = Loop(args.workload, B, nfl, dev, rank=rank, world=world, dist=None if os.environ.get("NPA_BENCH_NOGATHER") else dist,
              graph=args.graph, issue_threads=args.issue_threads)
    E = lp.pans[0].E
    barrier = dist.barrier if dist is not None else None
    r = lp.timed(args.steps, args.warmup, barrier)
    elapsed, t_issue, prof, last = r["elapsed"], r["t_issue"], r["prof"], r["last"]
    r["elapsed_rank"] = elapsed
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    out, gathered = last[0]
    timed_u = out["opt_u"].cpu().numpy().copy()

    plans = B * world * args.steps
    value = plans / elapsed
    km = lp.pans[0].key_mode()

    # the run-time audit of the geometric-key margin over everything this process launched so far (all planners): a
    # violation would mean plans computed from a wrong selection -- the number is reported AND must be zero
    audit = lp.audit()
    assert audit["violations"] == 0, f"geometric-key margin violated at run time: {audit}"

    # ---- roofline of the dominant kernel -------------------------------------------------------------------------
    # The step is dominated by the QP: a serial fp64 interior-point chain, one wave per scene -- VALU / latency bound, no MFMA,
    # no HBM stream.  Work per launch = interior-point iterations the launch executes (MEASURED HERE on the device: qp_info of
    # every scene after every PAN iteration of one untimed step) x fp64 flops per iteration + a fixed part per solve (set-up,
    # write-out, stop test); the two per-unit figures come from PMC counts (SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 x 64 lanes, FMA x2)
    # at two operating points (warm / cold solves), tests/tools/pmc_collect.py -> profiles/r04_pmc.json, DESIGN.md section 6.
    # Launch time: HIP events on the launches' own stream, inside the timed region.  SURVEY 8(d)'s figure -- algorithmic dense
    # flops of the encoder over the fp32-MFMA peak -- is not the denominator of the default path: with geometric keys the
    # encoder runs on ~2 % of the points (943 Mflop/plan x this rate would be 4x the peak); extra.paths reports it for the
    # un-pruned selections, roofline.select for the default one.
    pmc = load_pmc(args.workload)
    its = lp.qp_iterations()
    # steps that share a launch (merged chains): the members of a chain, <= 8 per run (a full round of the slots; a trailing
    # partial round has shorter runs -- the default step counts are whole rounds)
    mem = (nfl + lp.chains - 1) // lp.chains if lp.chains else 1
    mem = (mem + ((mem + 7) // 8) - 1) // ((mem + 7) // 8)
    spl = B * mem                               # scenes per launch
    its_per_launch = float(np.mean([s for s, _ in its])) * mem      # (batch 0's count stands for its chain mates)
    qp_ms, sel_ms, dune_ms = prof["nrmp_ms"], prof["select_ms"], prof["dune_ms"]
    qp_name = "nrmp_qp_group_kernel" if lp.chains else "nrmp_qp_kernel"
    sel_name = "select_geo_group_kernel" if lp.chains else "select_geo_kernel"
    roof = {"bound": "valu", "kernel": f"{qp_name}<{T},{cfg.nrmp_max_num}>", "unit": "TFLOP/s", "peak": PEAK_FP64_VALU_TFLOPS,
            "launch_ms": round(qp_ms, 4), "scenes_per_launch": spl, "launches_timed": prof["launches"], "select_launch_ms": round(sel_ms, 4),
            "dune_launch_ms": round(dune_ms, 4), "key_mode": km, "achieved": None, "frac": None, "traffic": None,
            "ipm_iterations_per_launch": round(its_per_launch, 1),
            "ipm_iterations_by_pan_iteration": [[int(s), int(m)] for s, m in its]}
    kq = (pmc["kernels"].get(qp_name) or pmc["kernels"].get("nrmp_qp_kernel")) if pmc else None
    if kq:
        kq_spl = kq.get("scenes_per_launch", pmc["scenes_per_launch"])
        model = pmc.get("qp_flops_model")
        if model:
            flops = model["per_iteration"] * its_per_launch + model["per_solve"] * spl
            roof["flops_model"] = dict(model, units="fp64 flop per interior-point iteration of one scene / per solve")
        else:                                   # (a record without the two-point fit: its launch average, scaled to this batch)
            flops = kq["fp64_flops_per_launch"] * spl / kq_spl
        roof["achieved"] = round(flops / (qp_ms * 1e-3) / 1e12, 4) if qp_ms > 0 else None
        roof["frac"] = round(roof["achieved"] / PEAK_FP64_VALU_TFLOPS, 5) if roof["achieved"] else None
        roof["flops_per_launch"] = int(flops)
        roof["traffic"] = int(kq["hbm_bytes_per_launch"] * spl / kq_spl)
        # algorithmic bytes of the QP launch: the rows it consumes (T+1 slices x M x (E + 5) floats + counts), the nominal in
        # and out, the references and the outputs -- what a fused loop would still have to move is far less
        roof["algorithmic_bytes_per_launch"] = int(spl * 4 * ((T + 1) * cfg.nrmp_max_num * (E + 5) + (T + 1) + 4 * (3 * (T + 1) + 2 * T) + 3 * T))
        roof["valu_issue_frac_alone"] = kq.get("valu_issue_frac")
        roof["frac_alone"] = (round(flops * (kq_spl / spl) / (kq["avg_ms_alone"] * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS, 5)
                              if kq.get("avg_ms_alone") else None)
        roof["lds_bank_conflict_frac"] = kq.get("lds_bank_conflict_frac")
        # the chip's fp64 rate over the whole timed region: every QP launch of every chain / wall time
        agg = flops / mem * K * args.steps / elapsed / 1e12
        roof["chip_aggregate"] = {"tflops": round(agg, 3), "frac": round(agg / PEAK_FP64_VALU_TFLOPS, 5)}
        roof["pmc"] = {"file": pmc["_file"],
                       "per_kernel": {k: dict({kk: v[kk] for kk in ("valu_insts_per_launch", "valu_issue_frac", "hbm_bytes_per_launch",
                                                                      "avg_ms_alone", "mfma_busy_frac", "source_hash", "isa_hash") if kk in v},
                                                  current=record_is_current(k, v))
                                      for k, v in pmc["kernels"].items()}}
    ks = (pmc["kernels"].get(sel_name) or pmc["kernels"].get("select_geo_kernel")) if pmc else None
    if ks and sel_ms > 0 and "SQ_INSTS_MFMA" in ks.get("counters", {}):
        # the exact encoder's matrix work: executed v_mfma_f32_32x32x2_f32 (4096 flop each) over the fp32-MFMA peak
        ks_spl = ks.get("scenes_per_launch", pmc["scenes_per_launch"])
        mf = ks["counters"]["SQ_INSTS_MFMA"] * 4096.0 * spl / ks_spl
        roof["select"] = {"kernel": f"{sel_name}<{E}>", "launch_ms": round(sel_ms, 4), "mfma_flops_per_launch": int(mf),
                          "mfma_tflops": round(mf / (sel_ms * 1e-3) / 1e12, 3), "mfma_peak": PEAK_FP32_MFMA_TFLOPS,
                          "mfma_frac": round(mf / (sel_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 5),
                          "valu_issue_frac_alone": ks.get("valu_issue_frac"), "mfma_busy_frac_alone": ks.get("mfma_busy_frac"),
                          "traffic": int(ks["hbm_bytes_per_launch"] * spl / ks_spl),
                          "algorithmic_bytes_per_launch": int(spl * (8 * N * (2 if lp.args[0][5] is not None else 1) + 400))}
    # the loop as a whole against the chip's vector pipes: SIMD-cycles the two kernels keep a SIMD's VALU or MFMA pipe busy per
    # plan (PMC: 4 x SQ_ACTIVE_INST_VALU quad-cycles + SQ_VALU_MFMA_BUSY_CYCLES per launch, alone on the chip; fp32 MFMA and VALU
    # do not co-execute on gfx950) -> the rate at which 1024 SIMDs at 2.4 GHz would run THIS instruction mix with no idle cycle
    if pmc and km["key_terms"] == 4:
        def pipe_cycles(name):
            k_ = pmc["kernels"].get(name)
            c_ = (k_ or {}).get("counters", {})
            if "SQ_ACTIVE_INST_VALU" not in c_:
                return None
            return (4.0 * c_["SQ_ACTIVE_INST_VALU"] + c_.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)) / k_.get("scenes_per_launch", pmc["scenes_per_launch"])
        pc_sel = pipe_cycles(sel_name) or pipe_cycles("select_geo_kernel")
        pc_qp = pipe_cycles(qp_name) or pipe_cycles("nrmp_qp_kernel")
        pc_st = pipe_cycles("stage_group_kernel" if lp.chains else "stage_kernel") or 0.0
        if pc_sel and pc_qp and args.workload == WORKLOAD:
            per_plan = K * (pc_sel + pc_qp) + pc_st
            bound = N_SIMD * 2.4e9 / per_plan
            roof["vector_pipe"] = {"simd_cycles_per_plan": int(per_plan), "selection_share": round(K * pc_sel / per_plan, 3),
                                   "bound_plans_per_s": int(bound), "frac": round(value / world / bound, 4),
                                   "note": "VALU + MFMA busy SIMD-cycles of selection, QP and staging per plan (PMC, kernels alone) -> plans/s "
                                           "of 1024 SIMDs at 2.4 GHz with no idle cycle; frac = this run's rate per GPU over it"}
    roof["note"] = ("dominant kernel by GPU time = the QP (fp64 Mehrotra IPM, one wave per scene, serial chain: latency / VALU-issue "
                    "bound).  achieved = (interior-point iterations per launch, measured on the device in this run) x (fp64 flops per "
                    "iteration) + solves x (fixed flops per solve), the two per-unit figures from PMC counts at two operating points "
                    "(roofline.flops_model), / launch time measured here with HIP events on the launch's stream (other batches' kernels "
                    "co-run on the same SIMDs; frac_alone: over the kernel's duration alone on the chip; chip_aggregate: all QP "
                    "launches of the region over its wall time); nothing is priced above what it executes.  DUNE: " +
                    ("geometric distance keys inside select_geo_kernel nominate the candidates, the exact fp32-MFMA encoder runs on "
                     "those only (~1.1 tiles of 32 points per slice instead of N/32); no dune_kernel launch" if km["key_terms"] == 4
                     else f"network keys (mode {km['key_terms']}) from dune_kernel over every point, exact re-encode of the candidates"))
    mfd = dune_mfma_rate(km, cfg, B, dune_ms, args.steps, elapsed)
    if mfd:
        roof["dune_executed_mfma"] = mfd

    line = {
        "metric": "MPC plans/sec (node), diff robot, 1k pts, T=10, K=10; ctrl L2 vs ref" if args.workload == WORKLOAD
                  else f"MPC plans/sec (node), workload {args.workload}; ctrl L2 vs ref",
        "value": round(value, 1), "unit": "plans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[1]: batch=256 synthetic scenes/GPU, diff robot, 1000 pts, "
                                "T=10, K=10 (iter_threshold=0), M=10, fp32 DUNE (MFMA) + fp64 QP") if args.workload == WORKLOAD
                               else f"{args.workload}: batch={B} synthetic scenes/GPU, {cfg.kinematics} robot, {N} pts, "
                                    f"T={T}, K={K} (iter_threshold=0), M={cfg.nrmp_max_num}, fp32 DUNE (MFMA) + fp64 QP"
                                    + ("; rows exact fp32 (the labelled bf16 tier of BASELINE configs[4]: NPA_ROWS_PRECISION=bf16, "
                                       "extra.other_configs of the default line)" if args.workload.startswith("poly8")
                                       and os.environ.get("NPA_ROWS_PRECISION") != "bf16" else "")
                                    + ("; rows from the LABELLED bf16 tier (not the reference's arithmetic)"
                                       if os.environ.get("NPA_ROWS_PRECISION") == "bf16" else ""),
                   "scenes_per_gpu": B, "points": N, "T": T, "K": K, "M": cfg.nrmp_max_num,
                   "batches_in_flight": nfl, "chains": lp.chains,
                   "schedule": ((f"{lp.chains} launch chains of {mem} steps each: the steps of a chain (own scenes, buffers, planner state) are issued "
                                 "as ONE npa_forward_batch_group call on ONE stream and every stage of theirs runs as one merged launch "
                                 f"over {spl} scenes (blockIdx.y = the step), " if lp.chains else
                                 "one HIP stream per batch in flight, one prepared library call per step (PAN.make_step), " +
                                 ("the chains of a round issued breadth-first (npa_forward_batch_group), " if lp.loop.groups is not None else "")) +
                                ("eager launches" if not args.graph else "HIP-graph replay except on the planners that carry timing events") +
                                ", gathers on one communication stream"),
                   "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "numa_node": numa, "issue_threads": lp.loop.threads,
                   "parallelism": f"scene-shard x{world}, RCCL all-gather of controls"
                                  + (f" (process group initialised; {lp.gatherer.collectives} collectives for {lp.gatherer.issued} "
                                     f"steps: one all_gather_into_tensor per {nfl} steps, all inside the timed loop)"
                                     if dist is not None else " (no process group in a single-process run: nothing to gather)"),
                   "inputs": "resident in HBM and re-planned every step; the 8 N B of points per scene a host caller ships per "
                             "step over PCIe are NOT in the timed region (DESIGN.md section 6)"},
        "roofline": roof,
        # host time to enqueue a step (one library call = 21 launches) next to the step's wall time: when the two are close
        # the step is bound by the host's launch rate, not by the kernels
        "host_issue_ms_per_step": round(1e3 * t_issue / args.steps, 4),
        # rank 0's timed region in its three parts: issuing the steps, waiting for the device, the closing barrier of the contract
        "region_ms": {"issue": round(1e3 * t_issue, 3), "drain": round(1e3 * (r["t_drain"] - t_issue), 3),
                      "closing_barrier": round(1e3 * (r["elapsed_rank"] - r["t_drain"]), 3)},
        "margin_audit": audit,
    }

    # ---- single-scene latency: the reference's actual use (neupan/neupan.py:104-166, one robot, README "15 Hz") -------
    if rank == 0 and not args.no_latency:
        lat = {}
        for tag, over in (("K10_N1000", {}), ("shipped_K2_N100", dict(iter_num=2, dune_max_num=100, iter_threshold=0.1))):
            p1 = make_gpu_pan(cfg, device=dev, **over)
            p1.printed = True                   # the reference prints a decimation notice once (pan.py:172): keep stdout to ONE line
            a1 = [a[:1].contiguous() if a is not None else None for a in lp.args[0]]
            ts = []
            for rep_ in range(60):
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

    extras = rank == 0 and world == 1 and not args.no_extras and args.workload == WORKLOAD
    with_cpu = rank == 0 and world == 1 and not args.no_cpu
    cores = args.cpu_cores if args.cpu_cores > 0 else (os.cpu_count() or 1)
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from parity_tools import gpu_last_qp_certificates, host_cores, run_ensemble
        try:                                    # the CPU baseline gets the whole host, not the GPU's NUMA node
            os.sched_setaffinity(0, range(os.cpu_count() or 1))
        except Exception:
            pass
        host = os.cpu_count() or 1
        n_sc = args.cpu_scenes if args.cpu_scenes > 0 else (B if host >= 64 else min(96, B))
        n_sc = min(n_sc, B)
        rep, cpu_rate, ncore, hip, sp, tr = parity_leg(lp, n_sc, cores, sweep=True)
        # (controls after every PAN iteration of the same batch, untimed; its last iteration IS the timed result)
        assert np.array_equal(tr["trace_u"].cpu().numpy()[:, -1], timed_u), "traced run differs from the timed run"
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
                                "sample_short": f"first {n_sc} scenes of the same workload, K={K} each, oracle/pan_oracle.py (numpy fp32 + fp64 IPM), "
                                                f"worker processes x 1 thread, best of a sweep over the worker count",
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
        if well.any() and n_sc == B:
            idx = torch.from_numpy(np.resize(np.flatnonzero(well), B)).to(dev)
            lw = Loop(args.workload, B, nfl, dev, issue_threads=args.issue_threads, scene_index=idx)
            rw = lw.timed(args.steps, args.warmup)
            rep["well_posed_only"] = {"scenes": int(well.sum()), "plans_per_s": round(B * args.steps / rw["elapsed"], 1), "steps": args.steps,
                                      "ctrl_l2_max": float(hip[well, -1].max()), "ctrl_l2_median": float(np.median(hip[well, -1])),
                                      "note": "the same loop on a batch made of the well-posed scenes only (cycled to 256)"}
            lw.close()
        rep["note"] = ("oracle = reference code restated + substituted fp64 QP solver (ECOS unavailable: parity unpinned at that "
                       "boundary).  Ensemble per scene = the oracle itself on inputs moved by +-1 float32 ulp (8 members) and with the "
                       "DUNE hidden units permuted (same function, other fp32 summation order; 4 members).  well posed = ensemble "
                       "spread of the final controls <= 1e-4.  A: HIP <= 1e-4 on every well-posed scene; B: HIP inside the ensemble "
                       "spread elsewhere (reported; a 14th sample of a chaotic scene need not fall inside the hull of 13); C: HIP <= 1e-5 at "
                       "every iteration before the ensemble itself first disagrees by > 1e-5; D (one_step): on every scene and "
                       "iteration ONE oracle iteration from the HIP path's own iterate reproduces the HIP path's next iterate; a step "
                       "above 1e-4 must be EXPLAINED (a tie at rank M / M+1 of a slice that the two fp32 encoders order differently, "
                       "or a QP flat enough that two 1e-14 solves agree in objective to 1e-12 and not in the controls) "
                       "(tests/parity_tools.py, DESIGN.md section 5)")
        # the kernel's own last QP, per scene: rebuilt on the host from the parameters the kernel built, fp64 solution certified
        batch0 = make_batch(cfg, rank * nfl * B, B)
        rep["gpu_last_qp"] = dict(gpu_last_qp_certificates(lp.pans[0], cfg, batch0),
                                  note="KKT certificate (NNLS stationarity, complementarity, feasibility) of the kernel's fp64 "
                                       "solution of its last QP and objective gap to the oracle's solve of the same problem, "
                                       "all scenes of the batch; stat_oracle = the same certificate on the oracle's solutions")
        line["parity"] = rep
    lp_chains = lp.chains
    lp.close()

    # ---- the other paths and configurations, in the same line (GPU legs are short loops; parity legs: 16 scenes each) -----
    if extras:
        t_ex = time.perf_counter()
        ex = {}
        nfx = 20                                # batches in flight of every extra leg (four chains of five where the steps merge)
        Loop.CHAINS = 4 if merged else 0
        paths = {}
        for tag, env in (("exact_fp32_keys", {"NPA_DUNE_FP32KEYS": "1"}), ("network_keys_1", {"NPA_KEY_TERMS": "1"}),
                         ("network_keys_3", {"NPA_KEY_TERMS": "3"})):
            res, l2 = short_run(WORKLOAD, B, nfx, dev, 60, 20, env=env, issue_threads=args.issue_threads)
            # the same plans as the default path (the keys only nominate): bitwise, checked on batch 0
            l2.pans[0].reset_stop_state()
            res["controls_equal_default_path"] = bool(np.array_equal(l2.pans[0].forward_batch(*l2.args[0])["opt_u"].cpu().numpy(), timed_u))
            res["env"] = env
            paths[tag] = res
            l2.close()
        ex["paths"] = dict(paths, note="the default loop (256 scenes / step, 20 in flight, 60 steps) with the selection's other key "
                                       "paths: dune_kernel encodes EVERY point of every slice (SURVEY 8(d)'s literal path), "
                                       "dune_executed_mfma prices that work against the peak of the MFMA it runs on")
        # SURVEY 8(d)'s uniform cloud
        res, l2 = short_run("uniform_1k_T10_K10", B, nfx, dev, 60, 20, issue_threads=args.issue_threads)

        def cand(workload, a_):
            with environ({"NPA_SEL_DEBUG": "1"}):
                dbg = make_gpu_pan(CONFIGS[workload], device=dev)
                c_ = dbg.dune_stage(a_[0], a_[4], a_[5])["count"].cpu().numpy()
                del dbg
            nc, fb = (c_ >> 8) & 0xFF, c_ >> 16
            return {"median": int(np.median(nc)), "mean": round(float(nc.mean()), 1), "p90": int(np.quantile(nc, 0.9)), "max": int(nc.max()),
                    "share_gt_32": round(float((nc > 32).mean()), 4),
                    # fb: 1 = exact keys for the whole list, 2 = the table filter shortened it first, 3 = the filter decided the slice
                    "share_overflow_to_exact_keys": round(float(((fb == 1) | (fb == 2)).mean()), 4),
                    "share_decided_by_table_filter": round(float((fb == 3).mean()), 4),
                    "histogram_0_16_32_64_128_256": np.histogram(nc, bins=[0, 16, 32, 64, 128, 256])[0].tolist()}
        res["candidates_per_slice"] = cand("uniform_1k_T10_K10", l2.args[0])
        b0 = make_batch(cfg, 0, B)
        res["candidates_per_slice_corridor_workload"] = cand(WORKLOAD, [b0["nom_s"], None, None, None, b0["points"], None])
        if with_cpu:
            # (32 scenes x 12 members, like the headline's ensemble: "no well-posed scene" is then a measured statement)
            res["parity"] = slim(parity_leg(l2, 32, cores, n_ulp=8, n_perm=4)[0])
        res["note"] = ("SURVEY 8(d) config 2's cloud exactly as specified (uniform x in [-2, 12], y in [-6, 6], rejection box): 6 points "
                       "per m^2 around a 1.6 x 2.0 m robot -- most scenes have no collision-free plan (d at d_min), the PAN iteration is "
                       "ill posed on them; candidates per slice = points the geometric keys could not rule out (255 = capped)")
        ex["uniform_cloud"] = res
        l2.close()
        others = {}
        exact_u = {}
        for tag, wl, b_, nf, env in (("acker_2k_T20_K15", "acker_2k_T20_K15", B, nfx, None),
                                     ("dyna_4k_T10_K10_batch1024", "dyna_4k_T10_K10", 1024, 4, None),
                                     ("poly8_5k_T10_K10_exact_fp32_rows", "poly8_5k_T10_K10", B, nfx, None),
                                     ("poly8_5k_T10_K10_bf16_keys", "poly8_5k_T10_K10", B, nfx, {"NPA_KEYS_PRECISION": "bf16"}),
                                     ("poly8_5k_T10_K10_bf16_rows", "poly8_5k_T10_K10", B, nfx, {"NPA_ROWS_PRECISION": "bf16"})):
            # (K = 15 / T = 20 and 5000-point chains are 20 - 30 ms long: enough steps for several rounds of the chains in flight;
            # these heavier workloads fill the chip with one launch chain per step: merged chains cost them 2 - 6 %, measured)
            res, l2 = short_run(wl, b_, nf, dev, 24 if b_ > B else 100, 8 if b_ > B else 20, env=env, issue_threads=args.issue_threads,
                                chains=0)
            if env:
                res["env"] = env
            lossy = bool(env) and "NPA_ROWS_PRECISION" in env
            l2.pans[0].reset_stop_state()
            u_ = l2.pans[0].forward_batch(*l2.args[0])["opt_u"].cpu().numpy()
            if not env:
                exact_u[wl] = u_
                if wl != "acker_2k_T20_K15":
                    # the same leg without the key table (NPA_GEO_TABLE=0: round 4's selection -- exact keys for every long
                    # candidate list): what the second-stage filter is worth on the dense clouds; controls bitwise equal
                    r0, l0 = short_run(wl, b_, nf, dev, 24 if b_ > B else 100, 8 if b_ > B else 20, env={"NPA_GEO_TABLE": "0"},
                                       issue_threads=args.issue_threads, chains=0)
                    l0.pans[0].reset_stop_state()
                    res["without_key_table"] = {"plans_per_s": r0["plans_per_s"], "select_launch_ms": r0["select_launch_ms"],
                                                "controls_equal": bool(np.array_equal(l0.pans[0].forward_batch(*l0.args[0])["opt_u"].cpu().numpy(), u_))}
                    l0.close()
            elif wl in exact_u:
                # the bf16 KEY tier nominates with the bf16-MFMA encoder and emits exact rows: bitwise the exact path's controls
                res["controls_equal_exact_path"] = bool(np.array_equal(u_, exact_u[wl]))
            if with_cpu:
                # (the bf16 ROWS tier is judged like the others -- it fails A / C / D by design, that is what its entry shows --
                # without the per-step explanations: a different selection through rounded distances needs none)
                res["parity"] = slim(parity_leg(l2, 16, cores, n_ulp=4, n_perm=2, explain=not lossy)[0])
                if lossy:
                    res["parity"]["note"] = ("LABELLED reduced-precision tier: not the reference's arithmetic; the verdicts are "
                                             "reported, not expected to hold")
            others[tag] = res
            l2.close()
        ex["other_configs"] = dict(others, note="BASELINE configs[2] (car, reverse gear on half the scenes), configs[3]'s per-GPU shape "
                                                "(8192 scenes / 8 GPUs = 1024 per step, 4 steps in flight), configs[4] (8-edge hull; the "
                                                "reference ships no E = 8 checkpoint: ours, trained with its recipe on closed-form labels) in "
                                                "exact fp32, with the bf16 tier of the KEYS (NPA_KEYS_PRECISION=bf16: v_mfma_f32_32x32x16_bf16 "
                                                "filters the overflowing candidate lists, the rows stay exact: controls bitwise the exact path's) "
                                                "and in the LABELLED bf16 tier of the ROWS (its parity entry shows what that costs); parity = "
                                                "ensemble verdicts on the first 16 scenes (6 members)")
        # the same 20 batches in flight under other schedules: one stream and one launch chain per step (round 4's default),
        # and other numbers of merged chains.  NOT other workloads: every step is 256 scenes with its own planner state.
        shapes = {}
        for tag, ch in (("one_chain_per_step", 0), ("2_chains", 2), ("10_chains", 10)):
            if ch == lp_chains:
                continue
            res, l2 = short_run(WORKLOAD, B, nfx, dev, 60, 20, issue_threads=args.issue_threads, chains=ch)
            shapes[tag] = {k: res[k] for k in ("plans_per_s", "ms_per_step", "steps", "scenes_per_step", "batches_in_flight", "chains",
                                               "select_launch_ms", "qp_launch_ms")}
            l2.close()
        ex["launch_shapes"] = dict(shapes, note="the default workload and batch size under other launch schedules (60 steps): "
                                                "one_chain_per_step = round 4's default (20 streams, 1 + 2K launches per step); n_chains = "
                                                "the 20 batches in flight merged into n launch chains")
        # the closed-loop cycle around the PAN loop, measured today (round 1's figure predates every kernel here)
        ex["fleet_cycle"] = fleet_cycle_leg(dev, with_cpu=with_cpu)
        # SURVEY 8(d)'s second run: the reference's default stop threshold (pan.py:243: iter_threshold = 0.1).  Every step starts
        # from a cleared stop-criterion state (like every other leg), so a scene runs at least 2 iterations: the first only
        # stores its iterate (pan.py:218-221)
        le = Loop(WORKLOAD, B, nfx, dev, issue_threads=args.issue_threads, over={"iter_threshold": 0.1})
        re_ = le.timed(60, 20)
        it = le.last_iters
        ex["early_exit"] = {"iter_threshold": 0.1, "plans_per_s": round(B * 60 / re_["elapsed"], 1), "steps": 60,
                            "iterations_mean": round(float(it.mean()), 3), "iterations_max": int(it.max()), "iterations_min": int(it.min()),
                            "share_stopped_before_K": round(float((it < K).mean()), 4),
                            "note": "the default loop with the reference's default iter_threshold (pan.py:243); executed PAN iterations per "
                                    "scene over the last step of every batch in flight; the launches of a step still number K (a stopped "
                                    "scene's waves return at once)"}
        le.close()
        # PCIe-inclusive: every step's inputs (nominal and reference trajectories, the cloud) uploaded from pinned host memory on
        # the step's own stream in front of the step -- what a host caller of the reference does per cycle (neupan.py:123-127)
        lh = Loop(WORKLOAD, B, nfx, dev, issue_threads=args.issue_threads, h2d=True)
        rh = lh.timed(60, 20)
        ex["h2d_inclusive"] = {"plans_per_s": round(B * 60 / rh["elapsed"], 1), "steps": 60, "bytes_per_step": int(lh.h2d_bytes),
                               "upload_GBps": round(lh.h2d_bytes * 60 / rh["elapsed"] / 1e9, 2),
                               "note": "inputs of every step copied host (pinned) -> device on the step's stream inside the timed region; "
                                       "NOT the headline (whose inputs are resident in HBM)"}
        lh.close()
        ex["seconds"] = round(time.perf_counter() - t_ex, 1)
        line["extra"] = ex
    if rank == 0:
        full_path = os.environ.get("NPA_BENCH_FULL", os.path.join(ROOT, "bench_full.json"))
        try:
            with open(full_path, "w") as f:
                f.write(json.dumps(line) + "\n")
        except OSError as e:                      # (a read-only tree: the compact line is still the record)
            full_path = f"not written: {e}"
        short = compact(line, os.path.relpath(full_path, ROOT) if os.path.isabs(full_path) and full_path.startswith(ROOT) else full_path)
        txt = json.dumps(short)
        print(txt, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
