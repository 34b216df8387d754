"""world_size-2 gloo test of the scene-shard + gather path (CPU).  The planner is replaced by
a stand-in whose output is a known function of the global scene index, so the test pins the
sharding arithmetic, the rank order of the gather and the ragged-shard padding."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from neupan_amd.dist import plan_sharded, shard_range


class _StandIn:
    T = 10

    def forward_batch(self, idx):
        u = torch.stack([torch.full((2, self.T), float(i)) for i in idx]) if len(idx) else torch.zeros((0, 2, self.T))
        return {"opt_u": u}


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = plan_sharded(_StandIn(), lambda lo, hi: (list(range(lo, hi)),), total, dist, rank, world)
    q.put((rank, out[:, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for total in (0, 1, 7, 256, 8192):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_gather_is_in_scene_order(total):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert got[r] == [float(i) for i in range(total)]


@pytest.mark.gpu
def test_rccl_all_gather_branch_on_one_gpu():
    """The production branch of gather_controls -- one flat all_gather_into_tensor over RCCL (backend "nccl") -- on the
    GPU at hand: a process group of ONE rank still goes through the collective (bench.py does the same when it is
    launched under torch.distributed.run with one process).  Ragged shards take the padded branch."""
    import torch.distributed as dist
    from neupan_amd.dist import gather_controls
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        u = torch.arange(256 * 2 * 10, dtype=torch.float32, device=dev).reshape(256, 2, 10)
        calls = []
        orig = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda out, inp, *a, **k: (calls.append(tuple(inp.shape)), orig(out, inp, *a, **k))[1]
        try:
            g = gather_controls(u, dist, 1, equal_shards=True)
            g2 = gather_controls(u, dist, 1, equal_shards=False)
        finally:
            dist.all_gather_into_tensor = orig
        torch.cuda.synchronize()
        assert len(calls) == 2 and calls[0] == (256, 2, 10)
        assert g.data_ptr() != u.data_ptr() and torch.equal(g, u) and torch.equal(g2, u)
    finally:
        dist.destroy_process_group()


def _value(rank, i):
    """controls of global step i on `rank` (every element)"""
    return float(10000 * rank + i)


def _loop_worker(rank, world, port, inflight, runs, threads, q, burst=False):
    """The serving loop (neupan_amd.serve: StepLoop / run_steps + ControlGatherer) with stand-in planners on CPU tensors over
    gloo: the SAME coalesced protocol the GPU runs (staging rows, alternating buffers, flush order, trailing partial group,
    consecutive runs on one gatherer), `threads` issuing threads per rank, and uneven progress: rank 1's planners sleep,
    and its slot 0 sleeps longest."""
    import time
    import torch.distributed as dist
    from neupan_amd.serve import ControlGatherer, StepLoop
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T = 4, 10
    counter = {"base": 0}
    ran = [0] * inflight

    def make(j):
        def step():
            i = counter["base"] + j + inflight * ran[j]          # slot j runs steps j, j + inflight, ... of the current run
            ran[j] += 1
            if rank == 1:
                time.sleep(0.004 if j == 0 else 0.0005 * (j % 3))
            return {"opt_u": torch.full((B, 2, T), _value(rank, i)), "index": i}
        return step

    snaps = []

    class Recording(ControlGatherer):
        def _flush(self, g):
            rows = self._pending
            super()._flush(g)
            snaps.append((g, rows, self.out[g & 1][:, :, 0, 0, 0].clone()))       # (world, slots): first element of every row

    g = Recording(dist, world, device=None, slots=inflight, shape=(B, 2, T))
    class Group:                                   # stand-in of neupan_amd.pan.StepGroup: the first n members, one library call
        calls = 0

        def __init__(self, steps, streams):
            self.steps = steps

        def issue(self, n=None):
            Group.calls += 1
            return [s() for s in self.steps[:len(self.steps) if n is None else n]]

    loop = StepLoop([make(j) for j in range(inflight)], None, g, None, threads=threads, burst=burst, group_cls=Group)
    assert (loop.groups is not None) == bool(burst)
    report = []
    for n in runs:
        base = g._next
        counter["base"] = base
        for j in range(inflight):
            ran[j] = 0
        last = loop.run(n)
        rows = []
        for j, item in enumerate(last):
            if item is None:
                rows.append(None)
                continue
            o, gathered = item
            rows.append((o["index"], tuple(gathered.shape), [float(gathered[r, 0, 0, 0]) for r in range(world)]))
        report.append((base, rows))
    loop.close()
    if burst:                                      # one group call per (thread, round) that has a step of that thread
        nw = max(min(threads, inflight), 1)
        want = sum(sum(1 for w in range(nw) if any((i % inflight) % nw == w for i in range(r0, min(n, r0 + inflight))))
                   for n in runs for r0 in range(0, n, inflight))
        assert Group.calls == want, (Group.calls, want)
    q.put((rank, report, [(gg, rows, t.tolist()) for gg, rows, t in snaps], g.issued, g.collectives))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("inflight,runs,threads,burst", [(1, (3,), 0, False), (4, (10,), 0, False), (5, (3,), 0, False),
                                                         (5, (3, 12, 23), 3, False), (4, (9, 2, 8), 2, False),
                                                         (4, (10, 3), 0, True), (5, (3, 12, 23), 3, True), (6, (4, 20), 2, True)])
def test_serving_loop_with_two_ranks(inflight, runs, threads, burst):
    """Two ranks over gloo drive the coalesced gather exactly as the GPU loop does: every group's collective carries the
    controls of the SAME steps from both ranks, in rank order; one collective per `inflight` steps plus one per trailing
    partial group; no deadlock with several issuing threads, fewer steps than slots, or a slow rank; consecutive runs on
    one gatherer start on fresh groups.  burst: the steps a thread issues in one round of the slots go out as one group call
    (StepLoop(burst=True), the breadth-first issue of npa_forward_batch_group) -- same collectives, same contents."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, inflight, runs, threads, q, burst)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, report, snaps, issued, collectives = q.get(timeout=180)
        got[rank] = (report, snaps)
        assert issued == sum(runs)
        assert collectives == sum((n + inflight - 1) // inflight for n in runs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]                       # both ranks saw the same gathered data, group by group
    report, snaps = got[0]
    # every collective: row r of group g holds step g * inflight + r of BOTH ranks
    expect_groups = []
    base = 0
    for n in runs:
        for k in range((n + inflight - 1) // inflight):
            expect_groups.append((base // inflight + k, min(inflight, n - k * inflight)))
        base += (n + inflight - 1) // inflight * inflight
    assert [(g, rows) for g, rows, _ in snaps] == expect_groups
    for g, rows, t in snaps:
        for r in range(rows):
            assert [t[w][r] for w in range(2)] == [_value(w, g * inflight + r) for w in range(2)], (g, r)
    # what run() returns per slot: the gathered controls of that slot's last step
    for (base, rows), n in zip(report, runs):
        for j, row in enumerate(rows):
            times = len(range(j, n, inflight))
            if times == 0:
                assert row is None
                continue
            idx, shape, firsts = row
            assert idx == base + j + inflight * (times - 1)
            assert shape == (2, 4, 2, 10)
            assert firsts == [_value(w, idx) for w in range(2)]


@pytest.mark.gpu
def test_coalesced_gather_over_rccl_on_one_gpu():
    """ControlGatherer on the GPU: the slots' controls sit in one staging buffer and cross the fabric with ONE
    all_gather_into_tensor per `slots` steps (plus one for a trailing partial group at join()).  A process group of one
    rank over RCCL exercises the production branch; the gathered rows must be the latest controls of every slot."""
    import torch.distributed as dist
    from neupan_amd.serve import ControlGatherer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        slots, B, T = 3, 8, 10
        g = ControlGatherer(dist, 1, device=dev, slots=slots, shape=(B, 2, T))
        streams = [torch.cuda.Stream(device=dev) for _ in range(slots)]
        outs = [torch.empty((B, 2, T), device=dev) for _ in range(slots)]       # "the planners'" reused output tensors
        views = [None] * slots
        base = g.begin(7)
        for i in range(7):                       # 2 full groups + 1 trailing step
            j = i % slots
            with torch.cuda.stream(streams[j]):
                outs[j].fill_(float(100 * i + j))
            g.stage(base + i, outs[j], streams[j])
            views[j] = g.collect(base + i, outs[j])
        g.join()
        torch.cuda.synchronize()
        assert g.issued == 7 and g.collectives == 3
        want = [600.0, 401.0, 502.0]             # the last step of every slot
        for j in range(slots):
            assert tuple(views[j].shape) == (1, B, 2, T)
            assert torch.all(views[j] == want[j]), (j, float(views[j].flatten()[0]))
        # a second run on the same gatherer starts on a fresh group
        base = g.begin(1)
        assert base == 9
        outs[1].fill_(7.0)
        g.stage(base, outs[1])
        v = g.collect(base, outs[1])
        g.join(); torch.cuda.synchronize()
        assert torch.all(v == 7.0) and g.collectives == 4
        # stage_many (round 6): the steps of a merged launch chain -- ONE stream -- staged with one fused copy and one event;
        # two chains of two steps per group of four slots, three groups, values that tell step and slot apart
        slots2 = 4
        g2 = ControlGatherer(dist, 1, device=dev, slots=slots2, shape=(B, 2, T))
        chains = [torch.cuda.Stream(device=dev) for _ in range(2)]
        outs2 = [torch.empty((B, 2, T), device=dev) for _ in range(slots2)]
        views2 = [None] * slots2
        n = 3 * slots2
        base = g2.begin(n)
        for r0 in range(0, n, slots2):
            for c in range(2):                                   # chain c owns slots c and c + 2
                mine = [r0 + c, r0 + c + 2]
                with torch.cuda.stream(chains[c]):
                    for i in mine:
                        outs2[i % slots2].fill_(float(10 * i + 1))
                g2.stage_many([base + i for i in mine], [outs2[i % slots2] for i in mine], chains[c])
            for i in range(r0, r0 + slots2):
                views2[i % slots2] = g2.collect(base + i, outs2[i % slots2])
        g2.join(); torch.cuda.synchronize()
        assert g2.collectives == 3
        for j in range(slots2):
            assert torch.all(views2[j] == float(10 * (2 * slots2 + j) + 1)), (j, float(views2[j].flatten()[0]))
    finally:
        dist.destroy_process_group()
