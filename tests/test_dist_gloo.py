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


def _loop_worker(rank, world, port, inflight, n_steps, q):
    """bench.py's serving loop (neupan_amd.serve.run_steps + ControlGatherer) with stand-in planners: `inflight` slots, each
    step's controls a known function of (rank, slot, how often the slot ran)."""
    import torch.distributed as dist
    from neupan_amd.serve import ControlGatherer, run_steps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T = 4, 10
    runs = [0] * inflight

    def make(j):
        def step():
            runs[j] += 1
            return {"opt_u": torch.full((B, 2, T), float(1000 * rank + 10 * j + runs[j]))}
        return step
    g = ControlGatherer(dist, world, device=None, slots=inflight)
    last = run_steps(n_steps, [make(j) for j in range(inflight)], None, g, None)
    rows = []
    for j, item in enumerate(last):
        if item is None:
            rows.append(None)
            continue
        o, gathered = item
        rows.append((tuple(gathered.shape), [float(gathered[r * B, 0, 0]) for r in range(world)]))
    q.put((rank, rows, g.issued))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("inflight,n_steps", [(1, 3), (4, 10), (5, 3)])
def test_serving_loop_schedule_with_two_ranks(inflight, n_steps):
    """The schedule bench.py times, two ranks over gloo: one gather per step, issued in step order on every rank (no
    deadlock with several slots in flight, also when fewer steps than slots run), every rank sees every rank's controls
    of the SAME step in rank order."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, inflight, n_steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, rows, issued = q.get(timeout=120)
        got[rank] = rows
        assert issued == n_steps
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    for j, row in enumerate(got[0]):
        ran = len(range(j, n_steps, inflight))           # how often slot j ran
        if ran == 0:
            assert row is None
            continue
        shape, firsts = row
        assert shape == (2 * 4, 2, 10)
        assert firsts == [float(1000 * r + 10 * j + ran) for r in range(2)]


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
        for i in range(7):                       # 2 full groups + 1 trailing step
            j = i % slots
            with torch.cuda.stream(streams[j]):
                outs[j].fill_(float(100 * i + j))
            g.after_step(outs[j], j, streams[j])
            views[j] = g.gather(outs[j], j, streams[j])
        g.join()
        torch.cuda.synchronize()
        assert g.issued == 7 and g.collectives == 3
        want = [600.0, 401.0, 502.0]             # the last step of every slot
        for j in range(slots):
            assert tuple(views[j].shape) == (1, B, 2, T)
            assert torch.all(views[j] == want[j]), (j, float(views[j].flatten()[0]))
        # a step whose issuer did not stage it is staged by gather() itself, behind its producer
        outs[1].fill_(7.0)
        v = g.gather(outs[1], 1, torch.cuda.current_stream(dev))
        g.join(); torch.cuda.synchronize()
        assert torch.all(v == 7.0)
    finally:
        dist.destroy_process_group()
