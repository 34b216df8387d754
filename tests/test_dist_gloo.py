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
