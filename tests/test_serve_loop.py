"""neupan_amd.serve.StepLoop's grouping on the host (no GPU): the members of a group call are kept in chain-major order, so that
npa_forward_batch_group -- which merges runs of CONSECUTIVE members sharing a stream and interleaves the runs breadth-first -- starts
every chain of a round from one call; a partial round selects its members by position (StepGroup.issue_members)."""
import pytest

from neupan_amd.serve import StepLoop


class _Stream:
    def wait_stream(self, other):
        pass


class _Group:
    """stand-in of neupan_amd.pan.StepGroup: records how it was built and called"""
    built, calls = [], []

    def __init__(self, steps, streams):
        self.steps, self.streams = steps, streams
        _Group.built.append(([s.slot for s in steps], list(streams)))

    def issue(self, n=None):
        n = len(self.steps) if n is None else n
        _Group.calls.append(("issue", n))
        return [s() for s in self.steps[:n]]

    def issue_members(self, idx):
        _Group.calls.append(("issue_members", tuple(idx)))
        return [self.steps[i]() for i in idx]


def _steps(n):
    def make(j):
        def step():
            step.ran += 1
            return {"opt_u": (j, step.ran), "slot": j}
        step.ran, step.slot = 0, j
        return step
    return [make(j) for j in range(n)]


@pytest.mark.parametrize("threads", [0, 1, 2, 4])
def test_group_members_are_chain_major_and_partial_rounds_select_by_position(threads):
    _Group.built, _Group.calls = [], []
    nfl, chains = 8, 4
    pool = [_Stream() for _ in range(chains)]
    streams = [pool[j % chains] for j in range(nfl)]
    steps = _steps(nfl)
    loop = StepLoop(steps, streams, None, _Stream(), threads=threads, burst=True, group_cls=_Group)
    try:
        nw = max(threads, 1)
        assert len(_Group.built) == nw
        for w, (slots, sts) in enumerate(_Group.built):
            assert sorted(slots) == [j for j in range(nfl) if j % nw == w]
            # the slots of one stream stand next to each other, in slot order
            runs = []
            for j, st in zip(slots, sts):
                assert st is streams[j]
                if runs and runs[-1][0] is st:
                    runs[-1][1].append(j)
                else:
                    runs.append((st, [j]))
            assert len({id(st) for st, _ in runs}) == len(runs)
            assert all(r == sorted(r) for _, r in runs)
        if threads == 0:
            assert _Group.built[0][0] == [0, 4, 1, 5, 2, 6, 3, 7]
        # a whole round: the prefix form, one call per thread; every slot's own result comes back under its own index
        last = loop.run(nfl)
        assert sorted(c for c in _Group.calls) == [("issue", nfl // nw)] * nw
        assert [o["slot"] for o, _ in last] == list(range(nfl)) and all(g == (j, 1) for j, (o, g) in enumerate(last))
        # a partial round (3 steps: slots 0, 1, 2): selected by position in the member list
        _Group.calls = []
        last = loop.run(3)
        assert [steps[j].ran for j in range(nfl)] == [2, 2, 2, 1, 1, 1, 1, 1]
        assert [last[j][0]["slot"] for j in range(3)] == [0, 1, 2] and last[3] is None
        if threads == 0:
            assert _Group.calls == [("issue_members", (0, 2, 4))]
        # two rounds in one run: 11 steps = a whole round and slots 0..2 again
        _Group.calls = []
        loop.run(nfl + 3)
        assert [steps[j].ran for j in range(nfl)] == [4, 4, 4, 2, 2, 2, 2, 2]
    finally:
        loop.close()
