"""The serving loop bench.py times, as a library: `inflight` independent batches kept in flight (one stream each), their
launches issued from a few host threads, and the controls of every step all-gathered across the ranks in COALESCED
collectives on one communication stream.

Protocol of the gather (ControlGatherer), identical on every rank and on every device type:

  * steps carry a global index i = 0, 1, 2, ... (the same on all ranks: every rank runs the same number of steps);
    group g = i // slots, row r = i % slots, buffer parity p = g & 1;
  * stage(i, opt_u, stream): whoever issued step i copies its controls, behind the step on the step's own stream, into
    row r of staging buffer p -- after the flush of group g - 2 (the previous reader of that row) has been issued (host:
    a condition variable, no spinning) and has completed (device: an event wait on the step's stream);
  * collect(i): ONE thread, in step order: the communication stream waits for row r's copy; when the group is complete
    one all-gather moves staging buffer p into result buffer p (world x slots rows).  The collectives are therefore
    issued in the same order on every rank whatever the issuing threads' relative progress;
  * join(): flushes a trailing partial group (every rank has the same one) and makes the current stream wait for the
    communication stream.  The next run starts on a fresh group.

A result row (world, B, 2, T) returned by collect() is valid once its group's collective has run (join() guarantees it
for everything issued) and is overwritten two groups later.  Every step's controls cross the fabric inside the loop, in
1/slots as many collectives as steps: a collective per step from 20 chains cost 25 % of the throughput even with ONE
rank (504 k vs 668 k plans/s, round 3).

Device-agnostic on purpose: with device=None the same protocol -- staging rows, parity alternation, flush order,
trailing group, worker threads -- runs on CPU tensors over gloo; tests/test_dist_gloo.py drives it with two ranks, several
issuing threads per rank, fewer steps than slots and uneven progress between the ranks.
"""
from __future__ import annotations

import threading

import torch


class _NullEvent:
    def record(self, stream=None):
        pass


class ControlGatherer:
    """Coalesced all-gather of the steps' controls (module docstring).  dist=None (or an uninitialised process group):
    inactive -- collect() hands the step's own controls back and nothing is staged."""

    def __init__(self, dist=None, world: int = 1, device=None, slots: int = 1, shape=None):
        self.dist, self.world, self.slots = dist, int(world), max(1, int(slots))
        self.active = dist is not None and dist.is_initialized() and shape is not None
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.issued = 0              # steps handed to collect()
        self.collectives = 0         # collectives actually issued
        self._next = 0               # first index of the next run (a multiple of slots)
        if not self.active:
            return
        self.comm = torch.cuda.Stream(device=self.device) if self.cuda else None
        shape = tuple(shape)
        kw = dict(dtype=torch.float32, device=self.device)
        self.stage_buf = [torch.zeros((self.slots,) + shape, **kw) for _ in range(2)]
        self.out = [torch.zeros((self.world, self.slots) + shape, **kw) for _ in range(2)]
        mk = (lambda: torch.cuda.Event()) if self.cuda else (lambda: _NullEvent())
        self._copied = [[mk() for _ in range(self.slots)] for _ in range(2)]
        self._flush_ev = [None, None]            # event behind the last all-gather that read stage_buf[parity]
        self._cv = threading.Condition()
        self._staged_group = [[-1] * self.slots for _ in range(2)]   # group whose controls row (parity, r) holds
        self._flushed = 0                        # groups flushed so far (groups flush in order)
        self._pending = 0                        # rows of the current group handed to collect()
        self._cur_group = 0
        self._gloo = dist.get_backend() == "gloo"
        self._last_waited = None

    # ------------------------------------------------------------------ indices
    def begin(self, n: int) -> int:
        """Reserve the indices of a run of n steps: returns the index of its first step (group aligned)."""
        base = self._next
        self._next = base + (n + self.slots - 1) // self.slots * self.slots
        return base

    # ------------------------------------------------------------------ issuing threads
    def stage(self, i: int, opt_u: torch.Tensor, stream=None):
        """Stage the controls of step i behind the step (call on the thread that issued it; stream = the step's stream,
        default the current one)."""
        if not self.active:
            return
        g, r = divmod(i, self.slots)
        p = g & 1
        with self._cv:
            # the row's previous occupant (group g - 2) must have been handed to its all-gather
            self._cv.wait_for(lambda: self._flushed >= g - 1)
            fl = self._flush_ev[p] if g >= 2 else None
        if self.cuda:
            st = stream if stream is not None else torch.cuda.current_stream(self.device)
            with torch.cuda.stream(st):
                if fl is not None:
                    st.wait_event(fl)
                self.stage_buf[p][r].copy_(opt_u, non_blocking=True)
                self._copied[p][r].record(st)
        else:
            self.stage_buf[p][r].copy_(opt_u)
        with self._cv:
            self._staged_group[p][r] = g
            self._cv.notify_all()

    def stage_many(self, idxs, opts, stream=None):
        """stage() for several steps that ran on ONE stream (the steps of a merged launch chain finish together: their last
        launch is the same): one stream switch, one fused copy (torch._foreach_copy_), one event for all their rows -- instead
        of that per step.  (Measured, one rank under torch.distributed.run on the driver's 20-step region: the per-step form cost
        8 % of the rate -- all of it host time on the issuing threads, `host_issue_ms_per_step` 0.05 -> 0.12 -- the communicator
        itself nothing, profiles/r06_torchrun_overhead.txt.)"""
        if not self.active or not idxs:
            return
        if not self.cuda or not hasattr(torch, "_foreach_copy_"):
            for i, o in zip(idxs, opts):
                self.stage(i, o, stream)
            return
        rows = [divmod(i, self.slots) for i in idxs]
        gmax = max(g for g, _ in rows)
        with self._cv:
            self._cv.wait_for(lambda: self._flushed >= gmax - 1)
            fls = {self._flush_ev[g & 1] for g, _ in rows if g >= 2}
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(st):
            for fl in fls:
                if fl is not None:
                    st.wait_event(fl)
            torch._foreach_copy_([self.stage_buf[g & 1][r] for g, r in rows], list(opts), non_blocking=True)
            ev = self._copied[rows[0][0] & 1][rows[0][1]]
            ev.record(st)
        with self._cv:
            for g, r in rows:
                self._copied[g & 1][r] = ev
                self._staged_group[g & 1][r] = g
            self._cv.notify_all()

    # ------------------------------------------------------------------ the collecting thread (one, in step order)
    def collect(self, i: int, opt_u: torch.Tensor | None = None):
        """Hand step i to the gather.  Returns the gathered controls of that step, a (world, B, 2, T) view of the result
        buffer (inactive gatherer: opt_u itself)."""
        self.issued += 1
        if not self.active:
            return opt_u
        g, r = divmod(i, self.slots)
        p = g & 1
        if g != self._cur_group:                 # (a new run after join(): groups are consecutive, rows start at 0)
            assert self._pending == 0 and g == self._flushed, "collect() out of order"
            self._cur_group = g
        with self._cv:
            self._cv.wait_for(lambda: self._staged_group[p][r] == g)
        if self.cuda:
            ev = self._copied[p][r]
            if ev is not self._last_waited:          # (rows staged by one stage_many share their event)
                self.comm.wait_event(ev)
                self._last_waited = ev
        self._pending += 1
        if r == self.slots - 1:
            self._flush(g)
        return self.out[p][:, r]

    def _flush(self, g):
        p = g & 1
        ev = None
        if self.cuda:
            with torch.cuda.stream(self.comm):
                self.dist.all_gather_into_tensor(self.out[p].view(-1), self.stage_buf[p].view(-1))
                ev = torch.cuda.Event()
                ev.record(self.comm)
        elif self._gloo:
            self.dist.all_gather([self.out[p][w] for w in range(self.world)], self.stage_buf[p])
        else:
            self.dist.all_gather_into_tensor(self.out[p].view(-1), self.stage_buf[p].view(-1))
        self.collectives += 1
        self._pending = 0
        with self._cv:
            self._flush_ev[p] = ev
            self._flushed = g + 1
            self._cur_group = g + 1
            self._cv.notify_all()

    def join(self, stream=None):
        """Flush a trailing partial group; the current stream (or `stream`) then waits for every collective issued."""
        if not self.active:
            return
        if self._pending > 0:
            self._flush(self._cur_group)
        if self.cuda:
            (stream if stream is not None else torch.cuda.current_stream(self.device)).wait_stream(self.comm)


def run_steps(n: int, steps, streams=None, gatherer: ControlGatherer | None = None, cur=None):
    """Issue n steps round-robin over `steps` (callables returning the output dict of a forward call, e.g.
    PAN.make_step(...)), step i on streams[i % len(steps)], its controls handed to `gatherer` behind it.
    Returns the last (out, gathered) of every slot.  Nothing synchronises the host."""
    nfl = len(steps)
    last = [None] * nfl
    base = gatherer.begin(n) if gatherer is not None else 0
    if streams is not None:
        for st in streams:
            st.wait_stream(cur)
    for i in range(n):
        j = i % nfl
        if streams is not None:
            with torch.cuda.stream(streams[j]):
                o = steps[j]()
            if gatherer is not None:
                gatherer.stage(base + i, o["opt_u"], streams[j])
        else:
            o = steps[j]()
            if gatherer is not None:
                gatherer.stage(base + i, o["opt_u"])
        g = gatherer.collect(base + i, o["opt_u"]) if gatherer is not None else o["opt_u"]
        last[j] = (o, g)
    if streams is not None:
        for st in streams:
            cur.wait_stream(st)
    if gatherer is not None:
        gatherer.join(cur)
    return last


class StepLoop:
    """run_steps with the launches issued from several host threads: slot j (one planner, one stream) belongs to thread
    j % threads, which issues that slot's steps in order (a handle plans one batch at a time) and stages their controls;
    the calling thread hands every step to the gatherer in step order.  One Python thread needs ~5 us per kernel launch
    -- 0.11 ms for the 21 launches of a step -- which is what a short timed region mostly measures; the library call
    releases the GIL, so the launches of different planners proceed in parallel.  threads = 0: plain run_steps.
    streams=None (CPU stand-in planners, tests): the same threading without streams."""

    def __init__(self, steps, streams, gatherer=None, cur=None, threads=0, burst=False, group_cls=None):
        import queue
        self.steps, self.streams, self.gatherer, self.cur = steps, streams, gatherer, cur
        self.nfl = len(steps)
        self.threads = min(int(threads), self.nfl)
        # burst: the steps a thread issues in one round of the slots go out as ONE breadth-first library call
        # (neupan_amd.pan.StepGroup -> npa_forward_batch_group): every chain of the round starts within the first two launches
        # per chain instead of behind the 21 launches of each chain in front of it.  Steps that cannot be grouped (HIP-graph
        # steps, stand-ins without a group class) keep the call-by-call order.
        self.groups, self.burst_refused = None, None
        if burst:
            if group_cls is None:
                from .pan import StepGroup as group_cls
            from ._lib import NeupanAmdError
            nw = max(self.threads, 1)
            # a thread's members in CHAIN-MAJOR order (the slots of one stream next to each other, streams in order of first
            # appearance): npa_forward_batch_group merges runs of consecutive members that share a stream, and interleaves the
            # runs breadth-first -- so ONE call from ONE thread starts every chain of the round within its first launches
            # (threads = 0: all chains; no hand-over of the interpreter lock between issuing threads in front of the launches)
            self._members, self._pos = [], []
            for w in range(nw):
                mem = [j for j in range(self.nfl) if j % nw == w]
                if streams is not None:
                    first = {}
                    for j in mem:
                        first.setdefault(streams[j], len(first))
                    mem.sort(key=lambda j: (first[streams[j]], j))
                self._members.append(mem)
                self._pos.append({j: k for k, j in enumerate(mem)})
            try:
                self.groups = [group_cls([steps[j] for j in self._members[w]],
                                         [streams[j] for j in self._members[w]] if streams is not None else None)
                               for w in range(nw)]
            except NeupanAmdError as e:          # members that cannot be grouped (HIP-graph steps, mixed flags): call by call
                self.groups, self.burst_refused = None, str(e)
        self._q = [queue.Queue() for _ in range(self.threads)]
        self._workers = []
        self._err = None
        for w in range(self.threads):
            t = threading.Thread(target=self._work, args=(w,), daemon=True)
            t.start()
            self._workers.append(t)

    def _issue(self, i, base):
        j = i % self.nfl
        if self.streams is not None:
            with torch.cuda.stream(self.streams[j]):
                o = self.steps[j]()
            if self.gatherer is not None:
                self.gatherer.stage(base + i, o["opt_u"], self.streams[j])
        else:
            o = self.steps[j]()
            if self.gatherer is not None:
                self.gatherer.stage(base + i, o["opt_u"])
        return o

    def _issue_round(self, w, nw, r0, n, base, outs, evs=None):
        """Thread w's slots of the round that starts at step r0 (a prefix of its slot list) as ONE breadth-first group call;
        their controls are staged behind them in step order."""
        mine = [i for i in range(r0, min(n, r0 + self.nfl)) if (i % self.nfl) % nw == w]
        if not mine:
            return
        pos = self._pos[w]
        mine.sort(key=lambda i: pos[i % self.nfl])                    # the group's (chain-major) order
        idx = [pos[i % self.nfl] for i in mine]
        if idx == list(range(len(idx))):
            res = self.groups[w].issue(len(idx))
        else:                                                         # (a partial round: not a prefix of the member list)
            res = self.groups[w].issue_members(idx)
        if self.gatherer is not None:
            if self.streams is not None and hasattr(self.gatherer, "stage_many"):
                # a merged chain's steps finish together -- their controls are staged together, one fused copy per stream
                by = {}
                for i, o in zip(mine, res):
                    by.setdefault(self.streams[i % self.nfl], []).append((i, o))
                for st, lst in by.items():
                    lst.sort(key=lambda x: x[0])
                    if len(lst) > 1:
                        self.gatherer.stage_many([base + i for i, _ in lst], [o["opt_u"] for _, o in lst], st)
                    else:
                        self.gatherer.stage(base + lst[0][0], lst[0][1]["opt_u"], st)
            else:
                for i, o in sorted(zip(mine, res), key=lambda x: x[0]):
                    if self.streams is not None:
                        self.gatherer.stage(base + i, o["opt_u"], self.streams[i % self.nfl])
                    else:
                        self.gatherer.stage(base + i, o["opt_u"])
        for i, o in zip(mine, res):
            outs[i] = o
            if evs is not None:
                evs[i].set()

    def _work(self, w):
        while True:
            cmd = self._q[w].get()
            if cmd is None:
                return
            n, base, outs, evs = cmd
            try:
                if self.groups is not None:
                    for r0 in range(0, n, self.nfl):
                        self._issue_round(w, self.threads, r0, n, base, outs, evs)
                    continue
                for i in range(n):
                    if (i % self.nfl) % self.threads != w:
                        continue
                    outs[i] = self._issue(i, base)
                    evs[i].set()
            except BaseException as e:      # surface it in run(); unblock the main thread
                self._err = e
                for ev in evs:
                    ev.set()

    def run(self, n):
        if self.threads == 0 and self.groups is None:
            return run_steps(n, self.steps, self.streams, self.gatherer, self.cur)
        if self.threads == 0:                       # the calling thread alone, round by round
            last = [None] * self.nfl
            base = self.gatherer.begin(n) if self.gatherer is not None else 0
            if self.streams is not None:
                for st in self.streams:
                    st.wait_stream(self.cur)
            outs = [None] * n
            for r0 in range(0, n, self.nfl):
                self._issue_round(0, 1, r0, n, base, outs)
                for i in range(r0, min(n, r0 + self.nfl)):
                    o = outs[i]
                    g = self.gatherer.collect(base + i, o["opt_u"]) if self.gatherer is not None else o["opt_u"]
                    last[i % self.nfl] = (o, g)
            if self.streams is not None:
                for st in self.streams:
                    self.cur.wait_stream(st)
            if self.gatherer is not None:
                self.gatherer.join(self.cur)
            return last
        last = [None] * self.nfl
        base = self.gatherer.begin(n) if self.gatherer is not None else 0
        if self.streams is not None:
            for st in self.streams:
                st.wait_stream(self.cur)
        outs, evs = [None] * n, [threading.Event() for _ in range(n)]
        for q in self._q:
            q.put((n, base, outs, evs))
        for i in range(n):
            evs[i].wait()
            if self._err is not None:
                raise self._err
            j = i % self.nfl
            o = outs[i]
            g = self.gatherer.collect(base + i, o["opt_u"]) if self.gatherer is not None else o["opt_u"]
            last[j] = (o, g)
        if self.streams is not None:
            for st in self.streams:
                self.cur.wait_stream(st)
        if self.gatherer is not None:
            self.gatherer.join(self.cur)
        return last

    def close(self):
        for q in self._q:
            q.put(None)
        for t in self._workers:
            t.join(timeout=5)
        self._workers = []


def bind_to_gpu_numa_node(device_index: int):
    """Pin this process to the CPUs of the NUMA node its GPU hangs off (one process per GPU: the launch thread and the
    pinned buffers should not sit across the socket).  Best effort: returns the node or None."""
    import os
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return None
        cpus = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.extend(range(int(lo), int(hi or lo) + 1))
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        return None
    return None
