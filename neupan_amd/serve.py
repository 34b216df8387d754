"""The serving loop bench.py times, as a library function: `inflight` independent batches kept in flight, one stream each,
the per-step all-gather of the controls issued from ONE dedicated communication stream.

Why one comm stream: collectives on one communicator execute in issue order on every rank.  Issued from the batches'
own streams (20 of them), each all-gather also waited for whatever that stream had queued, and every step became a
cross-rank rendezvous in the middle of a compute chain.  Here a step's gather waits for exactly one event -- that step's
last kernel -- on a stream that carries nothing else, and the ranks meet in the same order by construction (step i's
gather is the i-th collective everywhere).

Device-agnostic on purpose: with `streams=None` (CPU tensors, gloo) the same schedule runs synchronously, which is how
tests/test_dist_gloo.py drives it with two ranks.
"""
from __future__ import annotations

import torch

from .dist import gather_controls


class ControlGatherer:
    """All-gather of the steps' controls behind their work, on a stream of its own (or inline on CPU).

    On a GPU the gathers are COALESCED: behind every step its controls are copied (on the step's own stream, so the next
    step of that planner cannot overtake the copy) into that slot's row of a staging buffer [slots][B][2][T], and ONE
    all_gather_into_tensor per `slots` steps moves the buffer -- every step's controls still cross the fabric inside the
    loop, in 1/slots as many collectives (a collective per step from 20 chains cost 25 % of the throughput with ONE rank:
    504 k vs 668 k plans/s).  Two staging / result buffers alternate by group, and a row is rewritten only after the
    gather that read it two groups earlier has completed (an event wait that has practically always passed).
    after_step() is called by whoever issues the step (possibly a worker thread), gather() by one thread in step order:
    the collectives are issued in the same order on every rank."""

    def __init__(self, dist=None, world: int = 1, device=None, slots: int = 1, shape=None):
        self.dist, self.world, self.slots = dist, world, slots
        self.active = dist is not None and dist.is_initialized()
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.comm = torch.cuda.Stream(device=device) if (self.active and self.cuda) else None
        self.issued = 0              # gather() calls
        self.collectives = 0         # collectives actually issued
        self.coalesce = self.comm is not None and shape is not None
        if self.coalesce:
            self.stage = [torch.empty((slots,) + tuple(shape), dtype=torch.float32, device=device) for _ in range(2)]
            self.out = [torch.empty((world, slots) + tuple(shape), dtype=torch.float32, device=device) for _ in range(2)]
            self.copied = [[torch.cuda.Event() for _ in range(slots)] for _ in range(2)]
            self.flushed = [None, None]          # event behind the last all-gather that read stage[parity]
            self.group_of = [0] * slots          # how many steps each slot has staged
            self.gathered_of = [0] * slots
            self.pending = [0, 0]
            self.flush_gen = [0, 0]              # all-gathers issued per staging buffer
            self.gen_at = [[-1] * slots for _ in range(2)]   # flush_gen when a slot last contributed to that buffer
        elif self.comm is not None:
            self.events = [torch.cuda.Event() for _ in range(slots)]

    def after_step(self, opt_u: torch.Tensor, slot: int, stream=None):
        """Stage a finished step's controls (call right after issuing the step, on the thread that issued it; `stream` = the
        step's stream, default the current one)."""
        if not self.coalesce:
            return
        st = stream if stream is not None else torch.cuda.current_stream(opt_u.device)
        par = self.group_of[slot] & 1
        # (an issuing thread that runs ahead: the row's previous occupant -- two steps of this slot ago -- must have been
        # handed to the gather, or its flush event does not exist yet)
        import time
        while self.gathered_of[slot] < self.group_of[slot] - 1 or \
                (self.group_of[slot] >= 2 and self.flush_gen[par] <= self.gen_at[par][slot]):
            time.sleep(0)
        fl = self.flushed[par]
        with torch.cuda.stream(st):
            if fl is not None:
                st.wait_event(fl)                # the gather that read this row two groups ago
            self.stage[par][slot].copy_(opt_u, non_blocking=True)
            self.copied[par][slot].record(st)
        self.group_of[slot] += 1

    def gather(self, opt_u: torch.Tensor, slot: int = 0, producer=None):
        """Hand a step's controls to the gather (one thread, step order).  Returns the gathered controls of that step:
        (world x B, 2, T), or a (world, B, 2, T) view of the coalesced result -- valid once the slot's group has been gathered
        (join() makes everything valid on the current stream)."""
        if not self.active:
            return opt_u
        self.issued += 1
        if self.comm is None:
            self.collectives += 1
            return gather_controls(opt_u, self.dist, self.world, equal_shards=True)
        if not self.coalesce:
            ev = self.events[slot]
            ev.record(producer if producer is not None else torch.cuda.current_stream(opt_u.device))
            self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                out = gather_controls(opt_u, self.dist, self.world, equal_shards=True)
            opt_u.record_stream(self.comm)
            self.collectives += 1
            return out
        if self.group_of[slot] == self.gathered_of[slot]:     # the caller did not stage it: do it here, on the producer's stream
            self.after_step(opt_u, slot, producer)
        par = self.gathered_of[slot] & 1
        self.gathered_of[slot] += 1
        self.comm.wait_event(self.copied[par][slot])
        self.gen_at[par][slot] = self.flush_gen[par]
        self.pending[par] += 1
        if self.pending[par] >= self.slots:
            self._flush(par)
        return self.out[par][:, slot]

    def _flush(self, par):
        with torch.cuda.stream(self.comm):
            self.dist.all_gather_into_tensor(self.out[par].view(-1), self.stage[par].view(-1))
            ev = torch.cuda.Event()
            ev.record(self.comm)
        self.flushed[par] = ev
        self.flush_gen[par] += 1
        self.collectives += 1
        self.pending[par] = 0

    def join(self, stream=None):
        if self.comm is not None:
            if self.coalesce:
                for par in (0, 1):
                    if self.pending[par] > 0:
                        self._flush(par)
            (stream if stream is not None else torch.cuda.current_stream(self.comm.device)).wait_stream(self.comm)


def run_steps(n: int, steps, streams=None, gatherer: ControlGatherer | None = None, cur=None):
    """Issue n steps round-robin over `steps` (callables returning the output dict of a forward call, e.g.
    PAN.make_step(...)), step i on streams[i % len(steps)], its controls handed to `gatherer` behind it.
    Returns the last (out, gathered) of every slot.  Nothing synchronises the host."""
    nfl = len(steps)
    last = [None] * nfl
    if streams is not None:
        for st in streams:
            st.wait_stream(cur)
    for i in range(n):
        j = i % nfl
        if streams is not None:
            with torch.cuda.stream(streams[j]):
                o = steps[j]()
            if gatherer is not None:
                gatherer.after_step(o["opt_u"], j, streams[j])
            g = gatherer.gather(o["opt_u"], j, streams[j]) if gatherer is not None else o["opt_u"]
        else:
            o = steps[j]()
            g = gatherer.gather(o["opt_u"], j) if gatherer is not None else o["opt_u"]
        last[j] = (o, g)
    if streams is not None:
        for st in streams:
            cur.wait_stream(st)
        if gatherer is not None:
            gatherer.join(cur)
    return last


class StepLoop:
    """run_steps with the launches issued from several host threads: slot j (one planner, one stream) belongs to thread
    j % threads, which issues that slot's steps in order (a handle plans one batch at a time); the main thread hands every
    step's controls to the gatherer in step order.  One Python thread needs ~5 us per kernel launch -- 0.11 ms for the 21
    launches of a step -- which is what a short timed region mostly measures; the library call releases the GIL, so the
    launches of different planners proceed in parallel.  threads = 0: plain run_steps."""

    def __init__(self, steps, streams, gatherer=None, cur=None, threads=0):
        import queue
        import threading
        self.steps, self.streams, self.gatherer, self.cur = steps, streams, gatherer, cur
        self.nfl = len(steps)
        self.threads = min(int(threads), self.nfl) if streams is not None else 0
        self._q = [queue.Queue() for _ in range(self.threads)]
        self._workers = []
        self._err = None
        for w in range(self.threads):
            t = threading.Thread(target=self._work, args=(w,), daemon=True)
            t.start()
            self._workers.append(t)

    def _work(self, w):
        while True:
            cmd = self._q[w].get()
            if cmd is None:
                return
            n, outs, evs = cmd
            try:
                for i in range(n):
                    j = i % self.nfl
                    if j % self.threads != w:
                        continue
                    with torch.cuda.stream(self.streams[j]):
                        outs[i] = self.steps[j]()
                    if self.gatherer is not None:
                        self.gatherer.after_step(outs[i]["opt_u"], j, self.streams[j])
                    evs[i].set()
            except BaseException as e:      # surface it in run(); unblock the main thread
                self._err = e
                for ev in evs:
                    ev.set()

    def run(self, n):
        if self.threads == 0:
            return run_steps(n, self.steps, self.streams, self.gatherer, self.cur)
        import threading
        last = [None] * self.nfl
        for st in self.streams:
            st.wait_stream(self.cur)
        outs, evs = [None] * n, [threading.Event() for _ in range(n)]
        for q in self._q:
            q.put((n, outs, evs))
        for i in range(n):
            evs[i].wait()
            if self._err is not None:
                raise self._err
            j = i % self.nfl
            o = outs[i]
            g = self.gatherer.gather(o["opt_u"], j, self.streams[j]) if self.gatherer is not None else o["opt_u"]
            last[j] = (o, g)
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
