"""Batches of streams through the whole path as a software pipeline over HIP streams, with a DEFERRED guard.

The reference's streamer runs the transmitter (encode + quantize) and the receiver (lookup + decode) in two threads
joined by a queue (/root/reference/bin/stream.py:212-239).  ``StreamingPipeline`` is that split for a batch of
independent streams on one GPU: the transmitter on one HIP stream, the receiver -- lookup + the vocoder, the vocoder
possibly lowered as several programs (``HiFiGANStreamGenerator.set_stages``) -- on one HIP stream per program, batches
handed over by events.  The encoder works on batch i+2 while the first vocoder program decodes batch i+1 and the second
one batch i.  Every batch still goes through the whole path; nothing is skipped or reordered per stream.

The guard.  The product arithmetic is split-f16 (stream_generator.py); its one failure mode -- an operand beyond the
f16 range -- is caught on the device in a sticky word of the program that met it.  A synchronous caller has the word
checked after every program step (``_StreamBase._step``: one stream synchronisation per program and step); three
programs of three batches in flight cannot afford that.  Here the check is DEFERRED: behind every program step one
1-thread kernel moves the program's word into a pinned host word and an event is recorded (``adk_program_flags_post``);
nothing waits.  At the entry of a later ``step()`` the posts that have completed by then are read (``hipEventQuery``),
oldest batch first; only when ``depth`` batches are unverified does the host wait for the oldest.  A batch whose posts
all read 0 is final.  When a post reports an overflow, everything in flight is drained, every program is taken back to
where it stood before the first bad batch (``adk_program_rewind``, once per step taken since -- the rings carry
``rewind_depth`` steps of extra rows for exactly that: the history the bad step read is still there), the program that
overflowed is replaced by its exact-f32 twin (``HipProgram.demote``), and the batches are run again, in order, into the
tensors the caller already holds.  The encoder's repeat does not read the caller's input again: the rows are still in
its input ring (``ADK_STEP_REPLAY``).  Exact, for the reason the synchronous repair is: a step only reads rows earlier
steps wrote.  A result is FINAL once ``exit()`` / ``settle()`` has returned or ``depth`` further steps were taken --
which is when a caller may let it leave the device.

``guard=False`` generators run through the same schedule unchecked (failures surface at the caller's next
``native.raise_on_device_flags``), as bench.py's `unguarded` leg does.
"""
import collections
import time
import warnings

import torch

from . import lazy_guard, native

_STREAM_POOL = {}


def _pool_streams(dev, n, priorities):
    """Every pipeline object of a process uses the SAME HIP streams per (device, priorities): the runtime deals streams over
    4 hardware queues in creation order, so a second object with fresh streams may find two of its three sharing a queue
    (profiles/r4_few_streams.md section 5).  Objects of one key must not be stepped concurrently."""
    key = (str(dev), tuple(priorities[:n]))
    if key not in _STREAM_POOL:
        _STREAM_POOL[key] = [torch.cuda.Stream(dev, priority=priorities[i]) for i in range(n)]
    return _STREAM_POOL[key]


class _Batch:
    """One step() in flight: the tensors handed out and the program steps taken for it."""
    __slots__ = ("x", "z", "idx", "y", "steps", "units")

    def __init__(self, units=1):
        self.x = self.z = self.idx = self.y = None
        self.steps = []          # (program, frames, ticket), in issue order
        self.units = units       # frames per stream of this batch: what a rewind of it takes out of the rings' extra rows


class GuardLog:
    """Bookkeeping of the deferred guard, free of torch / HIP: which batches are unverified, when to wait, what to redo.

    `poll(program, ticket, block) -> (done, flags)`; `repair(batches, flags_by_program, culprit)` is called with the unverified
    batches from the first bad one on (oldest first) and must leave them correct.  `culprit` = the first program, in issue order,
    that reported in the first bad batch: programs behind it (and the same programs in later batches) may only have seen its garbage."""

    def __init__(self, depth, poll, repair, drain, budget=None):
        """depth: most batches unverified at a time; budget: most UNITS (frames per stream, summed over the unverified batches) -- what the
        rings can be rewound by; None = no limit besides depth."""
        assert depth >= 1
        self.depth, self._poll, self._repair, self._drain = int(depth), poll, repair, drain
        self.budget = budget
        self.pending = collections.deque()
        self.verified = 0            # batches found clean (or repaired) so far
        self.repairs = 0
        self.waits = 0               # times the host had to wait for the oldest batch (log full)

    def push(self, batch):
        self.pending.append(batch)

    def units_pending(self):
        return sum(b.units for b in self.pending)

    def collect(self, block=False, incoming=0):
        """Retire the batches whose posts have completed; block=True -- or a log that is full, or that could not be rewound any more with the
        `incoming` units of the batch about to be issued on top -- waits for the oldest first."""
        while self.pending:
            must = block or len(self.pending) >= self.depth or (self.budget is not None and self.units_pending() + incoming > self.budget)
            b = self.pending[0]
            found, ready = [], True
            for prog, _frames, ticket in b.steps:
                done, fl = self._poll(prog, ticket, False)
                if not done and must:
                    self.waits += 0 if block else 1
                    done, fl = self._poll(prog, ticket, True)
                if not done:
                    ready = False
                    break
                if fl:
                    found.append((prog, fl))
            if found:
                # (a poll reads AND clears the program's word -- ABI 14 --, so what was found is handed on; the posts of the head batch that were
                # not reached are read by _recover.  A word may already hold what a LATER batch's step reported: the blame can only be early.)
                self._recover(found)
                return
            if not ready:
                return
            self.pending.popleft()
            self.verified += 1

    def _recover(self, found):
        """found: [(program, flags)] read from posts of the OLDEST unverified batch, in issue order: that batch is the first bad one."""
        self._drain()                                   # everything issued has finished: every post can be read
        batches = list(self.pending)
        by_prog, culprit = {}, found[0][0]
        for prog, fl in found:
            by_prog[prog] = by_prog.get(prog, 0) | fl
        for b in batches:                               # the rest of the words: read (and thereby cleared) too
            for prog, _frames, ticket in b.steps:
                _done, fl = self._poll(prog, ticket, True)
                if fl:
                    by_prog[prog] = by_prog.get(prog, 0) | fl
        self.pending.clear()
        self._repair(batches, by_prog, culprit)
        self.verified += len(batches)
        self.repairs += 1


class StreamingPipeline:
    def __init__(self, ad, device=None, depth=4, priorities=(0, 0, 0, 0), rvq_stream="tx"):
        """ad: an AudioDec (or anything with tx_encoder / rx_encoder / decoder stream generators on one device).
        depth: batches that may be unverified before step() waits for the oldest (deferred guard); capped by the generators'
        rewind_depth.  rvq_stream: the HIP stream the residual-VQ search of a batch is launched on -- "tx" (with the encoder, as the
        reference's transmitter thread does), "rx", "last" or "own" (measured: profiles/r4_few_streams.md section 5)."""
        self.ad = ad
        self.tx, self.rx, self.dec = ad.tx_encoder, ad.rx_encoder, ad.decoder
        self.dev = torch.device(device) if device is not None else self.tx._dev()
        for g_ in (self.tx, self.rx, self.dec):
            d_ = getattr(g_, "_device", None)
            if d_ is not None and torch.device(d_) != torch.device(self.tx._dev()):
                # (the reference's --tx_cuda / --rx_cuda split runs the two halves on two GPUs: that is two pipelines joined by the wire, not this object)
                raise native.NativeError("StreamingPipeline: transmitter and receiver must live on one HIP device")
        self.n_dec = getattr(self.dec, "stages", 1)
        pr = list(priorities) + [0, 0, 0, 0]
        pool = _pool_streams(self.dev, 1 + max(self.n_dec, 1), pr)
        self.s_tx, self.s_rx = pool[0], pool[1]
        self.s_more = pool[2:2 + self.n_dec - 1]
        self.s_own = [torch.cuda.Stream(self.dev)] if rvq_stream == "own" else []
        self.s_rvq = {"tx": self.s_tx, "rx": self.s_rx, "last": (self.s_more[-1] if self.s_more else self.s_rx),
                      "own": (self.s_own or [None])[0]}[rvq_stream]
        self.last_z = self.last_idx = None
        # the guard: deferred when both generators want one and their rings can be rewound; else whatever the generators do themselves
        gens = [g for g in (self.tx, self.dec) if g is not None]
        self.guarded = all(getattr(g, "guard", False) for g in gens)
        room = min([getattr(g, "rewind_depth", 0) for g in gens] + [int(depth), native.POST_SLOTS // 4])
        offline = any(getattr(g, "offline", False) for g in gens)
        self.deferred = self.guarded and room >= 1 and not offline
        self.depth = room if self.deferred else 0
        # what the rings can be rewound by, in frames per stream: (rewind_depth + 1) x max_frames (rows = hist + that many frames)
        self.budget = min([(getattr(g, "rewind_depth", 0) + 1) * getattr(g, "max_frames", 1) for g in gens]) if self.deferred else 0
        self.log = GuardLog(self.depth, self._poll, self._repair, self._drain, self.budget) if self.deferred else None
        # host-side accounting (seconds, since construction / reset_host_times()): t_issue = handing a batch's launches to the runtime,
        # t_guard = reading / waiting for the posts of older batches at the entry of step() -- what a rank's Python costs per step
        self.t_issue = self.t_guard = 0.0
        self.n_steps = 0

    # ---- stream plumbing ----
    def _all(self):
        return [self.s_tx, self.s_rx] + list(self.s_more) + self.s_own

    def enter(self):
        """All pipeline streams start after whatever ran on the current stream."""
        for g in (self.tx, self.rx, self.dec):               # direct calls made so far (the warm-up) are verified before the pipeline takes over
            if g is not None and hasattr(g, "settle"):
                g.settle()
        cur = torch.cuda.current_stream(self.dev)
        for s in self._all():
            s.wait_stream(cur)

    def exit(self):
        """The current stream waits for all pipeline streams; with the deferred guard every batch handed out so far is verified
        (and repaired if need be) when this returns."""
        cur = torch.cuda.current_stream(self.dev)
        for s in self._all():
            cur.wait_stream(s)
        self.settle()

    def settle(self):
        """Wait until every batch handed out so far is verified; results are final afterwards.  Call it (or exit()) before anything that
        changes the generators' state behind the pipeline's back -- reset_buffer(), reset_stream(), set_*() -- a repair of a batch that is
        still unverified would rewind across it."""
        if self.log is not None:
            self.log.collect(block=True)

    def __enter__(self):
        self.enter()
        return self

    def __exit__(self, *exc):
        self.exit()
        return False

    # ---- one batch ----
    def reset_host_times(self):
        self.t_issue = self.t_guard = 0.0
        self.n_steps = 0

    def step(self, x):
        """x (B, C, frames*hop) on the device -> y (B, out, frames*hop); returns as soon as the work is enqueued.

        x may come straight from the caller's current stream (the transmitter stream waits for it).  y is written by the LAST pipeline
        stream: consume it on the current stream only after exit() (which makes the current stream wait for the pipeline streams and
        settles the guard), or make your stream wait yourself and -- with the deferred guard -- treat y as final only after settle() /
        `depth` further steps."""
        t0 = time.perf_counter()
        self.n_steps += 1
        if self.log is None:
            y = self._issue(x, None)
            self.t_issue += time.perf_counter() - t0
            return y
        frames = -(-int(x.shape[-1]) // self.tx.hop)
        if frames > self.budget:
            # a batch longer than the rings can be rewound by (a whole utterance in one call): checked synchronously, step by step
            self.log.collect(block=True)
            y = self._issue(x, None)
            self.t_issue += time.perf_counter() - t0
            return y
        self.log.collect(incoming=frames)
        t1 = time.perf_counter()
        self.t_guard += t1 - t0
        b = _Batch(frames)
        self.tx._defer = self.rx._defer = self.dec._defer = b.steps      # (this object owns the guard of the batch: the generators' own call logs are bypassed)
        try:
            y = self._issue(x, b)
        finally:
            self.tx._defer = self.rx._defer = self.dec._defer = None
        b.x, b.y = x, y
        self.log.push(b)
        self.t_issue += time.perf_counter() - t1
        return y

    def _issue(self, x, b):
        # x may have been produced on the caller's current stream right before this call (an H2D copy, any torch op): the transmitter
        # stream reads it, so it waits for that stream -- free when the stream is idle -- and the caching allocator is told that s_tx
        # uses x (with the guard off nothing else keeps x alive until the encoder's ring write has read it)
        if isinstance(x, torch.Tensor):
            xp = lazy_guard.plain(x)
            if xp.is_cuda:
                self.s_tx.wait_stream(torch.cuda.current_stream(self.dev))
                xp.record_stream(self.s_tx)
        with torch.cuda.stream(self.s_tx):
            z = self.tx.encode(x)
            if self.s_rvq is self.s_tx:
                idx = self.tx.quantize(z)
            ev = torch.cuda.Event()
            ev.record(self.s_tx)
        if self.s_rvq is not self.s_tx:
            with torch.cuda.stream(self.s_rvq):
                self.s_rvq.wait_event(ev)
                lazy_guard.plain(z).record_stream(self.s_rvq)
                idx = self.tx.quantize(z)
                ev = torch.cuda.Event()
                ev.record(self.s_rvq)
        self.last_z, self.last_idx = z, idx                # for the parity checks (tests, bench.py self_check); no extra work
        if b is not None:
            b.z, b.idx = z, idx
        with torch.cuda.stream(self.s_rx):
            self.s_rx.wait_event(ev)
            lazy_guard.plain(idx).record_stream(self.s_rx)
            zq = self.rx.lookup(idx)
            if self.n_dec < 2:
                return self.dec.decode(zq)
            mid = self.dec.decode_stage(0, zq)
            ev = torch.cuda.Event()
            ev.record(self.s_rx)
        for i, st in enumerate(self.s_more, 1):
            with torch.cuda.stream(st):
                st.wait_event(ev)
                lazy_guard.plain(mid).record_stream(st)
                mid = self.dec.decode_stage(i, mid)
                ev = torch.cuda.Event()
                ev.record(st)
        return mid

    # ---- deferred guard: device side of GuardLog ----
    @staticmethod
    def _poll(prog, ticket, block):
        return prog.poll_flags(ticket, block)

    def _drain(self):
        torch.cuda.synchronize(self.dev)

    def _repair(self, batches, by_prog, culprit):
        """batches: the unverified ones from the first bad one on, everything drained.  Rewind every program by the steps it took
        for them, demote the program that overflowed FIRST (what the programs behind it reported -- and it and they in the batches
        after -- may be nothing but its garbage), run the batches again serially on the current stream into the tensors already
        handed out.  The generators' own synchronous guard checks every repeated step: a program that overflows on clean input
        too is demoted there, at the step that does it."""
        other = 0
        for fl in by_prog.values():
            other |= fl & ~native.FLAG_F16_OVERFLOW
        native.raise_for_flags(other, "StreamingPipeline")
        for b in reversed(batches):
            for prog, frames, _t in reversed(b.steps):
                prog.rewind(frames)
        if by_prog.get(culprit, 0) & native.FLAG_F16_OVERFLOW:
            if not culprit.split16 or culprit.twin_builder is None or culprit.demoted:
                raise native.NativeError("StreamingPipeline: a program without an exact-f32 twin reported an f16 range overflow")
            culprit.demote()
        gens = [g for g in (self.tx, self.rx, self.dec) if g is not None]
        modes = [getattr(g, "guard_mode", None) for g in gens]
        for g in gens:
            g.guard_mode = "sync"                                # the repeats are checked step by step, before the next one is issued
        try:
            self._repeat(batches)
        finally:
            for g, m_ in zip(gens, modes):
                g.guard_mode = m_
        torch.cuda.synchronize(self.dev)
        self.enter()                                         # the pipeline streams continue behind the repeats
        warnings.warn(f"StreamingPipeline: an operand left the f16 range (|v| > 65504) in a split-f16 conv; the last {len(batches)} batch(es) "
                      "were repeated with the exact-f32 kernels for the program(s) concerned, which continue on them", RuntimeWarning, stacklevel=4)

    def _repeat(self, batches):
        with torch.no_grad():
            for b in batches:
                self.tx._replay = True                       # the caller's rows are still in the encoder's input ring
                try:
                    z = self.tx.encode(b.x)
                finally:
                    self.tx._replay = False
                idx = self.tx.quantize(z)
                y = self.dec.decode(self.rx.lookup(idx))
                b.z.copy_(z)
                b.idx.copy_(idx)
                b.y.copy_(y.reshape(b.y.shape))
