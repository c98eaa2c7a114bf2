"""The guard of DIRECT calls: checked when a result is first looked at, not when it is computed.

The reference's callers drive the model objects one call at a time -- ``encode``, ``quantize``, ``lookup``, ``decode``
(/root/reference/demoFile.py:58-61, utils/audiodec.py:100-106, bin/stream.py:212-239).  Until round 5 the guard of such calls
was synchronous: one 1-thread kernel + one stream synchronisation per program step before the next call could even be
enqueued (``_StreamBase._step``), i.e. the host's enqueue time of every call was exposed and three HIP streams could not
overlap -- 183 k frames/s against 271 k for ``pipeline.StreamingPipeline``, whose guard is deferred.

Here the same deferral for direct calls.  A generator's public call

  * posts the flag word of every program step behind the step (``adk_program_flags_post``: nothing waits),
  * records itself in a ``CallLog`` -- what it stepped, the tensors it read and wrote, how to run it again --,
  * returns its result as a ``GuardedTensor``: an ordinary ``torch.Tensor`` (same storage) whose every use through torch --
    ``.cpu()``, ``.to()``, ``.numpy()``, ``data_ptr()``, indexing, arithmetic, printing, ``record_stream`` ... -- first SETTLES
    the log: waits for the posts of everything recorded so far and, if one reports an f16 range overflow, repairs (below).
    A result can therefore not leave the device, or be looked at by anything but this library, unverified; ``.cpu()`` waits
    for the stream anyway, so the check costs a caller that reads its results nothing, and a caller that only hands them on
    to the next call of this library (``quantize(encode(x))``) nothing either: a guarded tensor of the SAME log is taken as
    it is, its verification is implied by the order of the log.  A guarded tensor of ANOTHER log (the codes of a
    transmitter arriving at a receiver on another device) settles its log first.
  * at its entry reads, without blocking, the posts that have completed (oldest first) and retires their calls; it waits for
    the oldest only when a program's rings could not be rewound by one more call (``rewind_depth``) or the log is full.

Repair (``CallLog._recover``), as ``StreamingPipeline._repair`` does for whole batches: the device is drained, every
program step of every unverified call from the first bad one on is rewound (newest first), the program that reported first
is demoted to its exact-f32 twin, and the calls are run again in order, each with the generators' synchronous guard, into the
tensors the callers already hold (later calls read the earlier calls' outputs through those very tensors).  A call whose
inputs all came from outside repeats with ``ADK_STEP_REPLAY`` (its input rows are still in the program's first ring: the
caller may have reused its input tensor); a call that read results of this log re-writes its input ring from the repeated
results.  One ``RuntimeWarning``, no exception.

One log is shared by the generators of an ``AudioDec`` object on one device (a lock makes it safe for the reference
streamer's transmitter / receiver threads); ``pipeline.StreamingPipeline`` bypasses it (it owns the guard of its batches).
``ADK_GUARD_MODE=sync`` / ``set_guard(True, mode="sync")`` keeps the synchronous check of rounds 3-5.
"""
import collections
import threading
import warnings

import torch

from . import native

MAX_PENDING = 64          # calls a log keeps unverified at most (entries without program steps -- quantize, lookup -- count too)


def plain(t):
    """The ordinary tensor behind a GuardedTensor (same storage), WITHOUT settling its log; anything else as it is."""
    if type(t) is GuardedTensor:
        p = t.__dict__.get("_adk_plain")
        if p is not None:
            return p
        with torch._C.DisableTorchFunctionSubclass():
            return t.as_subclass(torch.Tensor)
    return t


def log_of(t):
    return t.__dict__.get("_adk_log") if type(t) is GuardedTensor else None


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)


class GuardedTensor(torch.Tensor):
    """A result of a guarded direct call.  Shares the storage of the tensor the kernels wrote; any torch function applied to it
    settles the CallLog it belongs to first and then runs on the plain tensor (results are plain tensors)."""

    @staticmethod
    def wrap(t, log):
        g = torch.Tensor._make_subclass(GuardedTensor, t)
        g.__dict__["_adk_log"] = log
        g.__dict__["_adk_plain"] = t            # (what the kernels wrote into: handing a result on to the next call costs a dict lookup)
        return g

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        seen = set()
        for a in list(_tensors(args)) + list(_tensors(kwargs)):
            lg = log_of(a)
            if lg is not None and id(lg) not in seen:
                seen.add(id(lg))
                lg.settle()
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class _Call:
    __slots__ = ("gen", "impl", "args", "out", "steps", "replay")

    def __init__(self, gen, impl, args, out, steps, replay):
        self.gen, self.impl, self.args, self.out, self.steps, self.replay = gen, impl, args, out, steps, replay


class CallLog:
    """Unverified direct calls of the generators that share this log, oldest first (see the module docstring)."""

    def __init__(self, device=None, drain=None):
        """device: the HIP device of the generators that share the log; drain() (tests) replaces the device synchronisation of a repair."""
        self.device = device
        self._drain = drain if drain is not None else (lambda: torch.cuda.synchronize(device) if device is not None else torch.cuda.synchronize())
        self.lock = threading.RLock()
        self.pending = collections.deque()
        self._units = {}           # program -> hops of its steps in `pending` (what a repair would have to rewind)
        self.in_redo = False
        self.verified = 0          # calls found clean (or repaired)
        self.repairs = 0
        self.waits = 0             # times a call had to wait for the oldest one at its entry (rings / log full)

    # ---- bookkeeping ----
    # (this runs once per direct call on the host's issue path -- four times per frame for a single stream, where the host, not the device,
    # sets the pace: plain loops and a running count per program instead of generator expressions over the whole log)
    def _retire(self, c):
        units = self._units
        for prog, frames, _t in c.steps:
            units[prog] = units.get(prog, 0) - frames
        self.verified += 1

    def _units_of(self, prog):
        return self._units.get(prog, 0)

    def collect(self, block=False):
        """Retire the calls at the head of the log whose posts have completed; block=True waits for every one."""
        with self.lock:
            pending = self.pending
            while pending:
                c = pending[0]
                found = None
                for prog, _frames, ticket in c.steps:
                    done, fl = prog.poll_flags(ticket, block)
                    if not done:
                        if found:
                            break                            # (what was read is gone from the word: handle it now)
                        return False
                    if fl:
                        if found is None:
                            found = []
                        found.append((prog, fl))
                if found:
                    self._recover(found)
                    return True
                pending.popleft()
                self._retire(c)
            return True

    def settle(self):
        """Every call recorded so far is verified (and repaired if need be) when this returns."""
        if self.in_redo:
            return
        with self.lock:
            if self.pending:
                self.collect(block=True)

    def _full(self, progs, frames):
        if len(self.pending) >= MAX_PENDING:
            return True
        units = self._units
        for p in progs:
            if units.get(p, 0) + frames > (p.rewind_depth + 1) * p.max_frames:
                return True
        return False

    def make_room(self, progs, frames):
        """Before a call that steps each of `progs` by `frames` hops: retire what has completed; wait for the oldest call while a
        program's rings could not be rewound by this call on top of its unverified ones, or the log is full."""
        pending = self.pending
        if not pending:
            return
        self.collect(block=False)
        while pending and self._full(progs, frames):
            self.waits += 1
            c = pending[0]
            for prog, _f, ticket in c.steps:
                prog.poll_flags(ticket, True)
            if not c.steps:                                  # a call without program steps at the head: nothing to wait for
                pending.popleft()
                self._retire(c)
                continue
            self.collect(block=False)

    # ---- one guarded call ----
    def run(self, gen, impl, args, progs, frames, external_replay=True):
        """impl(*plain args) with the guard deferred; returns the result as a GuardedTensor of this log.  progs: the programs the call
        steps, frames: by how many hops (0: none -- quantize / lookup)."""
        with self.lock:
            own = False
            pargs = []
            for a in args:
                if type(a) is GuardedTensor:
                    d = a.__dict__
                    lg = d.get("_adk_log")
                    if lg is self:
                        own = True
                    elif lg is not None:
                        lg.settle()
                    pa = d.get("_adk_plain")
                    a = pa if pa is not None else plain(a)
                pargs.append(a)
            if progs:
                for p in progs:
                    if frames > (p.rewind_depth + 1) * p.max_frames:
                        # longer than the rings can be rewound by (a whole utterance in one call): checked synchronously, step by step
                        self.settle()
                        return impl(*pargs)
            self.make_room(progs, frames)
            steps = []
            gen._defer = steps
            try:
                out = impl(*pargs)
            finally:
                gen._defer = None
            if steps:
                units = self._units
                for prog, f, _t in steps:
                    units[prog] = units.get(prog, 0) + f
            self.pending.append(_Call(gen, impl, pargs, out, steps, external_replay and not own))
            return GuardedTensor.wrap(out, self)

    # ---- repair ----
    def _recover(self, found):
        """found: [(program, flags)] read from posts of the OLDEST unverified call, in issue order: that call is the first bad one.  (A poll
        reads AND clears the program's word -- ABI 14 --, so what collect() read is handed on; a word may already hold what a later call's
        step reported: the blame can only be early, and everything from the blamed call on is repeated.)"""
        self._drain()                                        # everything issued has finished: every post can be read
        calls = list(self.pending)
        first, culprit, by_prog = 0, found[0][0], {}
        for prog, fl in found:
            by_prog[prog] = by_prog.get(prog, 0) | fl
        for c in calls:                                      # the rest of the words: read (and thereby cleared) too
            for prog, _f, ticket in c.steps:
                _done, fl = prog.poll_flags(ticket, True)
                if fl:
                    by_prog[prog] = by_prog.get(prog, 0) | fl
        self.pending.clear()
        self._units.clear()
        other = 0
        for fl in by_prog.values():
            other |= fl & ~native.FLAG_F16_OVERFLOW
        native.raise_for_flags(other, "guarded call")
        if not (by_prog.get(culprit, 0) & native.FLAG_F16_OVERFLOW):
            self.verified += len(calls)
            return
        redo = calls[first:]
        for c in reversed(redo):
            for prog, frames, _t in reversed(c.steps):
                prog.rewind(frames)
        if not culprit.split16 or culprit.twin_builder is None or culprit.demoted:
            raise native.NativeError("guarded call: a program without an exact-f32 twin reported an f16 range overflow")
        culprit.demote()
        self.in_redo = True
        try:
            with torch.no_grad():
                for c in redo:
                    c.gen._replay = bool(c.replay)
                    try:
                        new = c.impl(*c.args)          # gen._defer is None and the log is in redo: the generators' synchronous guard checks every step
                    finally:
                        c.gen._replay = False
                    if new.data_ptr() != c.out.data_ptr():
                        c.out.copy_(new.reshape(c.out.shape))
            self._drain()
        finally:
            self.in_redo = False
        self.verified += len(redo)
        self.repairs += 1
        warnings.warn(f"an operand left the f16 range (|v| > 65504) in a split-f16 conv; the last {len(redo)} call(s) were repeated with the exact-f32 "
                      "kernels for the program concerned, which continues on them", RuntimeWarning, stacklevel=4)
