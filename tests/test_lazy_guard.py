"""CPU: the guard of direct calls (audiodec_amd/lazy_guard.py) against fake programs and generators -- what a GuardedTensor does when it is
looked at, when a call waits, what a repair rewinds, demotes and repeats, and in which order.  (On the GPU:
tests/test_gpu_parity.py::test_split16_range_overflow_is_repaired_by_the_f32_kernels[lazy-*], tests/test_gpu_lazy_guard.py.)"""
import threading
import warnings

import pytest
import torch

from audiodec_amd import lazy_guard
from audiodec_amd.lazy_guard import CallLog, GuardedTensor


class FakeProgram:
    """Posts complete when the fake device clock has passed them (or somebody waits); a post's word is what `fail_at` says for that ticket."""

    def __init__(self, name, rewind_depth=4, max_frames=1, events=None):
        self.name, self.rewind_depth, self.max_frames = name, rewind_depth, max_frames
        self.tickets, self.completed, self.fail_at = 0, -1, {}
        self.blocked = 0
        self.split16, self.twin_builder, self.demoted = True, object(), False
        self.events = events if events is not None else []

    def post(self):
        t = self.tickets
        self.tickets += 1
        return t

    def poll_flags(self, ticket, block):
        if ticket > self.completed:
            if not block:
                return False, 0
            self.blocked += 1
            self.completed = ticket
        return True, self.fail_at.get(ticket, 0)

    def rewind(self, frames):
        self.events.append(("rewind", self.name, frames))

    def demote(self):
        self.events.append(("demote", self.name))
        self.split16, self.demoted = False, True


class FakeGen:
    """A generator with one program: call(x) = x + 1 `written by the kernels` into a fresh tensor; a repeat writes x + 1 again (the 'bad'
    first result is simulated by the test poking the output)."""

    def __init__(self, name, log, events, **kw):
        self.prog = FakeProgram(name, events=events, **kw)
        self.log, self.events = log, events
        self._defer = None
        self._replay = False

    def _impl(self, x):
        self.events.append(("run", self.prog.name, bool(self._replay), self._defer is None))
        if self._defer is not None:
            self._defer.append((self.prog, 1, self.prog.post()))
        return x + 1

    def call(self, x):
        return self.log.run(self, self._impl, (x,), [self.prog], 1)


def make(n_gens=1, **kw):
    events, drained = [], []
    log = CallLog(None, drain=lambda: drained.append(1))
    gens = [FakeGen(f"p{i}", log, events, **kw) for i in range(n_gens)]
    return log, gens, events, drained


def test_results_are_guarded_tensors_that_settle_the_log_when_looked_at():
    log, (g,), events, _ = make()
    x = torch.zeros(3)
    y = g.call(x)
    assert type(y) is GuardedTensor and isinstance(y, torch.Tensor)
    assert len(log.pending) == 1 and g.prog.blocked == 0            # nothing waited
    p = lazy_guard.plain(y)
    assert type(p) is torch.Tensor and len(log.pending) == 1      # plain() does not settle ...
    assert torch.equal(p, torch.ones(3)) and len(log.pending) == 1                                               # ... nor does work on the plain tensor
    s = y.sum()                                                      # any torch function on the guarded tensor does
    assert float(s) == 3.0 and type(s) is torch.Tensor
    assert len(log.pending) == 0 and g.prog.blocked == 1 and log.verified == 1
    _ = y.cpu(); _ = y.shape; _ = y[0]
    assert g.prog.blocked == 1                                       # nothing left to wait for


def test_a_result_handed_to_the_next_call_of_the_same_log_is_not_waited_for_but_another_logs_is():
    log, (g0, g1), events, _ = make(2)
    y = g1.call(g0.call(torch.zeros(2)))
    assert len(log.pending) == 2 and g0.prog.blocked == 0 and g1.prog.blocked == 0
    assert [c.replay for c in log.pending] == [True, False]         # the first call's inputs came from outside, the second read this log's result
    other, (h,), _, _ = make()
    z = h.call(y)                                                    # a guarded tensor of ANOTHER log: its log is settled first
    assert len(log.pending) == 0 and g0.prog.blocked == 1 and len(other.pending) == 1
    assert torch.equal(lazy_guard.plain(z), torch.full((2,), 3.0))


def test_a_call_waits_for_the_oldest_only_when_the_rings_could_not_be_rewound_any_further():
    log, (g,), events, _ = make(rewind_depth=2)                      # 3 hops may be unverified
    x = torch.zeros(1)
    for n in range(3):
        g.call(x)
    assert g.prog.blocked == 0 and len(log.pending) == 3
    g.call(x)                                                        # the fourth: the oldest must be verified first
    assert g.prog.blocked == 1 and len(log.pending) == 3 and log.waits == 1
    g.prog.completed = g.prog.tickets - 1                            # the device caught up: the next call retires everything without waiting
    g.call(x)
    assert g.prog.blocked == 1 and len(log.pending) == 1


def test_overflow_is_repaired_in_call_order_into_the_tensors_already_handed_out():
    log, (enc, dec), events, drained = make(2)
    x = torch.zeros(2)
    z0 = enc.call(x); y0 = dec.call(z0)                              # clean
    z1 = enc.call(x); y1 = dec.call(z1)                              # the encoder's step of this pair overflows
    z2 = enc.call(x); y2 = dec.call(z2)
    enc.prog.fail_at[1] = 8
    dec.prog.fail_at[1] = 8                                          # the decoder saw the encoder's garbage: reports too -- it is NOT the culprit
    for t in (z1, y1, z2, y2):
        lazy_guard.plain(t).fill_(float("nan"))                      # what the kernels left
    del events[:]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        v = float(y2.sum())                                          # the first look at ANY result
    assert any(issubclass(i.category, RuntimeWarning) for i in w)
    assert v == 4.0 and torch.equal(lazy_guard.plain(z1), torch.ones(2)) and torch.equal(lazy_guard.plain(y1), torch.full((2,), 2.0))
    assert torch.equal(lazy_guard.plain(y0), torch.full((2,), 2.0))
    assert log.repairs == 1 and log.verified == 6 and not log.pending and len(drained) == 2
    # newest first: dec(2), enc(2), dec(1), enc(1) rewound; the FIRST reporter demoted; then the four calls again, oldest first
    assert events[:4] == [("rewind", "p1", 1), ("rewind", "p0", 1), ("rewind", "p1", 1), ("rewind", "p0", 1)]
    assert events[4] == ("demote", "p0") and not dec.prog.demoted
    # (run, program, replay, synchronous): calls fed from outside repeat with ADK_STEP_REPLAY, calls that read this log's results do not;
    # all of them with the generators' synchronous guard (_defer is None)
    assert events[5:] == [("run", "p0", True, True), ("run", "p1", False, True), ("run", "p0", True, True), ("run", "p1", False, True)]


def test_flags_other_than_the_f16_overflow_raise():
    log, (g,), _, _ = make()
    y = g.call(torch.zeros(1))
    g.prog.fail_at[0] = 1                                            # a code index outside the codebook
    with pytest.raises(IndexError):
        y.cpu()
    assert not log.pending


def test_calls_longer_than_the_rings_can_be_rewound_by_run_synchronously():
    log, (g,), events, _ = make(rewind_depth=1)
    out = log.run(g, g._impl, (torch.zeros(1),), [g.prog], 5)       # 5 hops > (1 + 1) x 1
    assert type(out) is torch.Tensor and not log.pending and events[-1] == ("run", "p0", False, True)


def test_two_threads_share_a_log():
    log, (g0, g1), _, _ = make(2)
    outs = [[], []]

    def work(k, g):
        for _ in range(200):
            outs[k].append(float(g.call(torch.zeros(1)).sum()))
    ts = [threading.Thread(target=work, args=(k, g)) for k, g in enumerate((g0, g1))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert outs[0] == [1.0] * 200 and outs[1] == [1.0] * 200 and not log.pending and log.verified == 400
