"""CPU: the bookkeeping of the deferred guard (audiodec_amd/pipeline.py: GuardLog) against fake programs -- which batches are
retired, when the host waits, what is handed to the repair -- and the host-side contract of StreamingPipeline that needs no device."""
import pytest

from audiodec_amd.pipeline import GuardLog, _Batch


class FakeProgram:
    """Posts complete when the fake device clock has passed them; a post's word is what `fail_at` says for that ticket."""

    def __init__(self, name):
        self.name, self.tickets, self.fail_at = name, 0, {}
        self.completed = -1          # tickets <= completed are done
        self.blocked = 0

    def post(self):
        t = self.tickets
        self.tickets += 1
        return t

    def poll(self, ticket, block):
        if ticket > self.completed:
            if not block:
                return False, 0
            self.blocked += 1
            self.completed = ticket      # waiting lets the device get there
        return True, self.fail_at.get(ticket, 0)


def make_log(depth, progs, repaired, drained):
    def poll(prog, ticket, block):
        return prog.poll(ticket, block)

    def repair(batches, by_prog, culprit):
        repaired.append(([b.x for b in batches], {p.name: f for p, f in by_prog.items()}, culprit.name))

    def drain():
        drained.append(1)
        for p in progs:
            p.completed = p.tickets - 1
    return GuardLog(depth, poll, repair, drain)


def issue(log, progs, n):
    b = _Batch()
    b.x = n
    for p in progs:
        b.steps.append((p, 1, p.post()))
    log.push(b)
    return b


def test_clean_batches_are_retired_without_waiting_while_the_log_has_room():
    enc, dec = FakeProgram("enc"), FakeProgram("dec")
    repaired, drained = [], []
    log = make_log(4, [enc, dec], repaired, drained)
    for n in range(10):
        log.collect()
        issue(log, [enc, dec], n)
        enc.completed = dec.completed = n - 2            # the device is two batches behind the host
    assert log.waits == 0 and enc.blocked == dec.blocked == 0 and not repaired
    assert log.verified == 7 and len(log.pending) == 3
    log.collect(block=True)                              # settle(): everything verified, by waiting
    assert log.verified == 10 and not log.pending and not repaired


def test_a_full_log_waits_for_the_oldest_batch_only():
    enc = FakeProgram("enc")
    repaired, drained = [], []
    log = make_log(3, [enc], repaired, drained)
    for n in range(6):
        log.collect()
        issue(log, [enc], n)                             # the device never reports by itself
    # entry of step n (n >= 3) found 3 unverified batches and waited for exactly one
    assert log.waits == 3 and enc.blocked == 3 and len(log.pending) == 3 and log.verified == 3 and not repaired


def test_a_bad_batch_hands_everything_from_it_on_to_the_repair_once():
    enc, dec = FakeProgram("enc"), FakeProgram("dec")
    repaired, drained = [], []
    log = make_log(4, [enc, dec], repaired, drained)
    for n in range(3):
        log.collect()
        issue(log, [enc, dec], n)
    dec.fail_at[1] = 8                                   # the decoder overflowed in batch 1; batch 2 is already issued
    enc.completed = dec.completed = 1
    log.collect()
    assert drained == [1] and repaired == [([1, 2], {"dec": 8}, "dec")]
    assert not log.pending and log.verified == 3 and log.repairs == 1
    # later batches are unaffected
    issue(log, [enc, dec], 3)
    log.collect(block=True)
    assert log.verified == 4 and len(repaired) == 1


def test_flags_of_later_batches_and_other_programs_reach_the_same_repair():
    enc, dec = FakeProgram("enc"), FakeProgram("dec")
    repaired, drained = [], []
    log = make_log(4, [enc, dec], repaired, drained)
    for n in range(4):
        issue(log, [enc, dec], n)
    enc.fail_at[2] = 8
    dec.fail_at[2] = 8
    dec.fail_at[3] = 10                                  # (a second program, and a different bit in a later batch)
    log.collect(block=True)
    # the culprit is the first program, in issue order, of the first bad batch
    assert repaired == [([2, 3], {"enc": 8, "dec": 10}, "enc")] and log.verified == 4 and log.repairs == 1


def test_the_log_never_holds_more_units_than_the_rings_can_be_rewound_by():
    """A batch of several frames takes several frames of the rings' extra rows: the log waits for old batches when the incoming one would
    exceed the budget, whatever the batch count."""
    enc = FakeProgram("enc")
    repaired, drained = [], []

    def poll(prog, ticket, block):
        return prog.poll(ticket, block)
    log = GuardLog(4, poll, lambda *a: repaired.append(a), lambda: drained.append(1), budget=10)
    for n in range(6):
        log.collect(incoming=3)
        b = _Batch(3)
        b.x = n
        b.steps.append((enc, 2, enc.post())); b.steps.append((enc, 1, enc.post()))      # three frames as two program steps
        log.push(b)
        assert log.units_pending() <= 10 and len(log.pending) <= 3                        # 3 batches = 9 frames; a fourth would be 12
    assert log.waits > 0 and not repaired
    log.collect(block=True)
    assert log.verified == 6


def test_pipeline_is_only_deferred_when_the_rings_can_be_rewound():
    """Host-side contract of StreamingPipeline.__init__ (no device needed for the decision itself)."""
    import inspect
    from audiodec_amd import pipeline
    src = inspect.getsource(pipeline.StreamingPipeline.__init__)
    assert "rewind_depth" in src and "offline" in src and "POST_SLOTS" in src
    with pytest.raises(AssertionError):
        GuardLog(0, None, None, None)
