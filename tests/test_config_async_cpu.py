"""CPU: the host logic of async mode (luciddreamer_amd.config) -- high-water mark, warm calls, the overflow policies
("verify": the forward waits for its own early header; "drop" / "raise": deferred check over header tickets) -- with the
binding's two ticket functions replaced by a fake that
completes read-backs on demand.  The real lr_header_post / lr_header_poll are exercised on the GPU
(tests/test_gpu_parity.py::test_header_tickets)."""
import types
import warnings

import pytest
import torch

from luciddreamer_amd import config


class FakeBinding:
    """header_post hands out tickets; header_poll returns None until `complete()` was called for the ticket."""

    def __init__(self):
        self.next, self.words, self.done, self.polled = 0, {}, set(), []

    def header_post(self, geom):
        t = self.next
        self.next += 1
        self.words[t] = tuple(int(v) for v in geom)
        return t

    def header_poll(self, ticket, block=False):
        self.polled.append((ticket, block))
        if not block and ticket not in self.done:
            return None
        return self.words.pop(ticket)

    def complete(self, *tickets):
        self.done.update(tickets)


@pytest.fixture
def fake(monkeypatch):
    import luciddreamer_amd
    f = FakeBinding()
    monkeypatch.setitem(__import__("sys").modules, "luciddreamer_amd._C", f)
    monkeypatch.setattr(luciddreamer_amd, "_C", f, raising=False)
    config.set_async(False)
    config.reset()
    config.dropped_views = config.rerendered_views = config.recovered_views = 0
    yield f
    config._pending.clear()
    config._stack().clear()
    config._tls.exact = 0
    config.set_async(True)                     # the module default
    config._hwm.clear()
    config._seen.clear()


def _rs(h=64, w=96):
    return types.SimpleNamespace(image_height=h, image_width=w)


def _header(num_rendered, overflow=0, trap=0, instances=None):
    inst = num_rendered if instances is None else instances
    return [num_rendered, overflow, trap, 0, 0, inst, inst, 0]


def test_defaults_are_async_with_verify_and_two_warm_calls():
    import importlib
    import os
    assert os.environ.get("LUCID_RASTER_EXACT", "0") != "1"
    mod = importlib.reload(config)
    assert mod.is_async() and mod.current_policy() == "verify" and mod._warm_calls == 2


def test_exact_until_a_mark_exists_then_capacity_from_the_mark(fake):
    means = torch.zeros(1000, 3)
    rs = _rs()
    assert config.capacity_for(means, rs) == 0                      # async off: always exact
    config.set_async(True, headroom=1.5, warm_calls=1)
    assert config.capacity_for(means, rs) == 0                      # first sighting: measured exactly
    config.note_forward(means, rs, 10_000, None, 0)                 # an exact forward feeds the mark directly, no ticket
    assert fake.next == 0
    cap = config.capacity_for(means, rs)
    assert cap == int(10_000 * 1.5) + 4096
    assert config.capacity_for(torch.zeros(0, 3), rs) == 0          # empty cloud: nothing to size
    assert config.capacity_for(torch.zeros(1000, 3), _rs(32, 32)) == 0      # another (P, H, W): its own first sighting


def test_verify_policy_the_forward_waits_for_its_own_header(fake):
    means, rs = torch.zeros(500, 3), _rs()
    config.set_async(True, headroom=1.0, warm_calls=1)               # on_overflow="verify" is the default
    config.note_forward(means, rs, 1_000, None, 0)
    cap = config.capacity_for(means, rs)
    assert config.verifying(cap) and not config.verifying(0)
    t_ok = fake.header_post(_header(900))                            # what lr_forward posts after its compaction scan
    assert config.verify(means, rs, t_ok) is False and fake.polled[-1] == (t_ok, True)      # blocks on ITS ticket
    t_over = fake.header_post(_header(50_000, overflow=1))
    assert config.verify(means, rs, t_over) is True                  # overflowed: the caller renders again in exact mode
    assert config.rerendered_views == 1 and config.dropped_views == 0
    assert config.capacity_for(means, rs) == 50_000 + 4096            # the mark follows the true count
    assert config.verify(means, rs, -1) is False                     # no ticket (exact mode, empty cloud)
    config.note_forward(means, rs, -1, _header(10), cap)             # "verify" posts nothing afterwards
    assert not config._pending
    with config.overflow_policy("drop"):
        assert not config.verifying(cap)


def test_deferred_check_polls_in_order_and_raises_the_mark(fake):
    means, rs = torch.zeros(500, 3), _rs()
    config.set_async(True, headroom=1.2, warm_calls=1, on_overflow="drop")
    config.note_forward(means, rs, 5_000, None, 0)
    cap = config.capacity_for(means, rs)
    config.note_forward(means, rs, -1, _header(7_000, instances=6_500), cap)    # async forward: a ticket, nothing blocks
    config.note_forward(means, rs, -1, _header(6_000, instances=5_500), cap)
    assert fake.next == 2 and len(config._pending) == 2
    config.capacity_for(means, rs)                                   # polls: nothing has completed
    assert len(config._pending) == 2
    fake.complete(1)                                                 # out of order: the queue waits for ticket 0
    config.capacity_for(means, rs)
    assert len(config._pending) == 2
    fake.complete(0)
    assert config.capacity_for(means, rs) == int(6_500 * 1.2) + 4096          # the mark follows the instances actually emitted
    assert not config._pending


def test_deferred_overflow_policies(fake):
    means, rs = torch.zeros(500, 3), _rs()
    config.set_async(True, headroom=1.0, warm_calls=1, on_overflow="raise")
    config.note_forward(means, rs, 1_000, None, 0)
    cap = config.capacity_for(means, rs)
    config.note_forward(means, rs, -1, _header(50_000, overflow=1), cap)
    with pytest.raises(RuntimeError, match="ZERO"):
        config.drain()                                               # blocks on the ticket, then raises
    assert fake.polled[-1] == (0, True)
    assert config.capacity_for(means, rs) == 50_000 + 4096            # raised from the true count either way
    config.set_async(True, headroom=1.0, warm_calls=1, on_overflow="drop")
    config.note_forward(means, rs, -1, _header(80_000, overflow=1), cap)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        config.drain()
    assert any("binning capacity" in str(x.message) and "ZERO" in str(x.message) for x in w)
    assert config.dropped_views == 2
    config.note_forward(means, rs, -1, _header(10, trap=1), cap)
    with pytest.raises(RuntimeError, match="prefiltered"):
        config.drain()
    with pytest.raises(ValueError):
        config.set_async(True, on_overflow="warn")                   # the round-2 policy (train on truncated gradients) is gone
    # a temporary policy (what parallel.ViewStreams does around its views)
    config.set_async(True, warm_calls=1)
    assert config.current_policy() == "verify"
    with config.overflow_policy("drop"):
        assert config.current_policy() == "drop"
        config.note_forward(means, rs, -1, _header(10), cap)
    assert config.current_policy() == "verify" and len(config._pending) == 1
    fake.complete(*range(fake.next))
    config.drain()


def test_check_every_samples_and_warm_calls_stay_exact(fake):
    means, rs = torch.zeros(500, 3), _rs()
    config.set_async(True, check_every=3, warm_calls=2, on_overflow="drop")
    assert config.capacity_for(means, rs) == 0
    config.note_forward(means, rs, 1_000, None, 0)
    assert config.capacity_for(means, rs) == 0                       # second warm call: still exact
    config.note_forward(means, rs, 2_000, None, 0)
    cap = config.capacity_for(means, rs)
    assert cap == int(2_000 * 1.3) + 4096
    base = config._calls
    for _ in range(6):
        config.note_forward(means, rs, -1, _header(1_500), cap)
    assert fake.next == (base + 6) // 3 - base // 3                  # every third async forward posts a ticket
    fake.complete(*range(fake.next))
    config.drain()


def test_recover_policy_marks_the_view_and_neither_warns_nor_counts_a_drop(fake):
    """What parallel.ViewStreams runs under: the entry of an overflowed view carries True after the drain (the owner runs
    the view again), nothing is warned about or counted as dropped, every view is checked whatever check_every says, and
    force_exact() makes the re-run an exact-mode forward."""
    means, rs = torch.zeros(500, 3), _rs()
    config.set_async(True, headroom=1.0, warm_calls=1, check_every=4)
    config.note_forward(means, rs, 1_000, None, 0)
    cap = config.capacity_for(means, rs)
    with config.overflow_policy("recover", _owner=object()):
        assert not config.verifying(cap)
        for h in (_header(900), _header(50_000, overflow=1), _header(950)):
            config.note_forward(means, rs, -1, h, cap)
    entries = list(config._pending)
    assert len(entries) == 3 and all(e[3] is None for e in entries)           # check_every does not sample under "recover"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        config.drain()
    assert [e[3] for e in entries] == [False, True, False] and not w and config.dropped_views == 0
    assert config.capacity_for(means, rs) == 50_000 + 4096
    with config.force_exact():
        assert config.capacity_for(means, rs) == 0
        with config.force_exact():
            assert config.capacity_for(means, rs) == 0
        assert config.capacity_for(means, rs) == 0
    assert config.capacity_for(means, rs) != 0


def test_a_forward_is_tied_to_its_own_entry_even_when_it_retires_older_ones(fake):
    """parallel.ViewStreams asks for the entry of the forward it just issued.  The forward's own capacity_for() polls and
    retires completed entries, so the LENGTH of the pending list can shrink across a forward that appended one (the round-4
    bookkeeping compared lengths and lost such a view: its overflow would have gone unnoticed under "recover")."""
    means, rs = torch.zeros(500, 3), _rs()
    config.set_async(True, headroom=1.0, warm_calls=1)
    config.note_forward(means, rs, 1_000, None, 0)
    with config.overflow_policy("recover", _owner=object()):
        cap = config.capacity_for(means, rs)
        config.note_forward(means, rs, -1, _header(900), cap)
        first = config.take_last_entry()
        config.note_forward(means, rs, -1, _header(950), cap)
        second = config.take_last_entry()
        assert first is not None and second is not None and first is not second
        assert config.take_last_entry() is None                       # cleared by the call
        fake.complete(0, 1)                                           # both headers have arrived ...
        n0 = len(config._pending)
        cap = config.capacity_for(means, rs)                          # ... so the next forward's poll retires them
        config.note_forward(means, rs, -1, _header(50_000, overflow=1), cap)
        assert len(config._pending) < n0 + 1                          # the length went DOWN across a forward that posted
        third = config.take_last_entry()
        assert third is config._pending[-1] and third is not second
        fake.complete(2)
        config.drain()
        assert (first[3], second[3], third[3]) == (False, False, True)
        config.note_forward(means, rs, 1_200, None, 0)                # an exact forward posts nothing
        assert config.take_last_entry() is None


def test_policy_stack_is_per_thread_and_tolerates_an_unbalanced_exit(fake):
    import threading
    seen = {}
    with config.overflow_policy("drop"):
        t = threading.Thread(target=lambda: seen.setdefault("other", config.current_policy()))
        t.start()
        t.join()
        assert config.current_policy() == "drop"
    assert seen["other"] == "verify"                                 # another thread never sees this thread's override
    config.overflow_policy("drop").__exit__(None, None, None)        # exit without enter: no IndexError, nothing popped
    assert config.current_policy() == "verify"


def test_recover_is_not_a_public_policy():
    """"recover" only makes sense with an owner that runs an overflowed view again (parallel.ViewStreams); as a
    global policy the view would silently contribute nothing."""
    with pytest.raises(ValueError):
        config.set_async(True, on_overflow="recover")
    with pytest.raises(ValueError):
        config.overflow_policy("recover")
    config.overflow_policy("recover", _owner=object())
