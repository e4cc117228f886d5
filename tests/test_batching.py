"""RequestBatcher (diskrag_amd/batching.py): host logic only -- a fake engine stands in for the device."""
import threading
import time

import numpy as np
import pytest

from diskrag_amd.batching import RequestBatcher

STATS = np.dtype([("steps", np.uint32), ("visited", np.uint32), ("exact", np.uint32), ("pq", np.uint32), ("status", np.uint32)])


class FakeEngine:
    """ids of query q = floor(q[0]) + 0..k-1, distances = q[1] + 0..k-1; remembers the batch sizes it was called with."""

    def __init__(self, delay=0.0, fail_on=None, short=None):
        self.calls, self.delay, self.fail_on, self.short = [], delay, fail_on, short
        self.lock = threading.Lock()

    def search_batch(self, query_vectors, k=10, L=None, beam_width=8, use_pq_search=True, band_policy=0):
        q = np.asarray(query_vectors)
        with self.lock:
            self.calls.append((q.shape[0], k, L, beam_width, use_pq_search))
        if self.delay:
            time.sleep(self.delay)
        if self.fail_on is not None and (q[:, 0] == self.fail_on).any():
            raise RuntimeError("device said no")
        nq = q.shape[0]
        ids = (np.floor(q[:, :1]).astype(np.uint32) + np.arange(k, dtype=np.uint32)[None, :])
        dist = (q[:, 1:2] + np.arange(k, dtype=np.float32)[None, :]).astype(np.float32)
        cnt = np.full(nq, k if self.short is None else min(k, self.short), np.uint32)
        st = np.zeros(nq, STATS)
        st["steps"] = np.arange(nq); st["visited"] = 2; st["exact"] = 3; st["pq"] = 4
        return ids, dist, cnt, st


def test_every_caller_gets_its_own_rows_and_requests_are_coalesced():
    eng = FakeEngine(delay=0.01)
    out = {}
    with RequestBatcher(eng, k_max=10, L=100, beam_width=8, max_batch=64, max_wait_ms=20) as rb:
        def client(i):
            res, stats = rb.search(np.array([100.0 * i, 0.5 * i, 0, 0], np.float32), k=1 + i % 10)
            out[i] = (res, stats)
        th = [threading.Thread(target=client, args=(i,)) for i in range(48)]
        for t in th: t.start()
        for t in th: t.join()
    assert len(out) == 48
    for i, (res, stats) in out.items():
        k = 1 + i % 10
        assert [int(x) for _, x in res] == [100 * i + t for t in range(k)]
        assert [float(d) for d, _ in res] == [np.float32(0.5 * i) + t for t in range(k)]
        assert isinstance(res[0][0], np.float32) and isinstance(res[0][1], np.uint32)
        assert stats["nodes_visited"] == 2 and stats["exact_distance_computations"] == 3 and stats["pq_distance_computations"] == 4
    assert sum(c[0] for c in eng.calls) == 48 and len(eng.calls) < 48          # coalesced: fewer calls than requests
    assert all(c[1:] == (10, 100, 8, True) for c in eng.calls)                   # every batch with k_max and the batcher's setting
    assert rb.queries_sent == 48 and rb.batches_sent == len(eng.calls)


def test_full_batches_leave_at_once_and_a_lone_request_waits_no_longer_than_max_wait():
    eng = FakeEngine()
    with RequestBatcher(eng, k_max=4, max_batch=8, max_wait_ms=5000) as rb:       # only a full batch can leave
        futs = [rb.submit(np.array([i, 0], np.float32)) for i in range(16)]
        t0 = time.perf_counter()
        for f in futs:
            f.result(timeout=5)
        assert time.perf_counter() - t0 < 2.0
        assert [c[0] for c in eng.calls] == [8, 8]
    eng2 = FakeEngine()
    with RequestBatcher(eng2, k_max=4, max_batch=8, max_wait_ms=30) as rb:
        t0 = time.perf_counter()
        res, _ = rb.search(np.array([7, 1], np.float32), k=2)
        dt = time.perf_counter() - t0
        assert 0.02 < dt < 1.0 and [int(x) for _, x in res] == [7, 8]


def test_an_engine_error_reaches_exactly_the_requests_of_its_batch_and_the_batcher_lives_on():
    eng = FakeEngine(fail_on=13.0)
    with RequestBatcher(eng, k_max=3, max_batch=4, max_wait_ms=1) as rb:
        with pytest.raises(RuntimeError, match="device said no"):
            rb.search(np.array([13.0, 0], np.float32))
        res, _ = rb.search(np.array([5.0, 0], np.float32))
        assert [int(x) for _, x in res] == [5, 6, 7]


def test_short_result_lists_argument_checks_and_close():
    eng = FakeEngine(short=2)
    rb = RequestBatcher(eng, k_max=5, max_wait_ms=0)
    res, _ = rb.search(np.array([1.0, 0], np.float32), k=4)
    assert len(res) == 2                                                       # the engine found two: count, not k
    with pytest.raises(ValueError):
        rb.submit(np.zeros(2, np.float32), k=6)
    with pytest.raises(ValueError):
        rb.submit(np.zeros(2, np.float32), k=0)
    f = rb.submit(np.array([9.0, 0], np.float32))
    rb.close()                                                                  # serves what is queued before it stops
    assert [int(x) for _, x in f.result(timeout=1)[0]] == [9, 10]
    with pytest.raises(RuntimeError):
        rb.submit(np.zeros(2, np.float32))
    with pytest.raises(ValueError):
        RequestBatcher(eng, k_max=0)


def test_a_cancelled_request_neither_kills_the_worker_nor_reaches_the_engine():
    """a caller that cancels its pending Future (a handler's timeout, a client that went away) must not make the
    worker thread die on InvalidStateError -- every later request would then block forever."""
    eng = FakeEngine(delay=0.05)
    with RequestBatcher(eng, k_max=4, L=50, max_batch=4, max_wait_ms=1) as rb:
        first = rb.submit(np.array([1.0, 0, 0, 0], np.float32))            # occupies the worker for 50 ms
        time.sleep(0.01)
        gone = rb.submit(np.array([2.0, 0, 0, 0], np.float32))
        assert gone.cancel()                                               # still pending: cancellable
        kept = rb.submit(np.array([3.0, 0, 0, 0], np.float32))
        assert [int(x) for _, x in first.result(5)[0]] == [1, 2, 3, 4]
        assert [int(x) for _, x in kept.result(5)[0]] == [3, 4, 5, 6]
        # the worker is alive and serves later requests
        assert [int(x) for _, x in rb.search(np.array([9.0, 0, 0, 0], np.float32), timeout=5)[0]] == [9, 10, 11, 12]
    assert sum(c[0] for c in eng.calls) == 3                               # the cancelled query never ran


def test_a_future_resolved_behind_the_workers_back_is_survived():
    eng = FakeEngine(delay=0.03)
    with RequestBatcher(eng, k_max=2, L=50, max_batch=2, max_wait_ms=0) as rb:
        f = rb.submit(np.array([5.0, 0, 0, 0], np.float32))
        time.sleep(0.01)                          # the batch is running: not cancellable, but a caller can still break the Future
        try:
            f.set_exception(RuntimeError("caller gave up"))
        except Exception:
            pass
        assert [int(x) for _, x in rb.search(np.array([7.0, 0, 0, 0], np.float32), timeout=5)[0]] == [7, 8]


def test_a_malformed_request_fails_alone():
    class Eng(FakeEngine):
        dimension = 4
    eng = Eng()
    with RequestBatcher(eng, k_max=3, L=50, max_batch=8, max_wait_ms=20) as rb:
        ok = rb.submit(np.zeros(4, np.float32))
        with pytest.raises(ValueError):
            rb.submit(np.zeros(5, np.float32))                             # refused in submit(): never stacked with the others
        assert len(ok.result(5)[0]) == 3
    eng2 = FakeEngine()                                                    # an engine that does not say its dimension: the first SERVED batch sets it
    with RequestBatcher(eng2, k_max=3, L=50, max_batch=8, max_wait_ms=20) as rb:
        ok = rb.submit(np.zeros(4, np.float32))
        assert len(ok.result(5)[0]) == 3
        with pytest.raises(ValueError):
            rb.submit(np.zeros(6, np.float32))


def test_default_list_size_is_fixed_from_k_max():
    """L=None: the batcher hands the engine max(2 * k_max, 20) explicitly, whatever k a request asks for"""
    eng = FakeEngine()
    with RequestBatcher(eng, k_max=16, max_batch=4, max_wait_ms=0) as rb:
        rb.search(np.zeros(4, np.float32), k=2, timeout=5)
    assert eng.calls[0][1:3] == (16, 32)
    eng = FakeEngine()
    with RequestBatcher(eng, k_max=5, max_batch=4, max_wait_ms=0) as rb:
        rb.search(np.zeros(4, np.float32), timeout=5)
    assert eng.calls[0][1:3] == (5, 20)


def test_a_malformed_first_request_does_not_fix_the_dimension():
    """An engine without a `dimension` attribute: the size is learnt from the first batch the engine has SERVED, never from a request
    alone -- a malformed first request fails by itself and well-formed ones keep working (and a later wrong size is refused)."""
    class Picky(FakeEngine):
        def search_batch(self, query_vectors, **kw):
            q = np.asarray(query_vectors)
            if q.shape[1] != 4:
                raise ValueError("wrong dimension")
            return super().search_batch(q, **kw)
    eng = Picky()
    with RequestBatcher(eng, k_max=3, L=20, beam_width=8, max_batch=8, max_wait_ms=30) as rb:
        bad = rb.submit(np.zeros(7, np.float32))              # first request of the batcher's life: the wrong size
        good = [rb.submit(np.array([10.0 * i, 1, 0, 0], np.float32)) for i in range(5)]      # coalesced with it
        with pytest.raises(ValueError):
            bad.result(5)
        for i, f in enumerate(good):
            res, _ = f.result(5)
            assert [int(x) for _, x in res] == [10 * i, 10 * i + 1, 10 * i + 2]
        with pytest.raises(ValueError):                       # the served size is the batcher's size from now on
            rb.submit(np.zeros(7, np.float32))
        assert rb.search(np.array([50.0, 0, 0, 0], np.float32), k=1)[0][0][1] == 50
