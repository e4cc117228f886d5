"""Request coalescing for the serving seam ("FastAPI handlers calling search_batch", app.py:84-130).

The reference's `/search` and `/faq-search` handlers ask ONE query per request and the engine answers it with one graph
walk on one core. On the device a one-query call costs as much as a several-thousand-query call (one wavefront walks the
query's expansions one after the other), so concurrent requests are worth collecting: `RequestBatcher` puts
the queries of requests that arrive within `max_wait_ms` of each other into one `search_batch` call and hands every caller
its own rows. A query's answer does not depend on what else is in its batch (one wavefront per query, no cross-query
state), so the rows are the bits a direct `search_batch(k=k_max, L=<the batcher's L>)` call returns for that query. (The
facade derives L = max(2k, 20) from k when L is None, search_engine.py:539-540: the batcher derives it ONCE from k_max, so a
request that asks for k < k_max is searched with the list of k_max and gets the first k rows of that search -- the same
bits as a direct call with that explicit L, not necessarily those of a direct call with L=None and the smaller k.)

Only host logic: threads, a queue, numpy. The engine is anything with the facade's
`search_batch(query_vectors, k, L, beam_width, use_pq_search) -> (ids, dist, count, stats)`.
"""
import threading
import time
from concurrent.futures import Future, InvalidStateError
from typing import List, Optional, Tuple

import numpy as np


class RequestBatcher:
    """Coalesces concurrent one-query requests into batched engine calls.

    engine         SearchEngineCorrect (or anything with its search_batch)
    k_max          every batch runs with this k; a request may ask for any k <= k_max and gets the first k rows
    L, beam_width, use_pq_search   the search parameters of every request served by this batcher (one batcher per setting);
                   L=None means max(2 * k_max, 20), fixed here -- the reference's default for k = k_max
    max_batch      a batch is sent as soon as it holds this many queries ...
    max_wait_ms    ... or when its oldest request has waited this long (0: send whatever is queued right away)
    """

    def __init__(self, engine, k_max: int = 10, L: Optional[int] = None, beam_width: Optional[int] = 8,
                 use_pq_search: bool = True, max_batch: int = 1024, max_wait_ms: float = 0.2):
        if k_max <= 0 or max_batch <= 0 or max_wait_ms < 0:
            raise ValueError("k_max and max_batch must be positive, max_wait_ms non-negative")
        self.engine = engine
        self.k_max, self.beam_width, self.use_pq_search = int(k_max), beam_width, use_pq_search
        self.L = max(2 * self.k_max, 20) if L is None else int(L)
        self.dimension = getattr(engine, "dimension", None)     # malformed requests are refused one by one in submit()
        self.max_batch, self.max_wait = int(max_batch), max_wait_ms / 1e3
        self._cv = threading.Condition()
        self._pending: List[Tuple[np.ndarray, int, Future, float]] = []
        self._closed = False
        self._dim_seen = None
        self.batches_sent = 0
        self.queries_sent = 0
        self._worker = threading.Thread(target=self._run, name="diskrag-request-batcher", daemon=True)
        self._worker.start()

    # ---- callers
    def submit(self, query_vector, k: Optional[int] = None) -> Future:
        """Queues one query; the Future resolves to (results, stats) in the shape of _pq_accelerated_graph_search:
        results = [(np.float32 distance, np.uint32 id), ...] (at most k), stats = the reference's four counters."""
        k = self.k_max if k is None else int(k)
        if k <= 0 or k > self.k_max:
            raise ValueError(f"k must be in 1..{self.k_max}")
        q = np.asarray(query_vector, dtype=np.float32).reshape(-1)
        if self.dimension is not None and q.shape[0] != int(self.dimension):
            # (refused here: inside a batch it would make np.stack fail for every request coalesced with it)
            raise ValueError(f"query has {q.shape[0]} components, the index {int(self.dimension)}")
        fut: Future = Future()
        with self._cv:
            if self._closed:
                raise RuntimeError("RequestBatcher is closed")
            # (an engine without a `dimension`: the size is learnt from the first batch the engine has actually SERVED -- _run latches it --,
            # never from a request alone: one malformed first request would otherwise refuse every well-formed one for good)
            if self._dim_seen is not None and q.shape[0] != self._dim_seen:
                raise ValueError(f"query has {q.shape[0]} components, the engine has served requests of {self._dim_seen}")
            self._pending.append((q, k, fut, time.perf_counter()))
            self._cv.notify()
        return fut

    def search(self, query_vector, k: Optional[int] = None, timeout: Optional[float] = None):
        """Blocking form of submit()."""
        return self.submit(query_vector, k).result(timeout)

    def close(self):
        """Serves what is queued, then stops the worker."""
        with self._cv:
            self._closed = True
            self._cv.notify()
        self._worker.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- worker
    def _take_batch(self):
        with self._cv:
            while not self._pending and not self._closed:
                self._cv.wait()
            if not self._pending:
                return None
            # collect until the batch is full or its oldest request has waited max_wait
            deadline = self._pending[0][3] + self.max_wait
            while len(self._pending) < self.max_batch and not self._closed:
                left = deadline - time.perf_counter()
                if left <= 0:
                    break
                self._cv.wait(left)
            batch, self._pending = self._pending[:self.max_batch], self._pending[self.max_batch:]
        # a caller may have given up on its request (a cancelled Future: an asyncio.wrap_future timeout, a client that went
        # away): it is dropped here and can no longer be cancelled once the batch runs
        return [b for b in batch if b[2].running() or b[2].set_running_or_notify_cancel()]      # (running: put back by _run, a batch of mixed sizes)

    def _run(self):
        while True:
            batch = self._take_batch()
            if batch is None:
                return
            if not batch:                       # every request of it was cancelled
                continue
            sizes = {b[0].shape[0] for b in batch}
            if len(sizes) > 1:                  # (only while no size is known: requests of each size are served on their own)
                first = batch[0][0].shape[0]
                rest = [b for b in batch if b[0].shape[0] != first]
                batch = [b for b in batch if b[0].shape[0] == first]
                with self._cv:
                    self._pending[:0] = rest
            try:
                qs = np.stack([b[0] for b in batch])
                ids, dist, cnt, st = self.engine.search_batch(qs, k=self.k_max, L=self.L, beam_width=self.beam_width,
                                                              use_pq_search=self.use_pq_search)
            except BaseException as e:          # every waiter of the batch learns why it failed
                for _, _, fut, _ in batch:
                    self._deliver(fut, exc=e)
                continue
            if self._dim_seen is None and self.dimension is None:
                with self._cv:
                    self._dim_seen = qs.shape[1]
            self.batches_sent += 1
            self.queries_sent += len(batch)
            for i, (_, k, fut, _) in enumerate(batch):
                n = min(int(cnt[i]), k)
                results = [(np.float32(dist[i, t]), np.uint32(ids[i, t])) for t in range(n)]
                stats = {"search_steps": int(st["steps"][i]), "nodes_visited": int(st["visited"][i]),
                         "exact_distance_computations": int(st["exact"][i]), "pq_distance_computations": int(st["pq"][i])}
                self._deliver(fut, value=(results, stats))

    @staticmethod
    def _deliver(fut, value=None, exc=None):
        """the worker thread must outlive anything a caller does to its Future"""
        try:
            if exc is not None:
                fut.set_exception(exc)
            else:
                fut.set_result(value)
        except InvalidStateError:
            pass
