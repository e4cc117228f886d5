"""Request coalescing for the serving seam ("FastAPI handlers calling search_batch", app.py:84-130).

The reference's `/search` and `/faq-search` handlers ask ONE query per request and the engine answers it with one graph
walk on one core. On the device a one-query call costs as much as a several-thousand-query call (one wavefront walks the
query's expansions one after the other), so concurrent requests are worth collecting: `RequestBatcher` puts
the queries of requests that arrive within `max_wait_ms` of each other into one `search_batch` call and hands every caller
its own rows. A query's answer does not depend on what else is in its batch (one wavefront per query, no cross-query
state), so the rows are the bits a direct call returns.

Only host logic: threads, a queue, numpy. The engine is anything with the facade's
`search_batch(query_vectors, k, L, beam_width, use_pq_search) -> (ids, dist, count, stats)`.
"""
import threading
import time
from concurrent.futures import Future
from typing import List, Optional, Tuple

import numpy as np


class RequestBatcher:
    """Coalesces concurrent one-query requests into batched engine calls.

    engine         SearchEngineCorrect (or anything with its search_batch)
    k_max          every batch runs with this k; a request may ask for any k <= k_max and gets the first k rows
    L, beam_width, use_pq_search   the search parameters of every request served by this batcher (one batcher per setting)
    max_batch      a batch is sent as soon as it holds this many queries ...
    max_wait_ms    ... or when its oldest request has waited this long (0: send whatever is queued right away)
    """

    def __init__(self, engine, k_max: int = 10, L: Optional[int] = None, beam_width: Optional[int] = 8,
                 use_pq_search: bool = True, max_batch: int = 1024, max_wait_ms: float = 0.2):
        if k_max <= 0 or max_batch <= 0 or max_wait_ms < 0:
            raise ValueError("k_max and max_batch must be positive, max_wait_ms non-negative")
        self.engine = engine
        self.k_max, self.L, self.beam_width, self.use_pq_search = int(k_max), L, beam_width, use_pq_search
        self.max_batch, self.max_wait = int(max_batch), max_wait_ms / 1e3
        self._cv = threading.Condition()
        self._pending: List[Tuple[np.ndarray, int, Future, float]] = []
        self._closed = False
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
        fut: Future = Future()
        with self._cv:
            if self._closed:
                raise RuntimeError("RequestBatcher is closed")
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
            return batch

    def _run(self):
        while True:
            batch = self._take_batch()
            if batch is None:
                return
            try:
                qs = np.stack([b[0] for b in batch])
                ids, dist, cnt, st = self.engine.search_batch(qs, k=self.k_max, L=self.L, beam_width=self.beam_width,
                                                              use_pq_search=self.use_pq_search)
            except BaseException as e:          # every waiter of the batch learns why it failed
                for _, _, fut, _ in batch:
                    fut.set_exception(e)
                continue
            self.batches_sent += 1
            self.queries_sent += len(batch)
            for i, (_, k, fut, _) in enumerate(batch):
                n = min(int(cnt[i]), k)
                results = [(np.float32(dist[i, t]), np.uint32(ids[i, t])) for t in range(n)]
                stats = {"search_steps": int(st["steps"][i]), "nodes_visited": int(st["visited"][i]),
                         "exact_distance_computations": int(st["exact"][i]), "pq_distance_computations": int(st["pq"][i])}
                fut.set_result((results, stats))
