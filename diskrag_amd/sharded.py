"""Graph-sharded search: the id space is cut into disjoint ranges, every range has its own
Vamana sub-graph and PQ codes on one GPU, every query runs on every shard, and the per-shard top-k lists are merged
in canonical (distance, id) order.

The GPU path is the C ABI's dr_sharded_search (include/diskrag_hip.h): lists of the shards of one process are merged by a
device kernel, lists of different processes (one per GPU) travel in ONE RCCL all-gather over xGMI -- packed 64-bit
(distance, id) keys plus the rank's status word, issued on the device-resident merged list -- and are merged on the
device again: no host staging, no PyTorch. `comm` is an `_ffi.Comm` (RCCL communicator; None for a single process).
1 process x 8 shards and 8 processes x 1 shard give the same answer. A rank whose shards fail still joins the exchange
and the call then fails on every rank (DR_E_REMOTE / ShardExchangeError) instead of hanging the others.
`search_submit` is the pipelined form (four exchanges in flight: batch i+1 searches while batch i is exchanged; `set_group(n)`:
n consecutive submits share one exchange).

The reference has no sharded search. Its PQ-only traversal is `beam_search_with_pq` (pydiskann/vamana_graph.py:535-605,
mode M3): a k-sized heap and a trim that pops the BEST candidates (quirk Q9) -- recall 0.00002-0.014 at c3 scale. The
engine's DR_MODE_PQ (M1's loop on ADC distances, diskrag_hip.h) is the flagged, intentional divergence for this path;
both are served. Parity anchor: the merged result equals the merge of the per-shard oracle runs
(tests/test_gpu_sharded.py, tests/test_gpu_round2.py).
"""
import numpy as np

from . import _ffi
from .parallel import PAD


class GraphShard:
    """One shard: an index over the vectors with global ids [base, base + index.N)."""

    def __init__(self, index, base):
        self.index = index
        self.base = int(base)


def globalize(local_ids, base):
    """Shard-local ids -> global ids, padding kept."""
    ids = np.asarray(local_ids, dtype=np.uint32)
    return np.where(ids == PAD, PAD, (ids.astype(np.uint64) + np.uint64(base)).astype(np.uint32))


class ShardedSearch:
    def __init__(self, shards, comm=None):
        """`shards`: the GraphShard objects this process owns (one per GPU in the 8-process layout), every one over a device index
        (`_ffi.HipIndex`). `comm`: `_ffi.Comm` spanning the processes (None: one process)."""
        self.shards = list(shards)
        self.comm = comm
        for sh in self.shards:
            if not isinstance(sh.index, _ffi.HipIndex):
                raise TypeError("ShardedSearch serves device indexes (_ffi.HipIndex)")

    def search_batch(self, queries, k, L=100, beam_width=8, mode=_ffi.MODE_M3, band_policy=0, flags=_ffi.F_USE_PQ):
        """Every local shard searches all queries; returns (global ids [nq,k] PAD-padded, distances [nq,k] NaN-padded,
        info). Raises if any shard reports a non-zero status (nothing is dropped silently)."""
        return self.search_submit(queries, k, L=L, beam_width=beam_width, mode=mode, band_policy=band_policy, flags=flags).wait()

    def search_submit(self, queries, k, L=100, beam_width=8, mode=_ffi.MODE_M3, band_policy=0, flags=_ffi.F_USE_PQ):
        """Pipelined form for device shards (dr_sharded_submit): returns an object whose wait() gives what search_batch
        returns. Four exchanges may be in flight; every rank must submit (and, with set_group, wait) in the same order."""
        job = _ffi.sharded_submit([sh.index for sh in self.shards], [sh.base for sh in self.shards], queries, k, L=L,
                                  beam_width=beam_width, mode=mode, band_policy=band_policy, flags=flags, comm=self.comm)
        return _PendingShardedSearch(job)


    def set_group(self, n):
        """n consecutive search_submit calls share ONE exchange (dr_sharded_set_group: one launch per shard, one all-gather; a count,
        never a timing, so that every rank forms the same exchanges); launched when full, when one of its jobs is waited for, or by flush()."""
        _ffi.sharded_set_group(self.shards[0].index, n)

    def flush(self):
        _ffi.sharded_flush(self.shards[0].index)


class _PendingShardedSearch:
    def __init__(self, job):
        self._job = job

    def wait(self):
        ids, dist, status, ms = self._job.wait()
        if int(status.max(initial=0)) != 0:
            raise _ffi.DiskragHipError(-5, f"sharded search: status {int(status.max())}")
        return ids, dist, {"status": status, "ms": {"search": float(ms[0]), "all_gather": float(ms[1]), "merge": float(ms[2])}}
