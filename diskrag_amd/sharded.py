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

A host-logic twin (per-shard `search_batch` calls merged with numpy, optionally exchanged through a torch.distributed
group that the caller passes in) is kept for objects that are not device indexes: it is what the 2-rank gloo test on
CPU exercises. torch is imported only when such a group is given.

The reference has no sharded search. Its PQ-only traversal is `beam_search_with_pq` (pydiskann/vamana_graph.py:535-605,
mode M3): a k-sized heap and a trim that pops the BEST candidates (quirk Q9) -- recall 0.00002-0.014 at c3 scale. The
engine's DR_MODE_PQ (M1's loop on ADC distances, diskrag_hip.h) is the flagged, intentional divergence for this path;
both are served. Parity anchor: the merged result equals the merge of the per-shard oracle runs
(tests/test_gpu_sharded.py, tests/test_gpu_round2.py).
"""
import numpy as np

from . import _ffi
from .parallel import PAD, merge_topk


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
    def __init__(self, shards, comm=None, group=None, collective_device=None):
        """`shards`: the GraphShard objects this process owns (one per GPU in the 8-process layout).
        `comm`: `_ffi.Comm` spanning the processes (device indexes). `group`: a torch.distributed process group for the
        host-logic twin (CPU test only)."""
        self.shards = list(shards)
        self.comm = comm
        self.group = group
        self.collective_device = collective_device

    def _on_device(self):
        return all(isinstance(sh.index, _ffi.HipIndex) for sh in self.shards)

    def search_batch(self, queries, k, L=100, beam_width=8, mode=_ffi.MODE_M3, band_policy=0, flags=_ffi.F_USE_PQ):
        """Every local shard searches all queries; returns (global ids [nq,k] PAD-padded, distances [nq,k] NaN-padded,
        info). Raises if any shard reports a non-zero status (nothing is dropped silently)."""
        if self._on_device():
            return self.search_submit(queries, k, L=L, beam_width=beam_width, mode=mode, band_policy=band_policy, flags=flags).wait()
        # ---- host-logic twin. The local phase may fail (a shard raises, a work area overflows): with a group the rank still
        # joins the one collective -- empty list, non-zero status word -- so that nobody is left waiting in it, and every
        # rank raises afterwards (the failing rank its own error, the others ShardExchangeError).
        ids_l, dist_l, stats_l, err = [], [], [], None
        try:
            for sh in self.shards:
                ids, dist, cnt, st = sh.index.search_batch(queries, k, L=L, beam_width=beam_width, mode=mode,
                                                           band_policy=band_policy, flags=flags)
                if int(st["status"].max(initial=0)) != 0:
                    raise _ffi.DiskragHipError(-5, f"shard at base {sh.base}: search status {int(st['status'].max())}")
                ids_l.append(globalize(ids, sh.base))
                dist_l.append(dist)
                stats_l.append(st)
            ids, dist = merge_topk(ids_l, dist_l, k)
        except Exception as e:           # noqa: BLE001 -- whatever it was, the other ranks must not hang on it
            if self.group is None:
                raise
            err = e
            nq = len(queries)
            ids, dist = np.full((nq, k), PAD, dtype=np.uint32), np.full((nq, k), np.nan, dtype=np.float32)
        if self.group is not None:
            from .parallel import ShardExchangeError, allgather_merge_topk
            try:
                # ids are already global: shard_base 0 in the exchange
                ids, dist = allgather_merge_topk(ids, dist, 0, k, group=self.group, device=self.collective_device,
                                                 local_status=0 if err is None else 1)
            except ShardExchangeError:
                if err is not None:
                    raise err
                raise
        return ids, dist, stats_l

    def search_submit(self, queries, k, L=100, beam_width=8, mode=_ffi.MODE_M3, band_policy=0, flags=_ffi.F_USE_PQ):
        """Pipelined form for device shards (dr_sharded_submit): returns an object whose wait() gives what search_batch
        returns. Four exchanges may be in flight; every rank must submit (and, with set_group, wait) in the same order."""
        if not self._on_device():
            raise TypeError("search_submit needs device indexes")
        job = _ffi.sharded_submit([sh.index for sh in self.shards], [sh.base for sh in self.shards], queries, k, L=L,
                                  beam_width=beam_width, mode=mode, band_policy=band_policy, flags=flags, comm=self.comm)
        return _PendingShardedSearch(job)


    def set_group(self, n):
        """n consecutive search_submit calls share ONE exchange (dr_sharded_set_group: one launch per shard, one all-gather; a count,
        never a timing, so that every rank forms the same exchanges); launched when full, when one of its jobs is waited for, or by flush()."""
        if not self._on_device():
            raise TypeError("set_group needs device indexes")
        _ffi.sharded_set_group(self.shards[0].index, n)

    def flush(self):
        if self._on_device():
            _ffi.sharded_flush(self.shards[0].index)


class _PendingShardedSearch:
    def __init__(self, job):
        self._job = job

    def wait(self):
        ids, dist, status, ms = self._job.wait()
        if int(status.max(initial=0)) != 0:
            raise _ffi.DiskragHipError(-5, f"sharded search: status {int(status.max())}")
        return ids, dist, {"status": status, "ms": {"search": float(ms[0]), "all_gather": float(ms[1]), "merge": float(ms[2])}}
