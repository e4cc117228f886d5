"""Graph-sharded search: the id space is cut into disjoint ranges, every range has its own
Vamana sub-graph and PQ codes on one GPU, every query runs on every shard, and the per-shard top-k lists are merged
in canonical (distance, id) order. Across processes the lists travel in ONE all-gather of nq*k*8 bytes per rank
(`parallel.allgather_merge_topk`; RCCL over xGMI when the group's backend is "nccl"); shards that live in the same
process are merged locally first, so 1 process x 8 shards and 8 processes x 1 shard give the same answer.

The reference has no sharded search; its PQ-only traversal is `beam_search_with_pq` (pydiskann/vamana_graph.py:535-605,
mode M3 here) and that is what each shard runs by default. Parity anchor: the merged result equals the merge of the
per-shard oracle runs (tests/test_gpu_sharded.py), and across ranks the gloo test in tests/test_parallel_gloo.py.
"""
import numpy as np

from . import _ffi
from .parallel import PAD, allgather_merge_topk, merge_topk


class GraphShard:
    """One shard: a HipIndex over the vectors with global ids [base, base + index.N)."""

    def __init__(self, index, base):
        self.index = index
        self.base = int(base)


def globalize(local_ids, base):
    """Shard-local ids -> global ids, padding kept."""
    ids = np.asarray(local_ids, dtype=np.uint32)
    return np.where(ids == PAD, PAD, (ids.astype(np.uint64) + np.uint64(base)).astype(np.uint32))


class ShardedSearch:
    def __init__(self, shards, group=None, collective_device=None):
        """`shards`: the GraphShard objects this process owns (one per GPU in the 8-process layout).
        `group`: a torch.distributed process group, or None with torch.distributed uninitialised for one process."""
        self.shards = list(shards)
        self.group = group
        self.collective_device = collective_device

    def _distributed(self):
        try:
            import torch.distributed as dist
        except ImportError:
            return False
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def search_batch(self, queries, k, L=100, beam_width=8, mode=_ffi.MODE_M3, band_policy=0, flags=_ffi.F_USE_PQ):
        """Every local shard searches all queries; returns (global ids [nq,k] PAD-padded, distances [nq,k] NaN-padded,
        per-shard stats list). Raises if any shard reports a non-zero status (nothing is dropped silently)."""
        ids_l, dist_l, stats_l = [], [], []
        for sh in self.shards:
            ids, dist, cnt, st = sh.index.search_batch(queries, k, L=L, beam_width=beam_width, mode=mode,
                                                       band_policy=band_policy, flags=flags)
            if int(st["status"].max(initial=0)) != 0:
                raise _ffi.DiskragHipError(-1, f"shard at base {sh.base}: search status {int(st['status'].max())}")
            ids_l.append(globalize(ids, sh.base))
            dist_l.append(dist)
            stats_l.append(st)
        ids, dist = merge_topk(ids_l, dist_l, k)
        if self._distributed():
            # ids are already global: shard_base 0 in the exchange
            ids, dist = allgather_merge_topk(ids, dist, 0, k, group=self.group, device=self.collective_device)
        return ids, dist, stats_l
