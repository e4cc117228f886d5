"""Multi-GPU host logic: how queries and id ranges are cut, and the canonical merge -- the host
statement of what the device does (dr_sharded_search: device merge kernel + RCCL all-gather, include/diskrag_hip.h;
bench.py: query-sharded replicas with file barriers). numpy only: the package imports no torch anywhere -- the
`torch.distributed` twin of the exchange that the 2-rank gloo test runs on CPU lives with the tests (tests/gloo_twin.py).

Two layouts:
  * query-sharded replicas (configs c2-c4): every rank holds the whole index, takes a contiguous slice of the
    batch (`shard_slice`) and searches it locally. No data-path collective; results are concatenated by rank order
    (rank 0 of bench.py reads them from the ranks' files).
  * graph-sharded (c5): every rank holds a disjoint id range [base, base + n_local) with its own sub-graph; every
    query runs on every shard; the per-shard top-k lists (local ids + shard base) are exchanged with ONE
    all-gather of (nq*k + 1) 64-bit words per rank -- packed (distance, id) keys and the rank's status word -- and merged
    in canonical (distance, id) order. A rank whose local phase failed still joins the collective,
    with an empty list and a non-zero status word: the call then fails on EVERY rank (`ShardExchangeError`) instead of
    leaving the others blocked in the collective (dr_sharded_submit, csrc/comm.inc, is the device statement of the same
    protocol).
"""
import numpy as np

PAD = np.uint32(0xFFFFFFFF)


def shard_slice(n_items, world_size, rank):
    """Contiguous slice of `n_items` for `rank` (sizes differ by at most one, earlier ranks get the extra)."""
    base, extra = divmod(int(n_items), int(world_size))
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def merge_topk(ids_list, dist_list, k):
    """k-way merge of per-shard results (each ids[nq,k_i] global ids with PAD, dist[nq,k_i] with NaN padding) in
    canonical (distance ascending, id ascending) order. Returns ids[nq,k] (PAD padded), dist[nq,k] (NaN padded)."""
    ids = np.concatenate([np.asarray(a, dtype=np.uint32) for a in ids_list], axis=1)
    dist = np.concatenate([np.asarray(a, dtype=np.float32) for a in dist_list], axis=1)
    nq = ids.shape[0]
    key_d = np.where(ids == PAD, np.float32(np.inf), dist)
    key_d = np.where(np.isnan(key_d), np.float32(np.inf), key_d)
    order = np.lexsort((ids, key_d), axis=1)[:, :k]          # last key is primary: distance, then id
    rows = np.arange(nq)[:, None]
    out_ids = ids[rows, order]
    out_dist = dist[rows, order]
    invalid = np.isinf(key_d[rows, order])
    out_ids[invalid] = PAD
    out_dist[invalid] = np.nan
    if out_ids.shape[1] < k:
        pad = k - out_ids.shape[1]
        out_ids = np.concatenate([out_ids, np.full((nq, pad), PAD, dtype=np.uint32)], axis=1)
        out_dist = np.concatenate([out_dist, np.full((nq, pad), np.nan, dtype=np.float32)], axis=1)
    return out_ids, out_dist


class ShardExchangeError(RuntimeError):
    """Another rank of a graph-sharded search failed its local phase: the call fails on every rank (DR_E_REMOTE)."""

    def __init__(self, statuses):
        bad = {r: int(s) for r, s in enumerate(statuses) if int(s) != 0}
        super().__init__("sharded search: rank(s) %s failed their local phase (status %s); every rank fails this call"
                         % (sorted(bad), bad))
        self.statuses = [int(s) for s in statuses]


def pack_keys(ids, dist):
    """(ids u32, dist f32) -> uint64 keys that order like (distance ascending, id ascending): an order-preserving map of
    the distance's bits << 32 | id; empty slots (PAD), NaN and +inf -> ~0 (sort last). -0.0 packs as +0.0. This is the
    form the lists travel in (csrc/comm.inc merge_topk_kernel, key_ord)."""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    d = np.ascontiguousarray(dist, dtype=np.float32)
    b = np.where(d == 0, np.float32(0.0), d).view(np.uint32).astype(np.uint64)
    o = np.where(b >> np.uint64(31) != 0, b ^ np.uint64(0xFFFFFFFF), b ^ np.uint64(0x80000000))
    keys = (o << np.uint64(32)) | ids.astype(np.uint64)
    invalid = (ids == PAD) | np.isnan(d) | (d == np.float32(np.inf))
    return np.where(invalid, np.uint64(0xFFFFFFFFFFFFFFFF), keys)


def unpack_keys(keys):
    """inverse of pack_keys: (ids, dist); ~0 -> (PAD, NaN)"""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    o = (keys >> np.uint64(32)).astype(np.uint32)
    b = np.where(o >> np.uint32(31) != 0, o ^ np.uint32(0x80000000), o ^ np.uint32(0xFFFFFFFF)).astype(np.uint32)
    ids = (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    d = b.view(np.float32).copy()
    empty = keys == np.uint64(0xFFFFFFFFFFFFFFFF)
    ids[empty] = PAD
    d[empty] = np.nan
    return ids, d
