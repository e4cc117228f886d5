"""Multi-GPU host logic: how queries and id ranges are cut, and the canonical merge -- the host
statement of what the device does (dr_sharded_search: device merge kernel + RCCL all-gather, include/diskrag_hip.h;
bench.py: query-sharded replicas with file barriers). Nothing on the GPU path imports torch; the `torch.distributed`
helpers below (`allgather_merge_topk`, `gather_rows`, `max_over_ranks`) exist for the 2-rank gloo test on CPU, which
checks the sharding and merge logic across processes without a GPU.

Two layouts:
  * query-sharded replicas (configs c2-c4): every rank holds the whole index, takes a contiguous slice of the
    batch (`shard_slice`) and searches it locally. No data-path collective; results are concatenated by rank order
    (`gather_rows` when one rank needs them all).
  * graph-sharded (c5): every rank holds a disjoint id range [base, base + n_local) with its own sub-graph; every
    query runs on every shard; the per-shard top-k lists (local ids + shard base) are exchanged with ONE
    all-gather of (nq*k + 1) 64-bit words per rank -- packed (distance, id) keys and the rank's status word -- and merged
    in canonical (distance, id) order (`allgather_merge_topk`). A rank whose local phase failed still joins the collective,
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


def allgather_merge_topk(local_ids, local_dist, shard_base, k, group=None, device=None, local_status=0):
    """Graph-sharded merge: local ids are shard-local; adds `shard_base`, packs the list into 64-bit keys, puts this rank's
    status word behind it and all-gathers the nq*k + 1 words of every rank with ONE collective, then merges. A rank calls
    this even when its local phase failed (`local_status` != 0, any lists): if any rank's status is non-zero every rank
    raises ShardExchangeError after the collective. Needs an initialised torch.distributed process group."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    local_ids = np.asarray(local_ids, dtype=np.uint32)
    gids = np.where(local_ids == PAD, PAD, (local_ids.astype(np.uint64) + np.uint64(shard_base)).astype(np.uint32))
    keys = pack_keys(gids, local_dist)
    if local_status:
        keys = np.full_like(keys, np.uint64(0xFFFFFFFFFFFFFFFF))
    nq, kk = keys.shape
    words = np.concatenate([keys.reshape(-1), np.array([local_status], dtype=np.uint64)])
    t = torch.from_numpy(words.view(np.int64).copy())      # (collectives have no unsigned 64-bit type: ship the bits)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    got = [o.cpu().numpy().view(np.uint64) for o in out]
    statuses = [int(g[-1]) for g in got]
    if any(statuses):
        raise ShardExchangeError(statuses)
    lists = [unpack_keys(g[:-1].reshape(nq, kk)) for g in got]
    return merge_topk([a for a, _ in lists], [b for _, b in lists], k)


def gather_rows(local_rows, group=None, device=None):
    """Concatenates per-rank row blocks (query-sharded results) on every rank, in rank order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rows = np.ascontiguousarray(local_rows)
    orig_dtype = rows.dtype
    if orig_dtype == np.uint32:          # collectives have no unsigned 32-bit type: ship the bits as int32
        rows = rows.view(np.int32)
    t = torch.from_numpy(rows)
    if device is not None:
        t = t.to(device)
    counts = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device), group=group)
    mx = int(max(c.item() for c in counts))
    padded = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    padded[:t.shape[0]] = t
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    res = np.concatenate([o[:int(c.item())].cpu().numpy() for o, c in zip(out, counts)], axis=0)
    return res.view(np.uint32) if orig_dtype == np.uint32 else res


def max_over_ranks(value, group=None, device=None):
    """Slowest rank's time: the bench divides the job's queries by this."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
