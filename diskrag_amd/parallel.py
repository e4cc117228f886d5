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
    all-gather of nq*k*(4+4) bytes per rank and merged in canonical (distance, id) order (`allgather_merge_topk`).
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


def allgather_merge_topk(local_ids, local_dist, shard_base, k, group=None, device=None):
    """Graph-sharded merge: local ids are shard-local; adds `shard_base`, all-gathers every rank's (ids, dist) with
    one collective each and merges. Needs an initialised torch.distributed process group."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    gids = np.where(local_ids == PAD, PAD, (local_ids.astype(np.uint64) + np.uint64(shard_base)).astype(np.uint32))
    t_ids = torch.from_numpy(gids.astype(np.int64))
    t_dist = torch.from_numpy(np.ascontiguousarray(local_dist, dtype=np.float32))
    if device is not None:
        t_ids, t_dist = t_ids.to(device), t_dist.to(device)
    all_ids = [torch.empty_like(t_ids) for _ in range(world)]
    all_dist = [torch.empty_like(t_dist) for _ in range(world)]
    dist.all_gather(all_ids, t_ids, group=group)
    dist.all_gather(all_dist, t_dist, group=group)
    return merge_topk([a.cpu().numpy().astype(np.uint32) for a in all_ids], [a.cpu().numpy() for a in all_dist], k)


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
