"""TEST INFRASTRUCTURE: the torch.distributed (gloo) twin of the graph-sharded exchange and of the query-sharded gather -- what
tests/test_parallel_gloo.py runs with two CPU ranks to hold the sharding / merge / failure PROTOCOL across processes without a GPU.
The product (diskrag_amd/) imports no torch: its exchange is csrc/comm.inc (RCCL called from the library), exercised with real
processes by tests/test_gpu_sharded_procs.py; the numpy statements both share are diskrag_amd/parallel.py (merge_topk, pack_keys)."""
import numpy as np

from diskrag_amd import _ffi
from diskrag_amd.parallel import PAD, ShardExchangeError, merge_topk, pack_keys, unpack_keys
from diskrag_amd.sharded import globalize


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


class HostShardedSearch:
    """The host-logic twin of diskrag_amd.sharded.ShardedSearch for objects that are not device indexes: per-shard `search_batch`
    calls merged with numpy and exchanged through a torch.distributed group. The local phase may fail (a shard raises, a work area
    overflows): the rank still joins the one collective -- empty list, non-zero status word -- so that nobody is left waiting in it,
    and every rank raises afterwards (the failing rank its own error, the others ShardExchangeError): comm.inc's protocol."""

    def __init__(self, shards, group=None, collective_device=None):
        self.shards, self.group, self.collective_device = list(shards), group, collective_device

    def search_batch(self, queries, k, L=100, beam_width=8, mode=_ffi.MODE_M3, band_policy=0, flags=_ffi.F_USE_PQ):
        ids_l, dist_l, stats_l, err = [], [], [], None
        try:
            for sh in self.shards:
                ids, dist, cnt, st = sh.index.search_batch(queries, k, L=L, beam_width=beam_width, mode=mode, band_policy=band_policy, flags=flags)
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
            try:
                # ids are already global: shard_base 0 in the exchange
                ids, dist = allgather_merge_topk(ids, dist, 0, k, group=self.group, device=self.collective_device, local_status=0 if err is None else 1)
            except ShardExchangeError:
                if err is not None:
                    raise err
                raise
        return ids, dist, stats_l
