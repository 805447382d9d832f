"""Data-parallel batch generation across the GPUs of one node: one process per GPU (torchrun), full weight replica
per rank, samples sharded by rank exactly like the reference's batch drivers (eval/gen/gen_images_mp.py:127-130,
188-190: contiguous blocks; no tensor ever crosses a rank during sampling).

The one exchange the reference does not have: when every sample shares one conditioning context, rank ``src``
computes the prefill once and the KV cache is broadcast over RCCL/xGMI (``torch.distributed`` backend "nccl" == RCCL on
ROCm; "gloo" on CPU in the tests) as ONE flat bf16 buffer [L][2][rows][nkv*Dp] -- one collective per context, not one
per layer (xGMI is point-to-point: few large messages).  T2I: 32 tokens x 57 KB = 1.8 MB; edit: ~0.5 GB.
"""
import torch
import torch.distributed as dist

from .modeling.bagel.qwen2_navit import NaiveCache


def shard_range(n_items, rank, world_size):
    """Contiguous block of items for ``rank`` (gen_images_mp.py:188-190)."""
    per = (n_items + world_size - 1) // world_size
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def cache_to_flat(cache: NaiveCache):
    """-> (meta int64[4 + B], flat bf16 tensor [L, 2, total, width]) or (meta, None) for an empty cache."""
    L = cache.num_layers
    if cache.is_empty(0):
        return torch.tensor([L, 0, 0, 0], dtype=torch.int64), None
    lens = list(cache.lens(0))
    total = sum(lens)
    width = cache._nkv * cache._dp
    flat = torch.stack([torch.stack([cache._k[i][:total], cache._v[i][:total]]) for i in range(L)])
    meta = torch.tensor([L, cache._nkv, cache._hd, cache._dp] + lens, dtype=torch.int64)
    return meta, flat.contiguous()


def cache_from_flat(meta, flat):
    L, nkv, hd, dp = (int(x) for x in meta[:4])
    cache = NaiveCache(L)
    if flat is None or nkv == 0:
        return cache
    lens = [int(x) for x in meta[4:]]
    total = sum(lens)
    cache._nkv, cache._hd, cache._dp, cache._total = nkv, hd, dp, total
    for i in range(L):
        cache._k[i] = flat[i, 0].contiguous()
        cache._v[i] = flat[i, 1].contiguous()
        cache._lens[i] = list(lens)
    return cache


def broadcast_cache(cache, src=0, group=None, device=None, stats=None):
    """Every rank returns a NaiveCache equal to rank ``src``'s.  Two collectives: a small int64 header, then the payload.
    ``stats`` (optional dict) receives ``bytes`` = what this call moved to every rank."""
    rank = dist.get_rank(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    if rank == src:
        meta, flat = cache_to_flat(cache)
        hdr = torch.tensor([meta.numel()], dtype=torch.int64, device=device)
    else:
        meta, flat = None, None
        hdr = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(hdr, src, group=group)
    if rank != src:
        meta = torch.zeros(int(hdr.item()), dtype=torch.int64, device=device)
    else:
        meta = meta.to(device)
    dist.broadcast(meta, src, group=group)
    meta = meta.cpu()
    L, nkv, hd, dp = (int(x) for x in meta[:4])
    if stats is not None:
        stats["bytes"] = stats.get("bytes", 0) + 8 * (1 + meta.numel())
    if nkv == 0:
        return NaiveCache(L)
    total = int(meta[4:].sum())
    if rank != src:
        flat = torch.empty((L, 2, total, nkv * dp), dtype=torch.bfloat16, device=device)
    else:
        flat = flat.to(device)
    dist.broadcast(flat, src, group=group)
    if stats is not None:
        stats["bytes"] += flat.numel() * flat.element_size()
    return cache if rank == src else cache_from_flat(meta, flat)


def allreduce_renorm_sums(partials, group=None):
    """Optional batch-global CFG renorm: sum the two fp32 norm partials over ranks so an
    N-GPU batch reproduces the single-process 'global' renorm.  Off by default (the reference renorms per rank)."""
    dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=group)
    return partials
