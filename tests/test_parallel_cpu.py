"""world_size-2 gloo test of the DP driver pieces (sharding + conditioning-KV broadcast) on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bagel_amd.parallel import shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 32, 33):
        for ws in (1, 2, 4, 8):
            seen = []
            for r in range(ws):
                lo, hi = shard_range(n, r, ws)
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from bagel_amd.parallel import broadcast_cache
    L, nkv, hd, dp = 3, 2, 32, 64
    lens = [5, 9]
    cache = NaiveCache(L)
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        cache._nkv, cache._hd, cache._dp, cache._total = nkv, hd, dp, sum(lens)
        for i in range(L):
            cache._k[i] = torch.randn(64, nkv * dp, generator=g).to(torch.bfloat16)   # capacity > used rows
            cache._v[i] = torch.randn(64, nkv * dp, generator=g).to(torch.bfloat16)
            cache._lens[i] = list(lens)
    got = broadcast_cache(cache, src=0)
    g = torch.Generator().manual_seed(0)
    ok = got.seq_lens == sum(lens) and got.lens(0) == lens
    for i in range(L):
        k = torch.randn(64, nkv * dp, generator=g).to(torch.bfloat16)[:14]
        v = torch.randn(64, nkv * dp, generator=g).to(torch.bfloat16)[:14]
        ok = ok and torch.equal(got.key_cache[i], k.view(14, nkv, dp)[..., :hd]) and torch.equal(got.value_cache[i], v.view(14, nkv, dp)[..., :hd])
    empty = broadcast_cache(NaiveCache(2), src=0)
    ok = ok and empty.seq_lens == 0
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_cache_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == {0: True, 1: True}
