"""GPU side of the data-parallel driver: the conditioning KV cache that a non-source rank receives --
rebuilt from the flat broadcast payload -- must drive ``generate_image`` to bit-identical latents, and the broadcast itself
must run over RCCL (backend "nccl") on device tensors.  One GPU is all a test box has, so the RCCL group has one rank;
the two-rank exchange is covered on CPU/gloo by tests/test_parallel_cpu.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle.configs import TINY, TINY_D128, NEW_TOKEN_IDS_TINY, StubTokenizer
from tests.test_model_gpu import cfg_kwargs, new_cache
from tests.util_models import product_model

pytestmark = pytest.mark.gpu


def _context(cfg, golden):
    g = golden(f"{cfg['name']}_t2i")
    model, _ = product_model(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)

    def run(c):
        return model.generate_image(past_key_values=c, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **g["gen_kwargs"],
                                    **g["latent_inputs"])
    return cache, run


@pytest.mark.parametrize("cfg", [TINY, TINY_D128], ids=lambda c: c["name"])
def test_received_cache_drives_generate_image_identically(golden, cfg):
    from bagel_amd.parallel import cache_from_flat, cache_to_flat
    cache, run = _context(cfg, golden)
    want = run(cache)
    meta, flat = cache_to_flat(cache)
    assert flat.is_cuda and flat.dtype == torch.bfloat16 and flat.shape[:2] == (cfg["llm"]["num_hidden_layers"], 2)
    got = run(cache_from_flat(meta, flat.clone()))
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_broadcast_cache_over_rccl_single_rank(golden):
    from bagel_amd.parallel import broadcast_cache
    cache, run = _context(TINY, golden)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = broadcast_cache(cache, src=0)
        empty = broadcast_cache(new_cache(TINY), src=0)
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    assert got.seq_lens == cache.seq_lens and empty.seq_lens == 0 and float(t.item()) == 1.5
    for a, b in zip(run(got), run(cache)):
        assert torch.equal(a, b)
