"""GPU check of the stream-batched CFG forward, promoted to the default in round 2 after measurement (profiles/r02_stream_batch.log):
bit-identical to sequential forwards without the marker-row side path, golden tolerance with it."""
import pytest
import torch

pytestmark = [pytest.mark.gpu]


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_stream_batched_cfg_on_the_gpu(golden, name):
    """Batched (marker rows inside the tile GEMM) == sequential bit for bit; batched + marker side path within the golden
    tolerance (the marker rows see the skinny GEMM's accumulation order)."""
    from oracle.configs import TINY, TINY_D128, NEW_TOKEN_IDS_TINY, StubTokenizer
    from tests.test_model_gpu import cfg_kwargs, new_cache, rel_l2
    from tests.util_models import product_model
    cfg = {"tiny": TINY, "tiny_d128": TINY_D128}[name]
    g = golden(f"{name}_t2i")
    model, _ = product_model(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)

    def run():
        return model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **g["gen_kwargs"],
                                    **g["latent_inputs"])
    try:
        model.cfg_batched = False
        seq = run()
        model.cfg_batched, model.und_side_path = True, False
        bat = run()
        assert all(torch.equal(a, b) for a, b in zip(seq, bat))
        model.und_side_path = True
        side = run()
        for a, b in zip(side, g["latents"]):
            assert rel_l2(a, b) <= 2e-2
    finally:
        model.cfg_batched, model.und_side_path = True, True
