"""GPU parity of TaylorSeer step skipping (generate_image(enable_taylorseer=True); modeling/cache_utils/taylorseer.py):
state machine + kernels bit-exact against the oracle's restatement, end-to-end latents against the reference's goldens."""
import pytest
import torch

from tests.test_ops_gpu import rnd

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16
DEV = "cuda"


# ------------------------------------------------------------------------------------------------------------
# TaylorSeer step skipping (generate_image(enable_taylorseer=True); modeling/cache_utils/taylorseer.py)
# ------------------------------------------------------------------------------------------------------------
def test_taylorseer_state_machine_and_kernels_bit_exact():
    """Product state + HIP kernels against the oracle's restatement (pinned bit-exact on the reference) on the SAME
    feature sequence: schedule, finite differences (bf16 chain, /distance) and the bf16 Taylor sum must agree bit for bit."""
    from bagel_amd.modeling.cache_utils.taylorseer import TaylorSeerState
    from oracle import bagel_oracle as OR
    rows, cols, steps = 37, 264, 24
    prod, ora = TaylorSeerState(steps + 1), OR.TaylorState(steps + 1)
    g = torch.Generator().manual_seed(3)
    base, drift = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g)
    kinds = ""
    for s in range(steps):
        typ = prod.next_type()
        assert typ == OR.taylor_cal_type(ora)
        kinds += typ[0]
        if typ == "full":
            t = s / steps
            feat = (base * (1 + 0.5 * t) + drift * t * t + 0.01 * torch.randn(rows, cols, generator=g)).to(BF16)
            prod.update(feat.to(DEV))
            OR.taylor_derivative_approximation(ora, 0, feat)
            assert prod.n_factors == len(ora.factors[0])
            for i in range(prod.n_factors):
                assert torch.equal(prod._bufs[i].cpu().view(torch.int16), ora.factors[0][i].view(torch.int16)), f"step {s} order {i}"
        else:
            out = torch.empty((rows, cols), dtype=BF16, device=DEV)
            prod.eval_into(out)
            ref = OR.taylor_formula(ora, 0)
            assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16)), f"Taylor step {s}"
        prod.advance()
        ora.step += 1
    assert kinds == "fffffTTfTTfTTfTTfTTfTTfT"
    assert prod.n_factors == 7, "orders must saturate at max_order + 1"


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_generate_image_taylorseer_matches_reference(golden, name):
    """End to end against the reference's enable_taylorseer=True latents.  Tolerance: the extrapolation amplifies the
    bf16 accumulation-order noise of the cached features by ~1.8x (measured on the oracle under a 2^-9 input perturbation),
    so rel-L2 <= 4e-2 (2x the plain sampler's 2e-2); and the run must be DISTINGUISHABLE from the plain sampler: the
    displacement (taylorseer - plain) must match the reference's displacement."""
    from oracle.configs import TINY, TINY_D128, NEW_TOKEN_IDS_TINY, StubTokenizer
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from tests.util_models import product_model
    cfg = {"tiny": TINY, "tiny_d128": TINY_D128}[name]
    g = golden(f"{name}_taylorseer")
    model, _ = product_model(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(NaiveCache(L), **gi)
    c = g["cfg_inputs"]
    ckw = dict(cfg_text_packed_position_ids=c["cfg_packed_position_ids"], cfg_text_packed_query_indexes=c["cfg_packed_query_indexes"],
               cfg_text_key_values_lens=c["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=c["cfg_packed_key_value_indexes"])

    def rel(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return ((a - b).norm() / b.norm()).item()
    for tag, run in g["runs"].items():
        kw = run["gen_kwargs"]
        lat = model.generate_image(past_key_values=cache, cfg_text_past_key_values=NaiveCache(L), enable_taylorseer=True, **ckw, **kw,
                                   **g["latent_inputs"])
        states = model._last_taylor_states
        n_fwd = kw["num_timesteps"] - 1
        assert states[0].full_steps + states[0].taylor_steps == n_fwd
        assert states[0].full_steps == 5 + (n_fwd - 5) // 3 and states[0].taylor_steps > 0
        if tag == "partial_cfg":
            assert 0 < states[1].full_steps + states[1].taylor_steps < n_fwd, "cfg-text stream must keep its own step counter"
        assert states[2].full_steps == 0
        plain = model.generate_image(past_key_values=cache, cfg_text_past_key_values=NaiveCache(L), **ckw, **kw, **g["latent_inputs"])
        for a, b, p, gp in zip(lat, run["latents"], plain, run["latents_plain_sampler"]):
            assert torch.isfinite(a).all()
            assert rel(a, b) <= 4e-2, f"{tag}: rel_l2 {rel(a, b):.4g} vs the reference TaylorSeer latents"
            assert rel(p, gp) <= 2e-2
        if run["rel_dev_from_plain_sampler"] >= 1.5e-2:
            d_gpu = torch.cat([(a - p).float().cpu().flatten() for a, p in zip(lat, plain)])
            d_ref = torch.cat([(b - gp).float().flatten() for b, gp in zip(run["latents"], run["latents_plain_sampler"])])
            cos = torch.dot(d_gpu, d_ref) / (d_gpu.norm() * d_ref.norm())
            assert cos >= 0.8 and 0.6 <= (d_gpu.norm() / d_ref.norm()).item() <= 1.6, \
                f"{tag}: TaylorSeer displacement does not match the reference's (cos {cos:.3f}, ratio {(d_gpu.norm() / d_ref.norm()).item():.3f})"
