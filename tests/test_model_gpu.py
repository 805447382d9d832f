"""GPU end-to-end parity of the drop-in model against (a) the golden vectors produced by the UNMODIFIED reference and
(b) the oracle, through the reference's own method surface (prepare_* -> forward_cache_update_* -> generate_*).

Tolerances (stated per SURVEY.md 8c): integer tensors bit-exact; bf16 KV caches after L layers: rel-L2 <= 1e-2;
final fp32 latents after the Euler loop: rel-L2 <= 2e-2; greedy tokens of the tiny models: exact.
"""
import copy

import pytest
import torch

from oracle.configs import TINY, TINY_D128, TINY_DENSE, TINY_MOE, TINY_ROPE, NEW_TOKEN_IDS_TINY, StubTokenizer
from tests.util_models import product_model

pytestmark = pytest.mark.gpu
CFGS = {"tiny": TINY, "tiny_d128": TINY_D128, "tiny_rope": TINY_ROPE, "tiny_dense": TINY_DENSE, "tiny_moe": TINY_MOE}


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def check(a, b, tol, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a.float()).all(), f"{what}: non-finite"
    e = rel_l2(a, b)
    assert e <= tol, f"{what}: rel_l2 {e:.4g} > {tol}"
    return e


def new_cache(cfg):
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    return NaiveCache(cfg["llm"]["num_hidden_layers"])


def cfg_kwargs(tag, cache, d):
    return {f"{tag}_past_key_values": cache, f"{tag}_packed_position_ids": d["cfg_packed_position_ids"],
            f"{tag}_packed_query_indexes": d["cfg_packed_query_indexes"], f"{tag}_key_values_lens": d["cfg_key_values_lens"],
            f"{tag}_packed_key_value_indexes": d["cfg_packed_key_value_indexes"]}


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_text_to_image_matches_reference(golden, name):
    cfg = CFGS[name]
    g = golden(f"{name}_t2i")
    model, _ = product_model(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, newlens, newrope = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    for i in range(L):
        check(cache.key_cache[i], g["key_cache"][i], 1e-2, f"K cache layer {i}")
        check(cache.value_cache[i], g["value_cache"][i], 1e-2, f"V cache layer {i}")
    assert cache.seq_lens == g["key_cache"][0].shape[0]
    for kw, key in ((g["gen_kwargs"], "latents"), (g["gen_kwargs_channel"], "latents_channel")):
        lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **kw,
                                   **g["latent_inputs"])
        assert len(lat) == 2 and lat[0].dtype == torch.float32
        for a, b in zip(lat, g[key]):
            check(a, b, 2e-2, f"{key}")
    # determinism: same inputs -> bit-identical latents
    lat2 = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **g["gen_kwargs_channel"],
                                **g["latent_inputs"])
    assert all(torch.equal(a, b) for a, b in zip(lat, lat2))


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_first_step_velocity(golden, name):
    """_forward_flow (one timestep, CFG text 4.0, global renorm) against the oracle's first-step velocity."""
    cfg = CFGS[name]
    g = golden(f"{name}_t2i")
    model, _ = product_model(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    li = dict(g["latent_inputs"])
    x = li.pop("packed_init_noises")
    ts, _ = model.flow_schedule(5, 3.0)
    c = g["cfg_inputs"]
    v = model._forward_flow(x_t=x, timestep=torch.tensor([ts[0]] * x.shape[0]), past_key_values=cache, cfg_text_scale=4.0,
                            cfg_renorm_type="global", cfg_text_past_key_values=new_cache(cfg),
                            cfg_text_packed_position_ids=c["cfg_packed_position_ids"],
                            cfg_text_packed_query_indexes=c["cfg_packed_query_indexes"],
                            cfg_text_key_values_lens=c["cfg_key_values_lens"],
                            cfg_text_packed_key_value_indexes=c["cfg_packed_key_value_indexes"], **li)
    check(v, g["v_first_step"], 2e-2, "first-step velocity")


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_vit_text_context_understanding(golden, name):
    """ViT prefill + text prefill (on a hand-made VAE-free context) and greedy decode; image-edit sampling with 3 forwards."""
    cfg = CFGS[name]
    g = golden(f"{name}_editund")
    model, vae = product_model(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ident = lambda t: t  # noqa: E731

    class FixedNoiseVae:   # the reference draws randn_like inside encode; feed the recorded draw
        def encode(self, x):
            return vae.encode(x, sample_noise=g["enc_noise"])
    vi, l1, r1 = model.prepare_vae_images([0], [0], [g["img_vae"]], ident, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_vae(FixedNoiseVae(), new_cache(cfg), **vi)
    ti, l2, r2 = model.prepare_vit_images(l1, r1, [g["img_vit"]], ident, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_vit(cache, **ti)
    for i in range(L):
        check(cache.key_cache[i], g["key_cache_img"][i], 1.5e-2, f"K cache (vae+vit) layer {i}")
    cfg_text_cache = copy.deepcopy(cache)
    pi, l3, r3 = model.prepare_prompts(l2, r2, [g["prompt"]], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(cache, **pi)
    for i in range(L):
        check(cache.key_cache[i], g["key_cache"][i], 1.5e-2, f"K cache (vae+vit+text) layer {i}")
        check(cache.value_cache[i], g["value_cache"][i], 1.5e-2, f"V cache layer {i}")
    assert cfg_text_cache.seq_lens == g["key_cache_img"][0].shape[0], "deepcopy must not alias the appended cache"
    pi2, l4, r4 = model.prepare_prompts([0], [0], [g["prompt"]], tok, NEW_TOKEN_IDS_TINY)
    cimg = model.forward_cache_update_text(new_cache(cfg), **pi2)
    for kw, key in ((g["gen_kwargs"], "latents"), (g["gen_kwargs_global"], "latents_global")):
        lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", cfg_text_cache, g["cfg_text_inputs"]),
                                   **cfg_kwargs("cfg_img", cimg, g["cfg_img_inputs"]), **kw, **g["latent_inputs"])
        check(lat[0], g[key][0], 3e-2, f"edit {key}")
    toks = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=8, do_sample=False, end_token_id=None,
                               **g["start_inputs"])
    assert toks.shape == g["tokens"].shape and toks.dtype == torch.int64
    # Greedy ids must equal the reference's up to the first step whose reference logits hold a NEAR TIE between the two
    # candidates (random-init weights make those frequent): there, bf16 accumulation-order noise legitimately flips the
    # argmax, the prefixes diverge and later steps are incomparable.  Near tie := the reference's own logit of our token
    # is within 2^-6 * max|logit| (two bf16 ulps at the top of the range) of its maximum.
    ours, ref, ref_logits = toks.cpu(), g["tokens"], g["logits"].float()
    for s in range(1, ref.shape[0]):
        if torch.equal(ours[s], ref[s]):
            continue
        row = ref_logits[s - 1]          # logits that chose token s, shape (B, V)
        gap = row.max(-1).values - row.gather(-1, ours[s].view(-1, 1)).squeeze(-1)
        tol = row.abs().max().item() * 2.0 ** -6
        assert (gap <= tol).all(), (f"greedy tokens differ at step {s} without a near tie (gap {gap.tolist()} > {tol:.4g}): "
                                    f"{ours.tolist()} vs {ref.tolist()}")
        break


@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_rope"])      # tiny_rope: SigLIP 2-D RoPE (siglip_navit.py:102-142)
def test_siglip_encoder(golden, name):
    cfg = CFGS[name]
    g = golden(f"{name}_siglip")
    model, _ = product_model(cfg)
    out = model.vit_model(packed_pixel_values=g["tokens"], packed_flattened_position_ids=g["pos"], cu_seqlens=g["cu"], max_seqlen=35)
    check(out, g["out"], 1e-2, "siglip features")


def test_vae_matches_reference(golden):
    """fp32 VAE: the MFMA is an exact fp32 fma chain, only the summation order differs from the CPU convolution."""
    g = golden("tiny_vae")
    _, vae = product_model(TINY)
    dec = vae.decode(g["z"])
    err = (dec.cpu() - g["decoded"]).abs().max().item() / g["decoded"].abs().max().item()
    assert dec.shape == g["decoded"].shape and err < 1e-4, f"vae.decode max rel error {err:.3g}"
    enc = vae.encode(g["x"], sample_noise=g["enc_noise"])
    err = (enc.cpu() - g["encoded"]).abs().max().item() / g["encoded"].abs().max().item()
    assert enc.shape == g["encoded"].shape and err < 1e-4, f"vae.encode max rel error {err:.3g}"
    # packed latent -> uint8 image through the inferencer's decode path (truncating cast): off-by-one allowed on ties
    from bagel_amd.inferencer import InterleaveInferencer
    model, _ = product_model(TINY)
    inf = InterleaveInferencer(model, vae, None, None, None, None)
    img = vae.decode(inf.latent_to_chw(g["packed_latent"].cuda(), (8 * 16, 12 * 16)))
    u8 = ((img * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255).to(torch.uint8).cpu()
    diff = (u8.int() - g["image_u8"].int()).abs()
    assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 0.01


@pytest.mark.parametrize("name", ["tiny_dense", "tiny_moe"])
def test_dense_and_moe_layer_kinds(golden, name):
    """Qwen2DecoderLayer / Qwen2MoEDecoderLayer run as degenerate routings of the MoT plan (no routing at all /
    shared attention with the und cast points + per-modality MLP and final norm) -- vs the reference's goldens."""
    cfg = CFGS[name]
    g = golden(f"{name}_t2i")
    model, _ = product_model(cfg)
    assert model.use_moe == (name == "tiny_moe")
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    for i in range(L):
        check(cache.key_cache[i], g["key_cache"][i], 1e-2, f"K cache layer {i}")
        check(cache.value_cache[i], g["value_cache"][i], 1e-2, f"V cache layer {i}")
    lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **g["gen_kwargs"],
                               **g["latent_inputs"])
    for a, b in zip(lat, g["latents"]):
        check(a, b, 2e-2, "latents")


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_distance_from_the_fp32_master_weights_reference(golden, name):
    """eval/gen/gen_images_mp.py:165-176 keeps FP32 master weights under autocast (fp32 residual stream); the product -- like app.py:105-113 --
    holds bf16 weights and a bf16 residual stream and casts an fp32 checkpoint once.  The two REFERENCE precisions themselves differ by
    ~1e-2 rel-L2 on the final latents (tests/golden/<cfg>_t2i_fp32master.pt, oracle/make_golden_fp32master.py); the product must stay within
    the plain tolerance (2e-2) plus that deviation of the fp32-master run -- the documented numerics difference against that driver."""
    cfg = CFGS[name]
    g, gm = golden(f"{name}_t2i"), golden(f"{name}_t2i_fp32master")
    dev = gm["deviation_of_bf16_weights_reference"]
    assert 1e-3 < dev["latents"] < 2e-2 and 1e-3 < dev["latents_channel"] < 2e-2
    model, _ = product_model(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    for kw, key in ((g["gen_kwargs"], "latents"), (g["gen_kwargs_channel"], "latents_channel")):
        lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **kw, **g["latent_inputs"])
        for a, b in zip(lat, gm[key]):
            check(a, b, 2e-2 + dev[key], f"{key} vs the fp32-master reference run")
