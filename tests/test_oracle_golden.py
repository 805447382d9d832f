"""CPU: the oracle restatement reproduces the reference's golden vectors bit-for-bit.

The fixtures under tests/golden/ were produced by oracle/make_golden.py from the UNMODIFIED reference
(/root/reference) -- this re-check runs on any box (no reference tree needed)."""
import pytest
import torch

from oracle import bagel_oracle as O
from oracle import packers as P
from oracle.configs import TINY, TINY_D128, TINY_DENSE, TINY_MOE, TINY_ROPE, NEW_TOKEN_IDS_TINY, StubTokenizer
from tests.util_models import oracle_weights

CFGS = {"tiny": TINY, "tiny_d128": TINY_D128, "tiny_rope": TINY_ROPE, "tiny_dense": TINY_DENSE, "tiny_moe": TINY_MOE}


def _cache(keys, vals):
    c = O.OracleCache(len(keys))
    for i, (k, v) in enumerate(zip(keys, vals)):
        c.key_cache[i], c.value_cache[i] = k.clone(), v.clone()
    return c


def _cfgd(cache, d):
    return dict(cache=cache, position_ids=d["cfg_packed_position_ids"], query_indexes=d["cfg_packed_query_indexes"],
                key_values_lens=d["cfg_key_values_lens"], key_value_indexes=d["cfg_packed_key_value_indexes"])


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_t2i_matches_reference(golden, name):
    cfg = CFGS[name]
    g = golden(f"{name}_t2i")
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, newlens, newrope = P.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    for k in gi:
        assert torch.equal(gi[k], g["prompt_inputs"][k]), k
    cache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi)
    for i in range(L):
        assert torch.equal(cache.key_cache[i], g["key_cache"][i])
        assert torch.equal(cache.value_cache[i], g["value_cache"][i])
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(O.OracleCache(L), g["cfg_inputs"]),
                           **g["gen_kwargs"])
    for a, b in zip(lat, g["latents"]):
        assert torch.equal(a, b)
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(O.OracleCache(L), g["cfg_inputs"]),
                           **g["gen_kwargs_channel"])
    for a, b in zip(lat, g["latents_channel"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_edit_and_understanding_match_reference(golden, name):
    cfg = CFGS[name]
    g = golden(f"{name}_editund")
    W, VW = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    cache = O.forward_cache_update_vae(W, cfg, VW, O.OracleCache(L), sample_noise=g["enc_noise"], **g["vae_inputs"])
    cache = O.forward_cache_update_vit(W, cfg, cache, **g["vit_inputs"])
    for i in range(L):
        assert torch.equal(cache.key_cache[i], g["key_cache_img"][i])
    cfg_text_cache = cache.clone()
    l1, l2, l3, l4 = g["lens"]
    r1, r2, r3, r4 = g["ropes"]
    pi = P.prepare_prompts(l2, r2, [g["prompt"]], tok, NEW_TOKEN_IDS_TINY)[0]
    cache = O.forward_cache_update_text(W, cfg, cache, **pi)
    for i in range(L):
        assert torch.equal(cache.key_cache[i], g["key_cache"][i])
        assert torch.equal(cache.value_cache[i], g["value_cache"][i])
    pi2 = P.prepare_prompts([0], [0], [g["prompt"]], tok, NEW_TOKEN_IDS_TINY)[0]
    cimg = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **pi2)
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(cfg_text_cache, g["cfg_text_inputs"]),
                           cfg_img=_cfgd(cimg, g["cfg_img_inputs"]), **g["gen_kwargs"])
    assert torch.equal(lat[0], g["latents"][0])
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(cfg_text_cache, g["cfg_text_inputs"]),
                           cfg_img=_cfgd(cimg, g["cfg_img_inputs"]), **g["gen_kwargs_global"])
    assert torch.equal(lat[0], g["latents_global"][0])
    si = g["start_inputs"]
    toks, logits = O.generate_text(W, cfg, cache.clone(), si["packed_key_value_indexes"], si["key_values_lens"],
                                   si["packed_start_tokens"], si["packed_query_position_ids"], 8, return_logits=True)
    assert torch.equal(toks, g["tokens"])
    assert torch.equal(logits, g["logits"])


def test_vae_matches_reference(golden):
    g = golden("tiny_vae")
    _, VW = oracle_weights(TINY)
    assert torch.equal(O.vae_decode(VW, TINY["vae"], g["z"]), g["decoded"])
    assert torch.equal(O.vae_encode(VW, TINY["vae"], g["x"], g["enc_noise"]), g["encoded"])
    h, w = 16 // 2, 24 // 2
    assert torch.equal(O.latent_to_image_uint8(VW, TINY["vae"], g["packed_latent"], h * 16, w * 16, 16, 2, 16),
                       g["image_u8"])


@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_rope"])      # tiny_rope: the 2-D RoPE variant (config.rope=True)
def test_siglip_matches_reference(golden, name):
    cfg = CFGS[name]
    g = golden(f"{name}_siglip")
    W, _ = oracle_weights(cfg)
    out = O.siglip_forward(W, cfg["vit"], g["tokens"], g["pos"], g["cu"], 35)
    assert torch.equal(out, g["out"])
    assert torch.equal(O.connector(W, out), g["connector_out"])


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_taylorseer_matches_reference(golden, name):
    """enable_taylorseer=True (bagel.py:678-689; taylorseer.py): schedule, finite differences and the bf16 Taylor sum are
    bit-exact, and evaluating only the LAST layer's extrapolation (what the MI355X engine does) is the same function."""
    cfg = CFGS[name]
    g = golden(f"{name}_taylorseer")
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = P.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi)
    for tag, run in g["runs"].items():
        for last_only in (False, True):
            lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(O.OracleCache(L), g["cfg_inputs"]),
                                   enable_taylorseer=True, taylor_last_layer_only=last_only, **run["gen_kwargs"])
            for a, b in zip(lat, run["latents"]):
                assert torch.equal(a, b), (tag, last_only)


def test_taylorseer_schedule_known_answer():
    """cal_type (taylorseer.py:83-117) with the reference constants: 5 full steps, then T T F repeating; the
    finite-difference distance is the gap between the last two full steps."""
    st = O.TaylorState(50)
    kinds = []
    for _ in range(49):
        kinds.append("F" if O.taylor_cal_type(st) == "full" else "T")
        st.step += 1
    assert "".join(kinds) == "FFFFF" + "TTF" * 14 + "TT"
    assert kinds.count("F") == 19
    assert st.activated_steps[:8] == [0, 0, 1, 2, 3, 4, 7, 10]


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_training_forward_matches_reference(golden, name):
    """Bagel.forward (bagel.py:101-229) with nested masks: per-token MSE and CE losses bit-exact; the mask rule restated."""
    cfg = CFGS[name]
    g = golden(f"{name}_train")
    W, _ = oracle_weights(cfg)
    out = O.bagel_forward_train(W, cfg, g["batch"], g["noise"], timestep_shift=cfg["bagel"]["timestep_shift"])
    assert torch.equal(out["mse"], g["mse"]) and torch.equal(out["ce"], g["ce"])
    # the masks in the fixture came from the reference's prepare_attention_mask_per_sample
    i = 0
    for n, m in zip(g["batch"]["sample_lens"], g["batch"]["nested_attention_masks"]):
        lens, modes, tot = [], [], 0
        while tot < n:
            lens.append(g["split_lens"][i]); modes.append(g["attn_modes"][i]); tot += lens[-1]; i += 1
        assert torch.equal(O.attention_mask_per_sample(lens, modes), m)


@pytest.mark.parametrize("name", ["tiny_dense", "tiny_moe"])
def test_dense_and_moe_layer_kinds_match_reference(golden, name):
    """Decoder_layer_dict alternates (qwen2_navit.py:936-940): Qwen2DecoderLayer and Qwen2MoEDecoderLayer, bit-exact."""
    cfg = CFGS[name]
    g = golden(f"{name}_t2i")
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = P.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi)
    for i in range(L):
        assert torch.equal(cache.key_cache[i], g["key_cache"][i]) and torch.equal(cache.value_cache[i], g["value_cache"][i])
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(O.OracleCache(L), g["cfg_inputs"]), **g["gen_kwargs"])
    for a, b in zip(lat, g["latents"]):
        assert torch.equal(a, b)
