"""CPU: the oracle restatement reproduces the reference's golden vectors.

The fixtures under tests/golden/ were produced by oracle/make_golden.py from the UNMODIFIED reference
(/root/reference) -- this re-check runs on any box (no reference tree needed).

Integer tensors (packer outputs, token ids) are compared bit-for-bit.  Float tensors are bit-for-bit on a host whose
torch CPU bf16 matmul backend is the one the fixtures were generated with (oneDNN AMX-bf16 Xeon); on any other host
(avx512 without AMX, Zen) the REFERENCE ITSELF produces different bf16 roundings (another accumulation order in
`addmm`), so there the comparison falls back to the written cross-host tolerances below -- measured as the
reference-vs-reference spread between two such hosts (KV <= 4e-3, t2i latents <= 1e-2, 3-forward edit latents
<= 2.8e-2).  The host-independent bit-exact pin (oracle == live reference on THIS host) is
tests/test_reference_crosscheck.py::test_oracle_bit_exact_vs_live_reference."""
import warnings

import pytest
import torch

from oracle import bagel_oracle as O
from oracle import packers as P
from oracle.configs import TINY, TINY_D128, TINY_DENSE, TINY_MOE, TINY_ROPE, NEW_TOKEN_IDS_TINY, StubTokenizer
from tests.util_models import oracle_weights

CFGS = {"tiny": TINY, "tiny_d128": TINY_D128, "tiny_rope": TINY_ROPE, "tiny_dense": TINY_DENSE, "tiny_moe": TINY_MOE}


TOL_KV = 1e-2          # rel-L2 of a bf16 KV cache / feature tensor after L layers
TOL_LATENT = 2e-2      # rel-L2 of the fp32 latents after the Euler loop (2 forwards per step)
TOL_LATENT3 = 4e-2     # same with 3 forwards per step (edit) or the TaylorSeer extrapolation
TOL_LOSS = 1e-2        # per-token training losses
_inexact = []


def same(a, b, tol, what=""):
    """Bit-exact, or -- floats on a host with another CPU bf16 matmul backend -- within ``tol`` (rel-L2)."""
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
    if torch.equal(a, b):
        return
    assert a.is_floating_point(), f"{what}: integer tensors must be bit-exact"
    rel = ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()
    assert rel <= tol, f"{what}: rel-L2 {rel:.3e} > {tol:.1e}"
    if not _inexact:
        warnings.warn("oracle != golden bit-for-bit on this host (different CPU bf16 matmul backend than the fixture "
                      "host); compared within the cross-host tolerances instead")
    _inexact.append((what, rel))


def same_tokens(toks, logits, g_toks, g_logits, what=""):
    """Greedy ids equal the fixture's up to the first NEAR TIE in the fixture's logits (same rule as test_model_gpu.py)."""
    assert toks.shape == g_toks.shape and toks.dtype == g_toks.dtype
    for s in range(1, g_toks.shape[0]):
        if torch.equal(toks[s], g_toks[s]):
            same(logits[s - 1], g_logits[s - 1], 2e-2, f"{what} logits step {s}")
            continue
        row = g_logits[s - 1].float()
        gap = row.max(-1).values - row.gather(-1, toks[s].view(-1, 1)).squeeze(-1)
        assert (gap <= row.abs().max().item() * 2.0 ** -6).all(), f"{what}: tokens differ at step {s} without a near tie"
        break


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
        same(cache.key_cache[i], g["key_cache"][i], TOL_KV, f"K cache layer {i}")
        same(cache.value_cache[i], g["value_cache"][i], TOL_KV, f"V cache layer {i}")
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(O.OracleCache(L), g["cfg_inputs"]),
                           **g["gen_kwargs"])
    for a, b in zip(lat, g["latents"]):
        same(a, b, TOL_LATENT, "latents")
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(O.OracleCache(L), g["cfg_inputs"]),
                           **g["gen_kwargs_channel"])
    for a, b in zip(lat, g["latents_channel"]):
        same(a, b, TOL_LATENT, "latents (channel renorm)")


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
        same(cache.key_cache[i], g["key_cache_img"][i], TOL_KV, f"K cache (vae+vit) layer {i}")
    cfg_text_cache = cache.clone()
    l1, l2, l3, l4 = g["lens"]
    r1, r2, r3, r4 = g["ropes"]
    pi = P.prepare_prompts(l2, r2, [g["prompt"]], tok, NEW_TOKEN_IDS_TINY)[0]
    cache = O.forward_cache_update_text(W, cfg, cache, **pi)
    for i in range(L):
        same(cache.key_cache[i], g["key_cache"][i], TOL_KV, f"K cache (vae+vit+text) layer {i}")
        same(cache.value_cache[i], g["value_cache"][i], TOL_KV, f"V cache layer {i}")
    pi2 = P.prepare_prompts([0], [0], [g["prompt"]], tok, NEW_TOKEN_IDS_TINY)[0]
    cimg = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **pi2)
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(cfg_text_cache, g["cfg_text_inputs"]),
                           cfg_img=_cfgd(cimg, g["cfg_img_inputs"]), **g["gen_kwargs"])
    same(lat[0], g["latents"][0], TOL_LATENT3, "edit latents")
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(cfg_text_cache, g["cfg_text_inputs"]),
                           cfg_img=_cfgd(cimg, g["cfg_img_inputs"]), **g["gen_kwargs_global"])
    same(lat[0], g["latents_global"][0], TOL_LATENT3, "edit latents (global renorm)")
    si = g["start_inputs"]
    toks, logits = O.generate_text(W, cfg, cache.clone(), si["packed_key_value_indexes"], si["key_values_lens"],
                                   si["packed_start_tokens"], si["packed_query_position_ids"], 8, return_logits=True)
    same_tokens(toks, logits, g["tokens"], g["logits"], "generate_text")


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
    same(out, g["out"], TOL_KV, "siglip features")
    same(O.connector(W, out), g["connector_out"], TOL_KV, "connector")


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
                same(a, b, TOL_LATENT3, f"taylorseer {tag} last_only={last_only}")


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


@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_dense", "tiny_moe"])
def test_training_forward_matches_reference(golden, name):
    """Bagel.forward (bagel.py:101-229) with nested masks: per-token MSE and CE losses bit-exact; the mask rule restated.  tiny_dense / tiny_moe:
    the dense and MoE layer kinds' forward_train (qwen2_navit.py:620-646,852-883)."""
    cfg = CFGS[name]
    g = golden(f"{name}_train")
    W, _ = oracle_weights(cfg)
    out = O.bagel_forward_train(W, cfg, g["batch"], g["noise"], timestep_shift=cfg["bagel"]["timestep_shift"])
    same(out["mse"], g["mse"], TOL_LOSS, "mse")
    same(out["ce"], g["ce"], TOL_LOSS, "ce")
    # the masks in the fixture came from the reference's prepare_attention_mask_per_sample
    i = 0
    for n, m in zip(g["batch"]["sample_lens"], g["batch"]["nested_attention_masks"]):
        lens, modes, tot = [], [], 0
        while tot < n:
            lens.append(g["split_lens"][i]); modes.append(g["attn_modes"][i]); tot += lens[-1]; i += 1
        assert torch.equal(O.attention_mask_per_sample(lens, modes), m)


def test_training_backward_matches_reference(golden):
    """``loss.backward()`` of the unmodified reference (train/pretrain_unified_navit.py:683-735) vs torch autograd over the oracle's
    restated forward: every parameter gradient of the tiny model (111 tensors, ViT included) -- bit for bit on the fixture host, within
    the cross-host bf16 spread elsewhere (gradients pass through ~4 bf16 roundings per layer, twice: 3e-2)."""
    cfg = CFGS["tiny"]
    g, fx = golden("tiny_train"), golden("tiny_train_grads")
    W, _ = oracle_weights(cfg)
    loss, grads, _ = O.training_step_grads(W, cfg, g["batch"], g["noise"], fx["ce_loss_weights"], names=set(fx["grads"]))
    assert abs(loss - fx["loss"]) <= 1e-2 * abs(fx["loss"])
    assert set(grads) == set(fx["grads"]) and len(grads) == 111
    for k, ref in fx["grads"].items():
        if float(ref.float().norm()) == 0.0:
            assert float(grads[k].float().norm()) == 0.0, k
        else:
            same(grads[k], ref, 3e-2, f"grad {k}")
    assert not O.GRAD_ENABLED, "the gradient oracle must leave the forward oracle graph-free"


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
        same(cache.key_cache[i], g["key_cache"][i], TOL_KV, f"K cache layer {i}")
        same(cache.value_cache[i], g["value_cache"][i], TOL_KV, f"V cache layer {i}")
    lat = O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=_cfgd(O.OracleCache(L), g["cfg_inputs"]), **g["gen_kwargs"])
    for a, b in zip(lat, g["latents"]):
        same(a, b, TOL_LATENT, "latents")


def test_fp32_master_reference_run_is_a_stated_distance_away(golden):
    """The reference's two deployed precisions -- bf16 weights (app.py:105-113, every other fixture) and fp32 master weights under autocast
    (eval/gen/gen_images_mp.py:165-176) -- give different latents on identical inputs; the fixture written by
    oracle/make_golden_fp32master.py states by how much (~1e-2 rel-L2), and the numbers are re-derived here from the stored tensors."""
    import torch
    for name in ("tiny", "tiny_d128"):
        g, gm = golden(f"{name}_t2i"), golden(f"{name}_t2i_fp32master")
        for key in ("latents", "latents_channel"):
            d = max(float((a - b).norm() / b.norm()) for a, b in zip(gm[key], g[key]))
            assert abs(d - gm["deviation_of_bf16_weights_reference"][key]) < 1e-6
            assert 5e-3 < d < 2e-2, (name, key, d)


def test_fp8_restatement_known_answers():
    """oracle/fp8.py (the CPU statement of the product's gen_weight_quant='fp8' option) on hand-checkable values: the row scale is
    absmax / 448, codes are OCP e4m3fn with round-to-nearest-even, and the GEMM restatement equals the fp32 product of the de-quantised
    operands."""
    import torch
    from oracle import fp8 as F8
    x = torch.tensor([[448.0, 224.0, 1.0, -0.4375, 0.0, 3.25, 17.0, 18.0]], dtype=torch.float32)
    q, s = F8.quantize_rows_fp8(x)
    assert float(s[0]) == 1.0
    back = F8.dequant(q, s)[0].tolist()
    # e4m3: 3 mantissa bits -> 17 lies between 16 and 18 (tie -> even mantissa = 16), 3.25 is exact, 0.4375 = 7 * 2^-4 is exact
    assert back == [448.0, 224.0, 1.0, -0.4375, 0.0, 3.25, 16.0, 18.0]
    y = torch.tensor([[100.0, -50.0, 25.0, 0.0, 0.0, 0.0, 0.0, 0.0]])
    q2, s2 = F8.quantize_rows_fp8(y)
    assert abs(float(s2[0]) - 100.0 / 448.0) < 1e-7          # fp32 division
    assert torch.allclose(F8.dequant(q2, s2)[0, :3], torch.tensor([100.0, -50.0, 25.0]), rtol=2 ** -4)
    g = torch.Generator().manual_seed(0)
    a, w = torch.randn(5, 64, generator=g), torch.randn(7, 64, generator=g)
    qa, sa = F8.quantize_rows_fp8(a)
    qw, sw = F8.quantize_rows_fp8(w)
    ref = (F8.dequant(qa, sa) @ F8.dequant(qw, sw).t()).to(torch.bfloat16)
    assert torch.equal(F8.gemm_fp8(qa, sa, qw, sw), ref) or (F8.gemm_fp8(qa, sa, qw, sw).float() - ref.float()).abs().max() <= 2 ** -7 * ref.float().abs().max()
    z = torch.zeros(2, 16)
    qz, sz = F8.quantize_rows_fp8(z)
    assert sz.tolist() == [1.0, 1.0] and int(qz.view(torch.uint8).max()) == 0


def test_mxfp4_restatement_known_answers():
    """oracle/mxfp4.py (the CPU statement of the product's decode weight_quant='mxfp4' option) against the OCP Microscaling v1.0
    definitions on hand-checkable values: the E2M1 grid and its round-to-nearest-even ties, the shared exponent max(exp(amax) - 2, 0),
    saturation at 6 X, nibble packing, the device scale order, and that de-quantisation inverts quantisation on representable blocks."""
    from oracle import mxfp4 as MX
    grid = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6, -.5, -6, .25, .75, 1.25, 1.75, 2.5, 3.5, 5, 7, -7] + [0] * 13)
    w = (grid * 0.125).to(torch.bfloat16)[None]                        # amax = 7/8 -> exponent 126 -> scale byte 124 -> X = 1/8
    q, sb = MX.quantize_mxfp4(w)
    assert sb.tolist() == [[124]]
    d = MX.dequant_mxfp4(q, sb)[0] / 0.125
    assert d[:19].tolist() == [0, .5, 1, 1.5, 2, 3, 4, 6, -.5, -6, 0, 1, 1, 2, 2, 4, 4, 6, -6]      # ties to even, saturation
    codes = [int(q[0, i // 2] >> (4 * (i % 2))) & 0xF for i in range(10)]
    assert codes == [0, 1, 2, 3, 4, 5, 6, 7, 9, 15]                    # element 2i in the low nibble of byte i, sign in bit 3
    z = torch.zeros(1, 32, dtype=torch.bfloat16)
    qz, sz = MX.quantize_mxfp4(z)
    assert sz.tolist() == [[0]] and int(qz.max()) == 0
    # a block whose values are all on the grid of its own scale survives exactly
    g = torch.Generator().manual_seed(0)
    vals = MX.E2M1[torch.randint(0, 8, (4, 96), generator=g)] * torch.where(torch.rand(4, 96, generator=g) < 0.5, -1.0, 1.0)
    vals[:, 0::32] = 6.0                                               # pin every block's amax so its scale is 2^0 * 2^k
    w2 = (vals * torch.tensor([1.0, 0.25, 8.0, 2.0 ** -10])[:, None]).to(torch.bfloat16)
    q2, s2 = MX.quantize_mxfp4(w2)
    assert torch.equal(MX.dequant_mxfp4(q2, s2), w2.float())
    # device order of the scale bytes: byte 4q + j of a 16-byte group = block q of k-step j
    nat = torch.arange(2 * 5 * 4, dtype=torch.uint8).view(2, 20)      # 5 k-steps -> 2 groups, 3 padded steps
    dev = MX.permute_scales(nat)
    assert dev.shape == (2, 32)
    assert dev[0, :16].tolist() == [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]
    assert dev[0, 16:].tolist() == [16, 127, 127, 127, 17, 127, 127, 127, 18, 127, 127, 127, 19, 127, 127, 127]
    # the projection restatement reduces to the plain product when nothing is lost: grid weights, activations that are exact in e4m3
    x = torch.tensor([[1.0, -2.0, 0.5, 448.0] + [0.0] * 92]).to(torch.bfloat16)
    y = MX.gemv_w4(x, q2, s2)
    assert torch.equal(y, (x.float() @ w2.float().t()).to(torch.bfloat16))


def test_nf4_restatement_known_answers():
    """oracle/nf4.py (the CPU statement of bitsandbytes' NF4, the reference's 4-bit load mode, app.py:114-125) against what the library
    publishes: the 16-entry code book (QLoRA appendix E: quantiles of N(0,1), exact zero, symmetric ends), the kernel's thresholds =
    midpoints of adjacent entries with ties going DOWN, blocks of 64 with an fp32 absmax, the EVEN element in the HIGH nibble."""
    from oracle import nf4
    c = nf4.NF4_CODE
    assert c[0] == -1 and c[7] == 0 and c[15] == 1 and (c[1:] > c[:-1]).all() and len(c) == 16
    assert torch.allclose(nf4.NF4_THRESH.double(), ((c[1:] + c[:-1]) / 2).double(), atol=5e-8)
    # every code-book value (times an absmax) quantises to itself; packing order
    w = (c.repeat(4) * 3.0).view(1, 64)
    p, a = nf4.quantize_nf4(w)
    assert a.tolist() == [[3.0]] and nf4.codes_of(p)[0, :16].tolist() == list(range(16))
    assert p[0, :8].tolist() == [0x01, 0x23, 0x45, 0x67, 0x89, 0xAB, 0xCD, 0xEF], "even element in the high nibble"
    assert torch.equal(nf4.dequantize_nf4(p, a, torch.float32), w)
    # a value exactly on a threshold goes to the lower code; just above it to the upper
    t = nf4.NF4_THRESH[9].item()
    w2 = torch.zeros(1, 64); w2[0, 0] = 1.0; w2[0, 1] = t; w2[0, 2] = float(torch.nextafter(torch.tensor(t), torch.tensor(1.0)))
    assert nf4.codes_of(nf4.quantize_nf4(w2)[0])[0, :3].tolist() == [15, 9, 10]
    # an all-zero block: 1 / absmax = inf, x = NaN, every comparison false -> code 0, de-quantised to (-1) * 0
    p0, a0 = nf4.quantize_nf4(torch.zeros(2, 128))
    assert (p0 == 0).all() and (a0 == 0).all() and (nf4.dequantize_nf4(p0, a0).float() == 0).all()
    # blocks are 64 consecutive elements of a row; a bf16 weight matrix costs ~9 % relative error
    g = torch.Generator().manual_seed(3)
    W = (torch.randn(8, 256, generator=g) * 0.05).to(torch.bfloat16)
    p3, a3 = nf4.quantize_nf4(W)
    assert p3.shape == (8, 128) and a3.shape == (8, 4) and torch.equal(a3, W.float().view(8, 4, 64).abs().amax(2))
    e = ((nf4.dequantize_nf4(p3, a3).float() - W.float()).norm() / W.float().norm()).item()
    assert 0.05 < e < 0.13, e


TOL_VAE_BF16 = 3e-2    # rel-L2 of a bf16-autocast VAE output between two hosts' oneDNN bf16 conv paths (measured 1.0e-2 between the fixture host and an avx512 host)


def test_bf16_autocast_vae_restatement_is_pinned(golden):
    """tests/golden/vae_full_bf16.pt (oracle/make_golden_vae_bf16.py): the UNMODIFIED reference VAE under torch.autocast("cpu", bfloat16) --
    the region inferencer.py:233 opens -- and the oracle's two cast-point policies.  On the fixture's host kind "cpu" reproduces the
    reference's stored outputs bit for bit and "cuda" -- the policy of the device the reference runs on, group_norm in fp32 -- its own; on a
    host with another oneDNN bf16 convolution path the REFERENCE ITSELF moves (checked: live reference == oracle "cpu" bit for bit there,
    both 1.0e-2 from the fixture), so the comparison follows the suite's cross-host rule (`same`).  The host-independent bit-exact pin is
    tests/test_reference_crosscheck.py::test_bf16_autocast_vae_bit_exact_vs_live_reference."""
    from oracle import bagel_oracle as O
    from oracle.configs import VAE_FULL
    from oracle.shapes import vae_shapes
    from oracle.weights import synth_state_dict
    g = golden("vae_full_bf16")
    VW = synth_state_dict(vae_shapes(VAE_FULL["vae"]), 0)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    try:
        O.VAE_AUTOCAST = "cpu"
        dec_cpu = O.vae_decode(VW, VAE_FULL["vae"], g["z"])
        enc_cpu = O.vae_encode(VW, VAE_FULL["vae"], g["x"], g["enc_noise"].to(torch.bfloat16))
        O.VAE_AUTOCAST = "cuda"
        dec_cuda = O.vae_decode(VW, VAE_FULL["vae"], g["z"])
        dec_cuda2 = O.vae_decode(VW, VAE_FULL["vae"], g["z"])
        enc_cuda = O.vae_encode(VW, VAE_FULL["vae"], g["x"], g["enc_noise"].to(torch.bfloat16))
    finally:
        O.VAE_AUTOCAST = None
    assert dec_cpu.dtype == torch.bfloat16
    same(dec_cpu, g["decoded_cpu"], TOL_VAE_BF16, "oracle ('cpu' policy) vs the reference under cpu autocast: decode")
    same(enc_cpu, g["encoded_cpu"], TOL_VAE_BF16, "oracle ('cpu' policy) vs the reference under cpu autocast: encode")
    same(dec_cuda, g["decoded_cuda"], TOL_VAE_BF16, "oracle ('cuda' policy) vs its stored output: decode")
    same(enc_cuda, g["encoded_cuda"], TOL_VAE_BF16, "oracle ('cuda' policy) vs its stored output: encode")
    assert torch.equal(dec_cuda, dec_cuda2)                                           # the checker is deterministic on one host
    d = rel(dec_cuda, dec_cpu)                                                        # the GroupNorm rounding point: the same order of magnitude on every host
    assert 1e-3 < g["distance"]["decode_cuda_vs_cpu"] < 3e-2 and 1e-3 < d < 3e-2 and abs(d - g["distance"]["decode_cuda_vs_cpu"]) < 1e-2
    assert O.vae_decode(VW, VAE_FULL["vae"], g["z"]).dtype == torch.float32          # default: the fp32 VAE, untouched
