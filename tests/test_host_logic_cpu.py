"""CPU: the HOST logic of the product above the C ABI -- ForwardPlan, NaiveCache, the MoT row lists, the prefill, the
denoise loop with CFG, the opt-in stream-batched CFG forward and the marker-row side path -- run with the launch wrappers of
``bagel_amd.ops`` replaced by the torch stand-ins of tests/mock_ops.py (test infrastructure; the kernels themselves are
checked on the GPU by the ``-m gpu`` tests) and compared against the REFERENCE's golden vectors.

What this pins without a GPU: every index tensor, buffer layout and call order the engines hand to the C ABI."""
import copy

import pytest
import torch

from oracle.configs import TINY, TINY_D128, TINY_DENSE, TINY_MOE, NEW_TOKEN_IDS_TINY, StubTokenizer
from tests import mock_ops
from tests.util_models import oracle_weights

CFGS = {"tiny": TINY, "tiny_d128": TINY_D128, "tiny_dense": TINY_DENSE, "tiny_moe": TINY_MOE}


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()


def cpu_model(cfg):
    from bagel_amd.factory import build_bagel
    W, _ = oracle_weights(cfg)
    model, _ = build_bagel(cfg, device="cpu", with_vae=False)
    model.load_state_dict(W, strict=True)
    return model.to(torch.bfloat16).eval()


def new_cache(cfg):
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    return NaiveCache(cfg["llm"]["num_hidden_layers"])


def cfg_kwargs(tag, cache, d):
    return {f"{tag}_past_key_values": cache, f"{tag}_packed_position_ids": d["cfg_packed_position_ids"],
            f"{tag}_packed_query_indexes": d["cfg_packed_query_indexes"], f"{tag}_key_values_lens": d["cfg_key_values_lens"],
            f"{tag}_packed_key_value_indexes": d["cfg_packed_key_value_indexes"]}


@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_dense", "tiny_moe"])
def test_text_to_image_host_path_matches_reference(golden, monkeypatch, name):
    """prepare_prompts -> forward_cache_update_text -> generate_image (cond + CFG-text, global / channel renorm) through the
    product's engines with stand-in operators vs the reference's KV cache and latents."""
    mock_ops.install(monkeypatch)
    cfg = CFGS[name]
    g = golden(f"{name}_t2i")
    model = cpu_model(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, newlens, newrope = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    for i in range(L):
        assert rel(cache.key_cache[i], g["key_cache"][i]) < 1e-2, f"K cache layer {i}"
        assert rel(cache.value_cache[i], g["value_cache"][i]) < 1e-2, f"V cache layer {i}"
    runs = [("gen_kwargs", "latents")] + ([("gen_kwargs_channel", "latents_channel")] if "gen_kwargs_channel" in g else [])
    for kw, key in runs:
        lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **g[kw],
                                   **g["latent_inputs"])
        assert len(lat) == len(g[key])
        for a, b in zip(lat, g[key]):
            assert a.shape == b.shape and a.dtype == torch.float32
            assert rel(a, b) < 2e-2, key


def _roundtrip_linears(model, kind):
    """Every decoder projection W <- dequantise(quantise(W)) in place: what a bf16 engine has to compute on to equal a quantised engine."""
    from oracle import nf4
    for L in model.language_model.model.layers:
        mods = [getattr(L.self_attn, n + s) for n in ("q_proj", "k_proj", "v_proj", "o_proj") for s in ("", "_moe_gen")]
        for s in ("", "_moe_gen"):
            m = getattr(L, "mlp" + s)
            mods += [m.gate_proj, m.up_proj, m.down_proj]
        for m in mods:
            w = m.weight.data
            if kind == "nf4":
                m.weight.data = nf4.dequantize_nf4(*nf4.quantize_nf4(w)).to(w.dtype)
            else:
                m.weight.data = mock_ops.dequantize_rows_i8(*mock_ops.quantize_rows_i8(w)).to(w.dtype)
    model.language_model.invalidate_packed()


@pytest.mark.parametrize("kind", ["nf4", "int8_rowwise"])
def test_whole_model_quantised_load_mode_equals_the_bf16_engine_on_dequantised_weights(golden, monkeypatch, kind):
    """Bagel.quantize_language_model (app.py:114-131's load modes for the WHOLE forward path): prefill, denoise loop and text decode of a model whose
    decoder projections are stored as NF4 / INT8 codes give exactly what the bf16 engine gives on the de-quantised weights -- the layer-by-layer
    materialisation is bitsandbytes' dequantise-then-F.linear, nothing else changes -- and the bf16 projection weights are really gone."""
    mock_ops.install(monkeypatch)
    monkeypatch.setenv("BAGEL_DECODE_GRAPH", "0")
    cfg = CFGS["tiny_d128"]          # head_dim 128 like 7B: no head padding, so the 64-weight blocks of the packed matrices are the nn.Linear's own
    g = golden("tiny_d128_t2i")
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    qm, bm = cpu_model(cfg), cpu_model(cfg)
    qm.to(torch.bfloat16); bm.to(torch.bfloat16)
    _roundtrip_linears(bm, kind)
    resident = qm.quantize_language_model(kind)
    full = sum(p.numel() * 2 for n, p in bm.language_model.model.layers.named_parameters() if "proj" in n and n.endswith("weight"))
    assert resident < (0.30 if kind == "nf4" else 0.52) * full          # 4 bits + an fp32 absmax per 64 weights = 0.28; 8 bits + a scale per row = 0.50
    assert all(p.numel() == 0 for n, p in qm.language_model.model.layers.named_parameters() if "proj" in n and n.endswith("weight"))
    outs = []
    for model in (qm, bm):
        gi, newlens, newrope = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
        cache = model.forward_cache_update_text(new_cache(cfg), **gi)
        lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **g["gen_kwargs"], **g["latent_inputs"])
        prefill = [(cache.key_cache[i].clone(), cache.value_cache[i].clone()) for i in range(cfg["llm"]["num_hidden_layers"])]
        st = model.prepare_start_tokens(newlens, newrope, NEW_TOKEN_IDS_TINY)
        # the quantised engine decodes on its stored codes (the stand-in's 4- / 8-bit gemv = the bf16 gemv on the de-quantised weight, which is what
        # the bf16 engine holds here); on the GPU the codes are expanded in fp32 inside the kernel (tests/test_nf4_gpu.py: tokens up to near ties)
        toks = model.generate_text(past_key_values=cache, max_length=5, do_sample=False, end_token_id=None, **st)
        outs.append((prefill, lat, toks))
    (ca, la, ta), (cb, lb, tb) = outs
    for (ka, va), (kb, vb) in zip(ca, cb):
        assert torch.equal(ka, kb) and torch.equal(va, vb), "prefill KV of the quantised engine differs from the bf16 engine on de-quantised weights"
    for a, b in zip(la, lb):
        assert torch.equal(a, b), "latents of the quantised engine differ from the bf16 engine on de-quantised weights"
    assert torch.equal(ta, tb)
    assert qm._last_decode_session.weight_quant == kind          # the decode streamed the stored codes
    with pytest.raises(NotImplementedError):
        qm.language_model.engine().refresh()
    qm.language_model.invalidate_packed()            # the codes were the only copy: rebuilding from the released parameters has to say so
    with pytest.raises(RuntimeError, match="reload the checkpoint"):
        qm.language_model.engine()


def _three_stream_setup(model, cfg, sizes):
    """cond / cfg-text / cfg-img contexts of DIFFERENT lengths from three prompts (what the edit flow produces, without the
    ViT / VAE prefill): exercises ragged contexts and NaiveCache.concat with three live caches."""
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ids = NEW_TOKEN_IDS_TINY
    B = len(sizes)
    out = []
    for prompts in (["a small red cube on the table", "sky"][:B], ["cube", "a b"][:B], ["x y z w v", "hello world again"][:B]):
        gi, lens, ropes = model.prepare_prompts([0] * B, [0] * B, prompts, tok, ids)
        out.append((model.forward_cache_update_text(new_cache(cfg), **gi), lens, ropes))
    (c0, l0, r0), (c1, l1, r1), (c2, l2, r2) = out
    torch.manual_seed(7)
    li = model.prepare_vae_latent(l0, r0, sizes, ids)
    ct = model.prepare_vae_latent_cfg(l1, r1, sizes)
    ci = model.prepare_vae_latent_cfg(l2, r2, sizes)
    return c0, li, (c1, ct), (c2, ci)


@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_dense", "tiny_moe"])
@pytest.mark.parametrize("side", [False, True], ids=["rows_in_tile_gemm", "marker_side_path"])
@pytest.mark.parametrize("streams", [2, 3])
def test_stream_batched_cfg_equals_sequential(monkeypatch, name, side, streams):
    """model.cfg_batched: the cond + CFG forwards of a step as ONE forward over [stream 0 | stream 1 | ...].  With operators
    whose row results do not depend on the batch (mock_ops) the latents must equal the sequential path BIT FOR BIT -- every
    routing list, context offset, RoPE row and V^T column of the concatenated plan / cache is in play -- with and without the
    dense side path for the marker rows, for 2 streams (text->image, second context empty) and 3 (edit-style ragged contexts),
    and with CFG switched off for part of the schedule (cfg_interval) so both code paths alternate inside one run."""
    mock_ops.install(monkeypatch)
    cfg = CFGS[name]
    model = cpu_model(cfg)
    sizes = [(64, 64), (32, 64)]
    c0, li, (c1, ct), (c2, ci) = _three_stream_setup(model, cfg, sizes)
    if streams == 2:
        c1, ct = new_cache(cfg), model.prepare_vae_latent_cfg([0, 0], [0, 0], sizes)     # text->image: no CFG context at all
    kw = dict(num_timesteps=5, timestep_shift=3.0, cfg_text_scale=4.0, cfg_interval=[0.6, 1.0], cfg_renorm_min=0.0,
              cfg_renorm_type="global" if streams == 2 else "text_channel", **cfg_kwargs("cfg_text", c1, ct))
    if streams == 3:
        kw.update(cfg_img_scale=2.0, **cfg_kwargs("cfg_img", c2, ci))
    model.cfg_batched = False
    ref = model.generate_image(past_key_values=copy.deepcopy(c0), **kw, **li)
    model.cfg_batched, model.und_side_path = True, side
    calls = []
    eng = model.language_model.engine()
    orig = eng.forward
    monkeypatch.setattr(eng, "forward", lambda seq, plan, *a, **k: (calls.append((plan.B, plan.M, plan.und_side)), orig(seq, plan, *a, **k))[1])
    got = model.generate_image(past_key_values=copy.deepcopy(c0), **kw, **li)
    B, M = len(sizes), sum(int(x) for x in li["packed_seqlens"])
    assert (streams * B, streams * M, side) in calls, "the batched forward never ran"     # (the flag; the side path itself is MoT-only)
    assert (B, M, False) in calls, "cfg_interval should have left single-stream steps in the schedule"
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_concat_plan_and_cache_layout(monkeypatch):
    """Known-answer check of concat_plans / NaiveCache.concat on a hand-sized case."""
    mock_ops.install(monkeypatch)
    from bagel_amd.modeling.bagel.qwen2_navit import ForwardPlan, NaiveCache, concat_plans
    dev = torch.device("cpu")
    inv = torch.tensor([1.0, 0.5])
    mk = lambda pos, ctx: ForwardPlan(dev, [3, 4], pos, key_values_lens=ctx, text_indexes=[0, 2, 3, 6], vae_indexes=[1, 4, 5],  # noqa: E731
                                      inv_freq=inv)
    a = mk(torch.tensor([5, 5, 5, 2, 2, 2, 2]), [5, 2])
    b = mk(torch.tensor([0, 0, 0, 0, 0, 0, 0]), [0, 0])
    p = concat_plans([a, b])
    assert (p.B, p.M, p.q_lens, p.ctx_lens, p.max_lq) == (4, 14, [3, 4, 3, 4], [5, 2, 0, 0], 4)
    assert p.cu_q.tolist() == [0, 3, 7, 10, 14] and p.vt_new_col.tolist() == [0, 64, 128, 192] and p.vt_cols == 256
    assert p.text_idx.tolist() == [0, 2, 3, 6, 7, 9, 10, 13] and p.vae_idx.tolist() == [1, 4, 5, 8, 11, 12]
    assert p.expert.tolist() == [0, 1, 0, 0, 1, 1, 0] * 2 and (p.n_text, p.n_vae) == (8, 6) and p.has_ctx
    assert torch.equal(p.cos, torch.cat([a.cos, b.cos])) and torch.equal(p.pos_ids, torch.cat([a.pos_ids, b.pos_ids]))
    c = NaiveCache(2)
    k = torch.arange(7 * 8, dtype=torch.float32).view(7, 8).to(torch.bfloat16)
    for i in range(2):
        c.store(i, k + i, -k - i, [5, 2], [0, 0], 1, 8, 8)
    m = NaiveCache.concat([c, NaiveCache(2), None], [2, 2, 2])
    assert m.lens(0) == [5, 2, 0, 0, 0, 0] and m.seq_lens == 7 and m.num_layers == 2
    assert torch.equal(m.key_cache[1], (k + 1).view(7, 1, 8)) and torch.equal(m.value_cache[0], (-k).view(7, 1, 8))
    cu, col, cols, mx = m._meta(0, dev)
    assert cu.tolist() == [0, 5, 7, 7, 7, 7, 7] and col.tolist() == [0, 64, 128, 192, 256, 320] and mx == 5
    assert NaiveCache.concat([NaiveCache(2), None], [2, 2]).is_empty(0)
    with pytest.raises(ValueError):
        NaiveCache.concat([c, c], [2, 3])


@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_rope"])
def test_siglip_host_path_matches_reference(golden, monkeypatch, name):
    """SiglipVisionModel (packed patches, learned position table or 2-D RoPE, head_dim 32 / 72 padded to the 64 / 128 slots) and
    the connector: weight packing with zero-padded heads, fused qkv layout, V^T offsets."""
    from oracle.configs import TINY_ROPE
    mock_ops.install(monkeypatch)
    cfg = dict(CFGS, tiny_rope=TINY_ROPE)[name]
    g = golden(f"{name}_siglip")
    model = cpu_model(cfg)
    out = model.vit_model(packed_pixel_values=g["tokens"], packed_flattened_position_ids=g["pos"], cu_seqlens=g["cu"], max_seqlen=35)
    assert out.shape == g["out"].shape and rel(out, g["out"]) < 1e-2


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_taylorseer_host_path_matches_reference(golden, monkeypatch, name):
    """generate_image(enable_taylorseer=True): the schedule (full / Taylor steps per stream), the last-layer-only feature cache and
    its hand-over to the final norm vs the reference's latents (tolerance as the GPU test: 4e-2)."""
    mock_ops.install(monkeypatch)
    cfg = CFGS[name]
    g = golden(f"{name}_taylorseer")
    model = cpu_model(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    for tag, run in g["runs"].items():
        kw = run["gen_kwargs"]
        lat = model.generate_image(past_key_values=cache, enable_taylorseer=True, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]),
                                   **kw, **g["latent_inputs"])
        st = model._last_taylor_states[0]
        assert st.full_steps + st.taylor_steps == kw["num_timesteps"] - 1 and st.taylor_steps > 0
        for a, b in zip(lat, run["latents"]):
            assert rel(a, b) < 4e-2, tag


@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_dense", "tiny_moe"])
@pytest.mark.parametrize("mask_api", ["nested", "splits"])
def test_training_forward_host_path_matches_reference(golden, monkeypatch, name, mask_api):
    """Bagel.forward: TrainPlan (block mask decomposed into per-split sequences with overlapping clean-key prefixes), und / gen
    routing, per-image timestep embedding rows, loss row selection -- vs the reference's per-token losses.  tiny_dense / tiny_moe: the
    training forward of Qwen2DecoderLayer (no routing) and Qwen2MoEDecoderLayer (shared attention, per-modality MLP + final norm)."""
    mock_ops.install(monkeypatch)
    cfg = CFGS[name]
    g = golden(f"{name}_train")
    model = cpu_model(cfg)
    batch = dict(g["batch"])
    if mask_api == "splits":
        batch.pop("nested_attention_masks")
        batch.update(split_lens=g["split_lens"], attn_modes=g["attn_modes"])
    out = model(noise=g["noise"], **batch)
    assert out["mse"].shape == g["mse"].shape and out["ce"].shape == g["ce"].shape
    assert rel(out["ce"], g["ce"]) < 2e-2 and rel(out["mse"], g["mse"]) < 5e-2


def _tokens_match(ours, ref, ref_logits, what):
    """Greedy ids equal the oracle's up to the first near tie in the oracle's logits (the rule of test_model_gpu.py)."""
    assert ours.shape == ref.shape and ours.dtype == torch.int64
    for s in range(1, ref.shape[0]):
        if torch.equal(ours[s], ref[s]):
            continue
        row = ref_logits[s - 1].float()
        gap = row.max(-1).values - row.gather(-1, ours[s].view(-1, 1)).squeeze(-1)
        assert (gap <= row.abs().max().item() * 2.0 ** -6).all(), f"{what}: tokens differ at step {s} without a near tie"
        break


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
@pytest.mark.parametrize("batch", [1, 2])
def test_understanding_flow_host_path_matches_oracle(monkeypatch, name, batch):
    """BASELINE configs[1] end to end on the host logic: SigLIP + connector prefill (non-causal, und) -> prompt prefill (causal)
    -> prepare_start_tokens -> generate_text through the DecodeSession (paged KV cache adopted from the NaiveCache, block
    tables, device-side step bookkeeping, write-back of the decoded K/V rows) -- eager launches (no hipGraph here) -- vs the
    oracle's restatement of the reference flow (bagel.py:299-415,232-297,909-1000): KV caches before and after the decode,
    greedy token ids.  batch 1 = the fused-RMSNorm gemv route, batch 2 = the RMSNorm + skinny GEMM route."""
    from oracle import bagel_oracle as O
    mock_ops.install(monkeypatch)
    cfg = CFGS[name]
    model = cpu_model(cfg)
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ids = NEW_TOKEN_IDS_TINY
    ident = lambda t: t  # noqa: E731
    g = torch.Generator().manual_seed(5)
    images = [torch.rand(3, 28, 42, generator=g) * 2 - 1, torch.rand(3, 42, 14, generator=g) * 2 - 1][:batch]
    prompts = ["what is in the picture", "a b c"][:batch]
    z = [0] * batch
    vi, l1, r1 = model.prepare_vit_images(z, z, images, ident, ids)
    cache = model.forward_cache_update_vit(new_cache(cfg), **vi)
    pi, l2, r2 = model.prepare_prompts(l1, r1, prompts, tok, ids)
    cache = model.forward_cache_update_text(cache, **pi)
    si = model.prepare_start_tokens(l2, r2, ids)
    # the oracle on the same inputs
    oc = O.forward_cache_update_vit(W, cfg, O.OracleCache(L), **vi)
    oc = O.forward_cache_update_text(W, cfg, oc, **pi)
    for i in range(L):
        assert rel(cache.key_cache[i], oc.key_cache[i]) < 1.5e-2 and rel(cache.value_cache[i], oc.value_cache[i]) < 1.5e-2
    n = 6
    otoks, ologits = O.generate_text(W, cfg, oc, si["packed_key_value_indexes"], si["key_values_lens"], si["packed_start_tokens"],
                                     si["packed_query_position_ids"], n, return_logits=True)
    toks = model.generate_text(past_key_values=cache, max_length=n, do_sample=False, end_token_id=None, use_graph=False, **si)
    _tokens_match(toks, otoks, ologits, f"{name} B={batch}")
    assert cache.seq_lens == sum(l2) + n * batch and cache.lens(0) == [x + n for x in l2]
    if torch.equal(toks, otoks):      # same prefix -> the decoded K/V rows written back to the caller's cache are comparable too
        for i in range(L):
            assert rel(cache.key_cache[i], oc.key_cache[i]) < 1.5e-2 and rel(cache.value_cache[i], oc.value_cache[i]) < 1.5e-2


def test_decode_mxfp4_weights_host_path_matches_oracle(monkeypatch):
    """generate_text(weight_quant='mxfp4') on the host logic: which weights are quantised (the und expert's fused qkv / o / interleaved
    gate+up / down; lm_head stays bf16), RMSNorm fused into the quantising projection, residual and SwiGLU epilogues -- vs the oracle's
    decode loop with the MXFP4 scheme switched into exactly those linears (oracle/mxfp4.py).  Prefill stays bf16 on both sides."""
    from oracle import bagel_oracle as O
    mock_ops.install(monkeypatch)
    cfg = CFGS["tiny_d128"]
    model = cpu_model(cfg)
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    pi, l2, r2 = model.prepare_prompts([0], [0], ["what is in the picture"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **pi)
    oc = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **pi)
    si = model.prepare_start_tokens(l2, r2, NEW_TOKEN_IDS_TINY)
    n = 5
    plain, _ = O.generate_text(W, cfg, copy.deepcopy(oc), si["packed_key_value_indexes"], si["key_values_lens"], si["packed_start_tokens"],
                               si["packed_query_position_ids"], 1, return_logits=True)
    O.MXFP4_WEIGHT_PTRS = O.mxfp4_decode_weight_ptrs(W)
    try:
        assert len(O.MXFP4_WEIGHT_PTRS) == 7 * L
        otoks, ologits = O.generate_text(W, cfg, oc, si["packed_key_value_indexes"], si["key_values_lens"], si["packed_start_tokens"],
                                         si["packed_query_position_ids"], n, return_logits=True)
    finally:
        O.MXFP4_WEIGHT_PTRS = set()
    toks = model.generate_text(past_key_values=cache, max_length=n, do_sample=False, end_token_id=None, use_graph=False, weight_quant="mxfp4", **si)
    assert model._last_decode_session.weight_quant == "mxfp4"
    _tokens_match(toks, otoks, ologits, "tiny_d128 mxfp4")
    first = model._last_decode_session
    assert first.w8 is not None and len(first.w8) == L


def test_decode_nf4_weights_host_path_matches_oracle(monkeypatch):
    """generate_text(weight_quant='nf4') -- the reference's own 4-bit load mode (app.py:114-125) -- on the host logic: the und expert's
    fused qkv / o / interleaved gate+up / down are quantised (row-wise blocks of 64: fusing and interleaving rows changes nothing),
    lm_head stays bf16 -- vs the oracle's decode loop with oracle/nf4.py switched into exactly those linears."""
    from oracle import bagel_oracle as O
    mock_ops.install(monkeypatch)
    cfg = CFGS["tiny_d128"]
    model = cpu_model(cfg)
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    pi, l2, r2 = model.prepare_prompts([0], [0], ["what is in the picture"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **pi)
    oc = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **pi)
    si = model.prepare_start_tokens(l2, r2, NEW_TOKEN_IDS_TINY)
    n = 5
    O.NF4_WEIGHT_PTRS = O.mxfp4_decode_weight_ptrs(W)          # the same seven linears per layer
    try:
        otoks, ologits = O.generate_text(W, cfg, oc, si["packed_key_value_indexes"], si["key_values_lens"], si["packed_start_tokens"],
                                         si["packed_query_position_ids"], n, return_logits=True)
    finally:
        O.NF4_WEIGHT_PTRS = set()
    with pytest.raises(NotImplementedError):
        model.generate_text(past_key_values=copy.deepcopy(cache), max_length=1, do_sample=False, end_token_id=None, use_graph=False, weight_quant="fp4", **si)
    toks = model.generate_text(past_key_values=cache, max_length=n, do_sample=False, end_token_id=None, use_graph=False, weight_quant="nf4", **si)
    sess = model._last_decode_session
    assert sess.weight_quant == "nf4" and sess.w8 is not None and len(sess.w8) == L
    _tokens_match(toks, otoks, ologits, "tiny_d128 nf4")


def cpu_model_and_vae(cfg):
    from bagel_amd.factory import build_bagel
    W, VW = oracle_weights(cfg)
    model, vae = build_bagel(cfg, device="cpu", with_vae=True)
    model.load_state_dict(W, strict=True)
    vae.load_state_dict(VW, strict=True)
    return model.to(torch.bfloat16).eval(), vae.eval()


def test_vae_mid_block_attention_is_blocked_over_query_rows(golden, monkeypatch):
    """VaeEngine.attn produces the score matrix in blocks of ATTN_ROWS query rows (1024^2: 16 384 tokens would be a 1.07 GB transient):
    ragged blocks give the same encode / decode as one block."""
    from bagel_amd.modeling import vae_engine
    mock_ops.install(monkeypatch)
    g = golden("tiny_vae")
    _, vae = cpu_model_and_vae(TINY)
    ref = vae.decode(g["z"]).clone()
    monkeypatch.setattr(vae_engine.VaeEngine, "ATTN_ROWS", 40)          # 16 x 24 = 384 latent tokens -> 10 blocks, the last one ragged
    assert torch.equal(vae.decode(g["z"]), ref)


def test_vae_host_path_matches_reference(golden, monkeypatch):
    """VaeEngine: weight packing (tap-major 3x3, channel padding), NHWC plumbing, conv modes (3x3, strided with the (0,1,0,1) pad,
    upsample-fused), ResnetBlock / AttnBlock residual folding -- encode / decode vs the reference, and the truncating uint8 image."""
    mock_ops.install(monkeypatch)
    g = golden("tiny_vae")
    _, vae = cpu_model_and_vae(TINY)
    dec = vae.decode(g["z"])
    assert dec.shape == g["decoded"].shape and (dec - g["decoded"]).abs().max() / g["decoded"].abs().max() < 1e-4
    enc = vae.encode(g["x"], sample_noise=g["enc_noise"])
    assert enc.shape == g["encoded"].shape and (enc - g["encoded"]).abs().max() / g["encoded"].abs().max() < 1e-4
    from bagel_amd.inferencer import InterleaveInferencer
    model = cpu_model(TINY)
    inf = InterleaveInferencer(model, vae, None, None, None, None)
    u8 = inf.image_to_u8(vae.decode(inf.latent_to_chw(g["packed_latent"], (8 * 16, 12 * 16))))
    diff = (u8.int() - g["image_u8"].int()).abs()
    assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 0.01


def test_bf16_autocast_vae_host_path_matches_the_oracle(golden, monkeypatch):
    """VaeEngineBf16 (the VAE inside the inferencer's bf16 autocast region): bf16 weight packing with channels padded to 8, bf16 NHWC
    plumbing, every conv mode, the residual folding and the bf16 attention (key axis padded to whole chunks) on the torch stand-ins --
    against the oracle with CUDA autocast's cast points (oracle VAE_AUTOCAST = "cuda": bf16 convs, fp32 GroupNorm, bf16 adds); precision
    selection by keyword; the uint8 image with eager-bf16 roundings."""
    mock_ops.install(monkeypatch)
    g = golden("tiny_vae")
    _, vae = cpu_model_and_vae(TINY)
    _, VW = oracle_weights(TINY)
    noise = g["enc_noise"].to(torch.bfloat16)
    from oracle import bagel_oracle as O
    O.VAE_AUTOCAST = "cuda"
    try:
        ref_dec, ref_enc = O.vae_decode(VW, TINY["vae"], g["z"]), O.vae_encode(VW, TINY["vae"], g["x"], noise)
    finally:
        O.VAE_AUTOCAST = None
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    dec = vae.decode(g["z"], precision="bf16")
    assert dec.dtype == torch.bfloat16 and dec.shape == ref_dec.shape and rel(dec, ref_dec) < 2e-2, rel(dec, ref_dec)       # (bf16 noise through ~30 convs: two summation orders sit ~1e-2 apart)
    enc = vae.encode(g["x"], sample_noise=noise.float(), precision="bf16")
    assert enc.dtype == torch.bfloat16 and enc.shape == ref_enc.shape and rel(enc, ref_enc) < 2e-2, rel(enc, ref_enc)
    assert vae.decode(g["z"], precision="fp32").dtype == torch.float32 and vae.decode(g["z"]).dtype == torch.float32
    with pytest.raises(ValueError):
        vae.decode(g["z"], precision="fp16")
    from bagel_amd.inferencer import InterleaveInferencer
    u8 = InterleaveInferencer.image_to_u8(dec)
    assert torch.equal(u8, ((dec * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255).to(torch.uint8))       # eager-bf16 roundings of inferencer.py:182-183


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_edit_flow_host_path_matches_reference(golden, monkeypatch, name):
    """BASELINE configs[4] on the host logic: VAE-encode + SigLIP prefill of a source image (two cache appends on a single-sample
    context), prompt on top, the three CFG contexts (deepcopy of the cache), the 3-forward sampler with text_channel / global
    renorm, and the greedy decode from the same context -- vs the reference's goldens (tolerances of the GPU test)."""
    mock_ops.install(monkeypatch)
    cfg = CFGS[name]
    g = golden(f"{name}_editund")
    model, vae = cpu_model_and_vae(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ids = NEW_TOKEN_IDS_TINY
    ident = lambda t: t  # noqa: E731

    class FixedNoiseVae:   # the reference draws randn_like inside encode; feed the recorded draw
        def encode(self, x):
            return vae.encode(x, sample_noise=g["enc_noise"])
    vi, l1, r1 = model.prepare_vae_images([0], [0], [g["img_vae"]], ident, ids)
    cache = model.forward_cache_update_vae(FixedNoiseVae(), new_cache(cfg), **vi)
    ti, l2, r2 = model.prepare_vit_images(l1, r1, [g["img_vit"]], ident, ids)
    cache = model.forward_cache_update_vit(cache, **ti)
    for i in range(L):
        assert rel(cache.key_cache[i], g["key_cache_img"][i]) < 1.5e-2
    cfg_text_cache = copy.deepcopy(cache)
    pi, l3, r3 = model.prepare_prompts(l2, r2, [g["prompt"]], tok, ids)
    cache = model.forward_cache_update_text(cache, **pi)
    for i in range(L):
        assert rel(cache.key_cache[i], g["key_cache"][i]) < 1.5e-2 and rel(cache.value_cache[i], g["value_cache"][i]) < 1.5e-2
    assert cfg_text_cache.seq_lens == g["key_cache_img"][0].shape[0], "deepcopy must not alias the appended cache"
    pi2, l4, r4 = model.prepare_prompts([0], [0], [g["prompt"]], tok, ids)
    cimg = model.forward_cache_update_text(new_cache(cfg), **pi2)
    for batched in (False, True):
        model.cfg_batched = batched
        for kw, key in ((g["gen_kwargs"], "latents"), (g["gen_kwargs_global"], "latents_global")):
            lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", cfg_text_cache, g["cfg_text_inputs"]),
                                       **cfg_kwargs("cfg_img", cimg, g["cfg_img_inputs"]), **kw, **g["latent_inputs"])
            assert rel(lat[0], g[key][0]) < 3e-2, (key, batched)
    toks = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=8, do_sample=False, end_token_id=None, use_graph=False,
                               **g["start_inputs"])
    _tokens_match(toks, g["tokens"], g["logits"], name)


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_interleave_inferencer_host_path_matches_reference(golden, monkeypatch, name):
    """bagel_amd/inferencer.py end to end (PIL / str in and out, the product's ImageTransform, the three contexts kept in step,
    think mode) against the REFERENCE inferencer's outputs, with the image tolerances of tests/test_inferencer_gpu.py."""
    import re
    import numpy as np
    from PIL import Image
    from bagel_amd.data.transforms import ImageTransform
    from bagel_amd.inferencer import InterleaveInferencer
    mock_ops.install(monkeypatch)
    monkeypatch.setenv("BAGEL_DECODE_GRAPH", "0")          # hipGraph capture needs a device; the eager launch sequence is the same
    cfg = CFGS[name]
    g = golden(f"{name}_inferencer")
    model, vae = cpu_model_and_vae(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    inf = InterleaveInferencer(model, vae, tok, ImageTransform(64, 32, 16, device="cpu"), ImageTransform(56, 28, 14, device="cpu"),
                               NEW_TOKEN_IDS_TINY)
    inf.vae_precision_in_autocast = "fp32"                 # the fixture's reference ran on CPU, where the CUDA autocast region leaves the VAE in fp32
    src = Image.fromarray(g["source_image"].numpy(), "RGB")

    def close(img, ref, mean_tol, p99_tol, what):
        d = np.abs(np.asarray(img).astype(np.int32) - ref.numpy().astype(np.int32))
        assert d.shape == tuple(ref.shape) and d.mean() <= mean_tol and np.percentile(d, 99) <= p99_tol, (what, d.mean(), np.percentile(d, 99))

    def ids_of(s):
        return re.findall(r"\[(\d+)\]", s)
    torch.manual_seed(g["t2i"]["seed"])
    r = inf(text=g["t2i"]["text"], **g["t2i"]["kwargs"])
    assert isinstance(r["image"], Image.Image) and r["text"] is None
    close(r["image"], g["t2i"]["image"], 1.5, 8, "text -> image")
    torch.manual_seed(g["edit"]["seed"])
    r = inf(image=src, text=g["edit"]["text"], **g["edit"]["kwargs"])
    close(r["image"], g["edit"]["image"], 3.0, 14, "image + text -> image")
    r = inf(image=src, text=g["understanding"]["text"], **g["understanding"]["kwargs"])
    ours, ref = ids_of(r["text"]), ids_of(g["understanding"]["answer"])
    first = next((i for i, (a, b) in enumerate(zip(ours, ref)) if a != b), len(ref))
    assert r["image"] is None and len(ours) == len(ref) and first >= 1 and ours[:first] == ref[:first]
    torch.manual_seed(g["think"]["seed"])
    r = inf(text=g["think"]["text"], **g["think"]["kwargs"])
    ours, ref = ids_of(r["text"]), ids_of(g["think"]["thought"])
    assert len(ours) == len(ref) and ours[0] == ref[0]
    if ours == ref:        # same planning text -> the image conditioned on it is comparable
        close(r["image"], g["think"]["image"], 1.5, 8, "think -> image")


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_interleaved_flows_match_oracle(monkeypatch, seed):
    """Differential fuzz of the host logic beyond the golden scenarios: a random batch (1-3 samples) goes through a random
    interleaving of prompt / ViT-image prefills (ragged lengths, multi-sample cache merges on top of existing contexts), then
    either a CFG text->image run on random latent sizes or a greedy decode -- product engines (stand-in operators) vs the oracle's
    restatement of the reference on identical packer outputs."""
    import random
    from oracle import bagel_oracle as O
    mock_ops.install(monkeypatch)
    rng = random.Random(seed)
    cfg = TINY if seed % 2 == 0 else TINY_D128
    model = cpu_model(cfg)
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ids = NEW_TOKEN_IDS_TINY
    ident = lambda t: t  # noqa: E731
    B = rng.randint(1, 3)
    words = ["a", "red", "cube", "on the table", "sky", "x y z", "hello world again", "what is it"]
    lens, ropes = [0] * B, [0] * B
    cache, oc = new_cache(cfg), O.OracleCache(L)
    g = torch.Generator().manual_seed(seed)
    for stage in range(rng.randint(1, 3)):
        if rng.random() < 0.4:
            imgs = [torch.rand(3, 14 * rng.randint(1, 3), 14 * rng.randint(1, 3), generator=g) * 2 - 1 for _ in range(B)]
            gi, lens, ropes = model.prepare_vit_images(lens, ropes, imgs, ident, ids)
            cache = model.forward_cache_update_vit(cache, **gi)
            oc = O.forward_cache_update_vit(W, cfg, oc, **gi)
        else:
            prompts = [" ".join(rng.choice(words) for _ in range(rng.randint(1, 3))) for _ in range(B)]
            gi, lens, ropes = model.prepare_prompts(lens, ropes, prompts, tok, ids)
            cache = model.forward_cache_update_text(cache, **gi)
            oc = O.forward_cache_update_text(W, cfg, oc, **gi)
        assert cache.lens(0) == lens and cache.seq_lens == sum(lens)
        for i in range(L):
            assert rel(cache.key_cache[i], oc.key_cache[i]) < 1.5e-2 and rel(cache.value_cache[i], oc.value_cache[i]) < 1.5e-2, (stage, i)
    if rng.random() < 0.5:
        si = model.prepare_start_tokens(lens, ropes, ids)
        n = rng.randint(2, 5)
        otoks, ologits = O.generate_text(W, cfg, oc, si["packed_key_value_indexes"], si["key_values_lens"], si["packed_start_tokens"],
                                         si["packed_query_position_ids"], n, return_logits=True)
        toks = model.generate_text(past_key_values=cache, max_length=n, do_sample=False, end_token_id=None, use_graph=False, **si)
        _tokens_match(toks, otoks, ologits, f"seed {seed}")
        assert cache.lens(0) == [x + n for x in lens]
    else:
        sizes = [(16 * rng.randint(1, 4), 16 * rng.randint(1, 4)) for _ in range(B)]
        torch.manual_seed(seed)
        li = model.prepare_vae_latent(lens, ropes, sizes, ids)
        ci = model.prepare_vae_latent_cfg([0] * B, [0] * B, sizes)
        renorm = rng.choice(["global", "channel"])
        model.cfg_batched = rng.random() < 0.5
        lat = model.generate_image(past_key_values=cache, num_timesteps=4, timestep_shift=3.0, cfg_text_scale=4.0, cfg_interval=[0.0, 1.0],
                                   cfg_renorm_type=renorm, **cfg_kwargs("cfg_text", new_cache(cfg), ci), **li)
        ocfg = dict(cache=O.OracleCache(L), position_ids=ci["cfg_packed_position_ids"], query_indexes=ci["cfg_packed_query_indexes"],
                    key_values_lens=ci["cfg_key_values_lens"], key_value_indexes=ci["cfg_packed_key_value_indexes"])
        ref = O.generate_image(W, cfg, li, oc, cfg_text=ocfg, num_timesteps=4, timestep_shift=3.0, cfg_text_scale=4.0,
                               cfg_interval=[0.0, 1.0], cfg_renorm_type=renorm)
        for a, b in zip(lat, ref):
            assert a.shape == b.shape and rel(a, b) < 3e-2, (seed, renorm, model.cfg_batched)


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_training_batches_match_oracle(monkeypatch, seed):
    """Differential fuzz of the training forward's host logic (TrainPlan: the causal / full / noise block mask as per-split
    sequences with overlapping clean-key prefixes; und / gen routing; per-image timestep rows; loss row selection): random packed
    batches (1-3 samples, 1-5 splits: prompts, ViT images, clean and noised VAE images of random sizes, CE / MSE losses anywhere)
    through Bagel.forward (stand-in operators, both mask APIs) vs the oracle's restatement of the reference."""
    import random
    from oracle import bagel_oracle as O
    from tests.util_models import pack_training_batch
    mock_ops.install(monkeypatch)
    rng = random.Random(1000 + seed)
    cfg = TINY if seed % 2 == 0 else TINY_D128
    samples = []
    for _ in range(rng.randint(1, 3)):
        sp = []
        for _ in range(rng.randint(1, 5)):
            kind = rng.choice(["text", "text", "vit", "vae", "vae"])
            if kind == "text":
                sp.append(("text", rng.randint(1, 7), rng.random() < 0.5))
            elif kind == "vit":
                sp.append(("vit", 14 * rng.randint(1, 4), 14 * rng.randint(1, 4)))
            else:
                sp.append(("vae", 16 * rng.randint(1, 4), 16 * rng.randint(1, 4), rng.random() < 0.6))
        samples.append(sp)
    if not any(s[0] == "vae" and s[3] for sp in samples for s in sp):
        samples[-1].append(("vae", 32, 48, True))          # the oracle's restatement expects at least one MSE target
    batch, noise, split_lens, attn_modes = pack_training_batch(cfg, samples, seed)
    model = cpu_model(cfg)
    W, _ = oracle_weights(cfg)
    ref = O.bagel_forward_train(W, cfg, batch, noise, timestep_shift=cfg["bagel"]["timestep_shift"])
    for api in ("nested", "splits"):
        b = dict(batch)
        if api == "splits":
            b.pop("nested_attention_masks")
            b.update(split_lens=split_lens, attn_modes=attn_modes)
        out = model(noise=noise, **b)
        assert out["mse"].shape == ref["mse"].shape and rel(out["mse"], ref["mse"]) < 5e-2, (seed, api, samples)
        if ref["ce"] is not None:
            assert out["ce"].shape == ref["ce"].shape and rel(out["ce"], ref["ce"]) < 2e-2, (seed, api, samples)
        else:
            assert out["ce"] is None


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_edit_flows_match_oracle(monkeypatch, seed):
    """Differential fuzz of the edit-style flow: 1-2 requests with source images of random (different) sizes -> VAE-encode prefill
    (padded image batch, per-image latent grids, gen-mode prefill with t = 0) [+ SigLIP prefill] + prompt, the three CFG contexts,
    then the 3-forward sampler (random renorm type, sequential or stream-batched) -- product engines vs the oracle."""
    import random
    from oracle import bagel_oracle as O
    mock_ops.install(monkeypatch)
    rng = random.Random(500 + seed)
    cfg = TINY if seed % 2 == 0 else TINY_D128
    model, vae = cpu_model_and_vae(cfg)
    W, VW = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ids = NEW_TOKEN_IDS_TINY
    ident = lambda t: t  # noqa: E731
    B = rng.randint(1, 2)
    g = torch.Generator().manual_seed(seed)
    z = [0] * B
    imgs = [torch.rand(3, 16 * rng.randint(1, 3), 16 * rng.randint(1, 3), generator=g) * 2 - 1 for _ in range(B)]
    Hm, Wm = max(i.shape[1] for i in imgs), max(i.shape[2] for i in imgs)
    enc_noise = torch.randn(B, cfg["vae"]["z_channels"], Hm // 8, Wm // 8, generator=g)

    class FixedNoiseVae:
        def encode(self, x):
            return vae.encode(x, sample_noise=enc_noise)
    vi, l1, r1 = model.prepare_vae_images(z, z, imgs, ident, ids)
    cache = model.forward_cache_update_vae(FixedNoiseVae(), new_cache(cfg), **vi)
    oc = O.forward_cache_update_vae(W, cfg, VW, O.OracleCache(L), sample_noise=enc_noise, **vi)
    if rng.random() < 0.6:
        vimgs = [torch.rand(3, 14 * rng.randint(1, 3), 14 * rng.randint(1, 3), generator=g) * 2 - 1 for _ in range(B)]
        ti, l1, r1 = model.prepare_vit_images(l1, r1, vimgs, ident, ids)
        cache = model.forward_cache_update_vit(cache, **ti)
        oc = O.forward_cache_update_vit(W, cfg, oc, **ti)
    for i in range(L):
        assert rel(cache.key_cache[i], oc.key_cache[i]) < 1.5e-2 and rel(cache.value_cache[i], oc.value_cache[i]) < 1.5e-2
    ctext, octext = copy.deepcopy(cache), oc.clone()
    prompts = [rng.choice(["make it blue", "a b", "remove the cube please"]) for _ in range(B)]
    pi, l3, r3 = model.prepare_prompts(l1, r1, prompts, tok, ids)
    cache = model.forward_cache_update_text(cache, **pi)
    oc = O.forward_cache_update_text(W, cfg, oc, **pi)
    pi2, l4, r4 = model.prepare_prompts(z, z, prompts, tok, ids)
    cimg = model.forward_cache_update_text(new_cache(cfg), **pi2)
    ocimg = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **pi2)
    sizes = [(16 * rng.randint(1, 3), 16 * rng.randint(1, 3)) for _ in range(B)]
    torch.manual_seed(seed)
    li = model.prepare_vae_latent(l3, r3, sizes, ids)
    ct, cim = model.prepare_vae_latent_cfg(l1, r1, sizes), model.prepare_vae_latent_cfg(l4, r4, sizes)
    kw = dict(num_timesteps=4, timestep_shift=3.0, cfg_text_scale=4.0, cfg_img_scale=2.0, cfg_interval=[0.0, 1.0],
              cfg_renorm_type=rng.choice(["text_channel", "global", "channel"]))
    model.cfg_batched = rng.random() < 0.5
    lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", ctext, ct), **cfg_kwargs("cfg_img", cimg, cim), **kw, **li)

    def od(c, d):
        return dict(cache=c, position_ids=d["cfg_packed_position_ids"], query_indexes=d["cfg_packed_query_indexes"],
                    key_values_lens=d["cfg_key_values_lens"], key_value_indexes=d["cfg_packed_key_value_indexes"])
    ref = O.generate_image(W, cfg, li, oc, cfg_text=od(octext, ct), cfg_img=od(ocimg, cim), **kw)
    for a, b in zip(lat, ref):
        assert a.shape == b.shape and rel(a, b) < 4e-2, (seed, kw["cfg_renorm_type"], model.cfg_batched)


def test_fp32_master_weights_are_cast_once(golden, monkeypatch):
    """eval/gen/gen_images_mp.py:174 keeps fp32 weights and relies on autocast: the first hot-path call casts them to bf16 in place
    (with a warning) and the run equals the bf16-weights run bit for bit."""
    from bagel_amd.factory import build_bagel
    mock_ops.install(monkeypatch)
    cfg = TINY
    g = golden("tiny_t2i")
    W, _ = oracle_weights(cfg)
    model32, _ = build_bagel(cfg, device="cpu", dtype=torch.float32, with_vae=False)
    model32.load_state_dict(W, strict=True)
    assert model32.llm2vae.weight.dtype == torch.float32
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    outs = []
    for m in (model32.eval(), cpu_model(cfg)):
        gi, _, _ = m.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
        if m is model32:
            with pytest.warns(UserWarning, match="cast to bfloat16"):
                cache = m.forward_cache_update_text(new_cache(cfg), **gi)
            assert m.llm2vae.weight.dtype == torch.bfloat16
        else:
            cache = m.forward_cache_update_text(new_cache(cfg), **gi)
        outs.append(m.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **g["gen_kwargs"],
                                     **g["latent_inputs"]))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_chat_entry_matches_oracle(monkeypatch):
    """Bagel.chat (bagel.py:1004-1074, the eval entry): ViT prefill per image -> prompt prefill -> greedy decode until <|im_end|> ->
    tokenizer.decode and the marker split -- vs the same chain restated with the oracle's functions."""
    from oracle import bagel_oracle as O
    mock_ops.install(monkeypatch)
    monkeypatch.setenv("BAGEL_DECODE_GRAPH", "0")
    cfg = TINY
    model = cpu_model(cfg)
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ids = NEW_TOKEN_IDS_TINY
    ident = lambda t: t  # noqa: E731
    g = torch.Generator().manual_seed(11)
    images = [torch.rand(3, 28, 28, generator=g) * 2 - 1, torch.rand(3, 14, 42, generator=g) * 2 - 1]
    out = model.chat(tok, ids, ident, images, "what is it", max_length=6)
    oc, lens, ropes = O.OracleCache(L), [0], [0]
    for im in images:
        gi, lens, ropes = model.prepare_vit_images(lens, ropes, [im], ident, ids)
        oc = O.forward_cache_update_vit(W, cfg, oc, **gi)
    gi, lens, ropes = model.prepare_prompts(lens, ropes, ["what is it"], tok, ids)
    oc = O.forward_cache_update_text(W, cfg, oc, **gi)
    si = model.prepare_start_tokens(lens, ropes, ids)
    toks, logits = O.generate_text(W, cfg, oc, si["packed_key_value_indexes"], si["key_values_lens"], si["packed_start_tokens"],
                                   si["packed_query_position_ids"], 6, end_token_id=ids["eos_token_id"], return_logits=True)
    ref = tok.decode(toks[:, 0]).split("<|im_end|>")[0].split("<|im_start|>")[1]
    import re
    a, b = re.findall(r"\[(\d+)\]", out), re.findall(r"\[(\d+)\]", ref)
    first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
    assert isinstance(out, str) and first >= 1 and a[:first] == b[:first], (out, ref)


@pytest.mark.parametrize("interval", [[0.0, 1.0], [0.45, 1.0]], ids=["cfg_everywhere", "cfg_off_late"])
def test_stream_batched_cfg_with_taylorseer_equals_sequential(monkeypatch, interval):
    """enable_taylorseer=True under model.cfg_batched: one TaylorSeer state per stream inside the batched forward (full steps refresh
    every stream's cache from its slice, Taylor steps replace the slice by the stream's extrapolation; with cfg_interval switching
    CFG off for the late steps the streams' schedules drift apart and a batched forward mixes both kinds) == the sequential path
    bit for bit, and the schedules (full / Taylor counts per stream) are the same."""
    mock_ops.install(monkeypatch)
    cfg = TINY
    model = cpu_model(cfg)
    sizes = [(32, 32), (16, 48)]
    c0, li, (c1, ct), _ = _three_stream_setup(model, cfg, sizes)
    kw = dict(num_timesteps=14, timestep_shift=3.0, cfg_text_scale=4.0, cfg_interval=interval, cfg_renorm_min=0.0,
              cfg_renorm_type="global", enable_taylorseer=True, **cfg_kwargs("cfg_text", c1, ct))
    model.cfg_batched = False
    ref = model.generate_image(past_key_values=copy.deepcopy(c0), **kw, **li)
    sched_ref = [(s.full_steps, s.taylor_steps) for s in model._last_taylor_states[:2]]
    model.cfg_batched, model.und_side_path = True, True
    got = model.generate_image(past_key_values=copy.deepcopy(c0), **kw, **li)
    sched = [(s.full_steps, s.taylor_steps) for s in model._last_taylor_states[:2]]
    assert sched == sched_ref and sched[0][1] > 0, (sched, sched_ref)
    if interval[0] > 0:
        assert sched[0] != sched[1], "the streams were meant to drift apart"
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_understanding_only_model_chat(monkeypatch):
    """The construction recipe of eval/vlm/utils.py:30-63 -- BagelConfig(visual_gen=False, visual_und=True) (no VAE config, no
    latent modules), fp32 weights from a full checkpoint via load_state_dict(strict=False), .eval() -- then model.chat with a PIL
    image through the ImageTransform, as the VLM benchmark drivers call it (eval/vlm/eval/mme/eval.py:61-68).  The answer must equal
    the full model's (the und path does not touch the generation modules)."""
    import numpy as np
    from PIL import Image
    from bagel_amd.data.transforms import ImageTransform
    from bagel_amd.modeling.bagel import Bagel, BagelConfig, Qwen2Config, Qwen2ForCausalLM, SiglipVisionConfig, SiglipVisionModel
    mock_ops.install(monkeypatch)
    monkeypatch.setenv("BAGEL_DECODE_GRAPH", "0")
    cfg = TINY
    W, _ = oracle_weights(cfg)
    llm_config, vit_config = Qwen2Config(**cfg["llm"]), SiglipVisionConfig(**cfg["vit"])
    config = BagelConfig(visual_gen=False, visual_und=True, llm_config=llm_config, vit_config=vit_config,
                         vit_max_num_patch_per_side=cfg["bagel"]["vit_max_num_patch_per_side"], connector_act="gelu_pytorch_tanh")
    model = Bagel(Qwen2ForCausalLM(llm_config), SiglipVisionModel(vit_config), config)
    model.vit_model.vision_model.embeddings.convert_conv2d_to_linear(vit_config)
    msg = model.load_state_dict(W, strict=False)
    assert not msg.missing_keys and all(k.split(".")[0] in ("vae2llm", "llm2vae", "time_embedder", "latent_pos_embed") for k in msg.unexpected_keys)
    assert not hasattr(model, "vae2llm")
    model = model.eval()
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    tf = ImageTransform(56, 28, 14, device="cpu")
    img = Image.fromarray(np.random.default_rng(3).integers(0, 256, (60, 80, 3), dtype=np.uint8), "RGB")
    with pytest.warns(UserWarning, match="cast to bfloat16"):
        out = model.chat(tok, NEW_TOKEN_IDS_TINY, tf, images=[img], prompt="what is it", max_length=6)
    full = cpu_model(cfg).chat(tok, NEW_TOKEN_IDS_TINY, tf, images=[img], prompt="what is it", max_length=6)
    assert isinstance(out, str) and out == full and out.startswith("[")


def test_packed_weights_follow_a_reload_through_the_parent(monkeypatch):
    """The engines keep packed copies of the weights (fused Wqkv, interleaved gate/up).  After a forward, loading ANOTHER checkpoint
    through the parent Bagel (gen_images_mp.py:165-176 does exactly that), or rewriting parameters in place (EMA swap), must be
    seen by the next call: the result has to equal a fresh model built from the new weights -- not a mixture of old packed and
    new plain tensors."""
    mock_ops.install(monkeypatch)
    cfg = TINY
    from bagel_amd.factory import build_bagel
    W, _ = oracle_weights(cfg)
    g = torch.Generator().manual_seed(11)
    W2 = {k: (v.float() + 0.05 * v.float().std().nan_to_num(0.0).clamp_min(1e-3) * torch.randn(v.shape, generator=g)).to(v.dtype) if v.is_floating_point() and "pos_embed" not in k else v
          for k, v in W.items()}
    tok = StubTokenizer(cfg["llm"]["vocab_size"])

    def prefill(model):
        gi, _, _ = model.prepare_prompts([0, 0], [0, 0], ["a small red cube", "sky"], tok, NEW_TOKEN_IDS_TINY)
        c = model.forward_cache_update_text(new_cache(cfg), **gi)
        return [c.key_cache[i].clone() for i in range(cfg["llm"]["num_hidden_layers"])]

    def fresh(weights):
        m, _ = build_bagel(cfg, device="cpu", with_vae=False)
        m.load_state_dict(weights, strict=True)
        return m.to(torch.bfloat16).eval()

    want1, want2 = prefill(fresh(W)), prefill(fresh(W2))
    assert not torch.equal(want1[-1], want2[-1])
    model = fresh(W)
    assert all(torch.equal(a, b) for a, b in zip(prefill(model), want1))
    model.load_state_dict({k: v.to(torch.bfloat16) if v.is_floating_point() else v for k, v in W2.items()}, strict=True)   # parent load
    assert all(torch.equal(a, b) for a, b in zip(prefill(model), want2)), "stale packed weights after Bagel.load_state_dict"
    with torch.no_grad():                                                                 # in-place rewrite (EMA swap)
        sd = {k: v.to(torch.bfloat16) if v.is_floating_point() else v for k, v in W.items()}
        for k, p in model.named_parameters():
            p.copy_(sd[k])
    assert all(torch.equal(a, b) for a, b in zip(prefill(model), want1)), "stale packed weights after an in-place parameter rewrite"
    sd2 = {k: v.to(torch.bfloat16) if v.is_floating_point() else v for k, v in W2.items()}
    for k, p in model.named_parameters():                                                 # re-seated storage
        p.data = sd2[k].clone()
    assert all(torch.equal(a, b) for a, b in zip(prefill(model), want2)), "stale packed weights after param.data = ..."
    # writes through a detached alias (param.data.copy_) bump no version counter and keep the pointer: the documented route for
    # those is an explicit invalidate_packed() (what init_moe does itself)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.data.copy_(sd[k])
    model.language_model.invalidate_packed()
    assert all(torch.equal(a, b) for a, b in zip(prefill(model), want1))


def test_model_built_under_inference_mode_runs(golden, monkeypatch):
    """Inference tensors keep no version counter: the packed-weight signature (modeling/packed.py) must not read `_version`
    on them -- every public entry point calls _check_packed, so a model built, loaded or moved inside torch.inference_mode()
    would otherwise fail with 'Inference tensors do not track version counter'."""
    mock_ops.install(monkeypatch)
    cfg = TINY
    g = golden("tiny_t2i")
    with torch.inference_mode():
        model = cpu_model(cfg)
        assert all(p.is_inference() for p in model.parameters())
        tok = StubTokenizer(cfg["llm"]["vocab_size"])
        gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
        cache = model.forward_cache_update_text(new_cache(cfg), **gi)
        sig = model.language_model._signature()
        cache2 = model.forward_cache_update_text(new_cache(cfg), **gi)          # second call: signature compared, packed copies kept
        assert model.language_model._signature() == sig
    for i in range(cfg["llm"]["num_hidden_layers"]):
        assert rel(cache.key_cache[i], g["key_cache"][i]) < 1e-2
        assert torch.equal(cache.key_cache[i], cache2.key_cache[i])


def test_generate_text_runs_without_autograd_and_refuses_the_llm_int8_name(monkeypatch):
    """``generate_text`` had lost its ``@torch.no_grad()`` / ``@_bf16_weights`` decorators to a function inserted above it.  Every public inference
    entry point runs with autograd OFF even when the caller left it on; and the name "int8" -- the reference's LLM.int8 load mode (app.py:126-131), which is not
    built -- is refused everywhere instead of silently selecting the row-wise option."""
    from bagel_amd.modeling.bagel import decode as decode_mod
    mock_ops.install(monkeypatch)
    cfg = TINY_D128
    model = cpu_model(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, lens, ropes = model.prepare_prompts([0], [0], ["a small red cube"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    si = model.prepare_start_tokens(lens, ropes, NEW_TOKEN_IDS_TINY)
    seen = []
    real_init = decode_mod.DecodeSession.__init__

    def spy(self, *a, **k):
        seen.append(torch.is_grad_enabled())
        return real_init(self, *a, **k)
    monkeypatch.setattr(decode_mod.DecodeSession, "__init__", spy)
    with torch.enable_grad():
        toks = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=3, do_sample=False, end_token_id=None, use_graph=False, **si)
        assert torch.is_grad_enabled()
    assert seen == [False], "generate_text must run under torch.no_grad()"
    assert toks.shape == (3, 1) and not toks.requires_grad
    with pytest.raises(NotImplementedError, match="LLM.int8"):
        model.generate_text(past_key_values=copy.deepcopy(cache), max_length=2, do_sample=False, end_token_id=None, use_graph=False, weight_quant="int8", **si)
    with pytest.raises(NotImplementedError, match="LLM.int8"):
        model.quantize_language_model("int8")
    assert getattr(model.language_model, "weight_store", None) is None, "a refused mode must not leave the model half-switched"


def test_naive_cache_deepcopy_copy_on_write_semantics(monkeypatch):
    """``copy.deepcopy(NaiveCache)`` (inferencer.py:189,230-231,244,253) shares the layers' buffers until one side writes: the copies are equal, a prefill appended
    to the COPY leaves the original untouched (rows, lengths, buffer identity) and the original can still be extended on its own afterwards; ``concat`` with one live
    stream shares as well.  (Memory is asserted on the device: tests/test_decode_gpu.py.)"""
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    mock_ops.install(monkeypatch)
    cfg = TINY_D128
    model = cpu_model(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, lens, ropes = model.prepare_prompts([0], [0], ["a small red cube on a table"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    snap = [(cache.key_cache[i].clone(), cache.value_cache[i].clone()) for i in range(L)]
    c1, c2 = copy.deepcopy(cache), copy.deepcopy(cache)
    for c in (c1, c2):
        assert c.seq_lens == cache.seq_lens
        for i in range(L):
            assert c._k[i] is cache._k[i] and cache._own[i][0] == 3                  # shared, three owners
            assert torch.equal(c.key_cache[i], snap[i][0]) and torch.equal(c.value_cache[i], snap[i][1])
    joined = NaiveCache.concat([c2, None], [1, 1])
    assert joined._k[0] is cache._k[0] and joined.lens(0) == [lens[0], 0]
    # extend the first copy: it must move to buffers of its own
    gi2, lens2, ropes2 = model.prepare_prompts(lens, ropes, ["and a blue ball"], tok, NEW_TOKEN_IDS_TINY)
    c1 = model.forward_cache_update_text(c1, **gi2)
    assert c1.seq_lens == lens2[0] > lens[0]
    for i in range(L):
        assert c1._k[i] is not cache._k[i] and c1._own[i][0] == 1
        assert torch.equal(cache.key_cache[i], snap[i][0]) and torch.equal(cache.value_cache[i], snap[i][1]), "the original changed under a write to its copy"
        assert torch.equal(c2.key_cache[i], snap[i][0])
        assert torch.equal(c1.key_cache[i][: lens[0]], snap[i][0])
    # the original, extended by the same prompt, arrives at the same rows
    cache2 = model.forward_cache_update_text(cache, **gi2)
    for i in range(L):
        assert torch.equal(cache2.key_cache[i], c1.key_cache[i]) and torch.equal(cache2.value_cache[i], c1.value_cache[i])
        assert torch.equal(c2.key_cache[i], snap[i][0]), "the remaining sharer changed"


def test_bench_depth_parity_legs_run_on_the_host_path(monkeypatch):
    """bench.py's two round-5 parity legs -- ``edit_depth_step`` (the 3-forward edit step: cond / CFG-text / CFG-img, text_channel renorm) and the CFG-combine
    self-consistency gate of ``full_depth_step`` -- exercised end to end on the tiny model with the stand-in operators: the plumbing (contexts, packers, the oracle
    calls, the gates) is covered without a GPU; the numbers at 7B width are tests/test_full_depth_gpu.py's."""
    import argparse
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module_depth", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    mock_ops.install(monkeypatch)
    cfg = TINY_D128
    model = cpu_model(cfg)
    ids = NEW_TOKEN_IDS_TINY
    args = argparse.Namespace(resolution=64)
    monkeypatch.setattr(torch, "set_num_threads", lambda n: None)      # the legs size torch's thread team for the bench box: other tests' bit-exact golden checks depend on it
    out = bench.edit_depth_step(args, cfg, model, ids, threads=2, ctx_tokens=(20, 6))
    assert out["contexts"] == [20 + 2 + 6 + 2, 20 + 2, 6 + 2]
    for k in ("rel_l2_cond_forward", "rel_l2_cfg_text_forward", "rel_l2_cfg_img_forward"):
        assert out[k] <= 2e-2, (k, out[k])
    assert out["cfg_combine_self_consistency"]["sequential_forward_flow"] <= bench.FULL_DEPTH_TOL_COMBINE
    tok = bench.FixedTokenizer(list(range(8, 20)))
    out2 = bench.full_depth_step(args, cfg, model, tok, ids, threads=2)
    sc = out2["cfg_combine_self_consistency"]
    assert sc["sequential_forward_flow"] <= bench.FULL_DEPTH_TOL_COMBINE and sc["generate_image_sequential"] <= bench.FULL_DEPTH_TOL_COMBINE
    assert out2["rel_l2_cond_forward"] <= 2e-2 and out2["rel_l2_cfg_text_forward"] <= 2e-2
    # round 6: the per-stream velocities of the stream-batched forward (the timed path) are taken out before the combine and gated like single forwards
    for o_ in (out, out2):
        sb = o_["stream_batched"]
        assert sb["ran_batched"] and max(v for k, v in sb.items() if k.startswith("rel_l2_")) <= 2e-2, sb
    assert max(out["context_kv_rel_l2_max"].values()) <= 2e-2
    assert model.velocity_hook is None


def test_bench_edit_request_parity_leg_runs_the_real_chain_on_the_host_path(monkeypatch):
    """``bench.edit_depth_step(vae=...)`` -- the REAL request chain of configs[4] (VAE-encode -> gen-mode prefill -> SigLIP -> und-mode prefill -> prompt, then the
    3-forward step) -- on the tiny model with the stand-in operators, in its three phases: inputs (host), oracle (here computed separately, as the bench's worker
    process does, and handed in as ``oracle_out``), product + compare; and the weight fingerprint refuses an oracle that ran on other weights."""
    import argparse
    import importlib.util
    import os
    from bagel_amd.factory import build_bagel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module_edit_request", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    mock_ops.install(monkeypatch)
    cfg = TINY_D128
    W, VW = oracle_weights(cfg)
    model, vae = build_bagel(cfg, device="cpu", with_vae=True)
    model.load_state_dict(W, strict=True)
    vae.load_state_dict(VW, strict=True)
    model = model.to(torch.bfloat16).eval()
    ids = NEW_TOKEN_IDS_TINY
    args = argparse.Namespace(resolution=64)
    monkeypatch.setattr(torch, "set_num_threads", lambda n: None)
    side = cfg["vit"]["patch_size"] * 3
    monkeypatch.setattr(bench.edit_depth_inputs, "__defaults__", ((576, 30), False, 5, side))     # ctx_tokens, real, prompt_tokens, vit_side: the tiny tower's image
    inp = bench.edit_depth_inputs(args, cfg, model, ids, real=True)
    n_vae, n_vit = (64 // 16) ** 2 + 2, 9 + 2
    assert [int(inp[k][0][0]) for k in ("cond", "text", "img")] == [n_vae + n_vit + 7, n_vae + n_vit, 7]
    Wk = {k: v for k, v in model.state_dict().items() if k.startswith(bench.EDIT_KEEP_REAL)}
    ora = bench.edit_depth_oracle(cfg, Wk, {k: v.float() for k, v in vae.state_dict().items()}, inp, threads=2)
    out = bench.edit_depth_step(args, cfg, model, ids, threads=2, vae=vae, oracle_out=ora)
    assert out["contexts"] == [n_vae + n_vit + 7, n_vae + n_vit, 7] and "worker process" in out["oracle_ran"]
    # the oracle phase takes its three single-forward velocities out of ONE 3-forward pass (forward_flow's `parts`): the same bits as a stand-alone forward on that context
    from oracle import bagel_oracle as O
    li, ct = inp["latent"], inp["cfg_text"]
    x0 = li["packed_init_noises"]
    lis = dict(li, packed_position_ids=ct["cfg_packed_position_ids"], packed_indexes=ct["cfg_packed_query_indexes"], key_values_lens=ct["cfg_key_values_lens"],
               packed_key_value_indexes=ct["cfg_packed_key_value_indexes"])
    octext = O.OracleCache(cfg["llm"]["num_hidden_layers"])
    for i, (k_, v_) in enumerate(ora["kv"]["cfg_text"]):
        octext.key_cache[i], octext.value_cache[i] = k_, v_
    alone = O.forward_flow(Wk, cfg, x0, torch.tensor([1.0] * x0.shape[0]), lis, octext, None, None, 1.0, 1.0, 0.0, "global").float()
    assert torch.equal(alone, ora["o_t"])
    assert max(out["context_kv_rel_l2_max"].values()) <= 3e-2, out["context_kv_rel_l2_max"]
    for k in ("rel_l2_cond_forward", "rel_l2_cfg_text_forward", "rel_l2_cfg_img_forward"):
        assert out[k] <= 3e-2 and out["stream_batched"][k] <= 3e-2, (k, out[k], out["stream_batched"][k])
    assert out["cfg_combine_self_consistency"]["sequential_forward_flow"] <= bench.FULL_DEPTH_TOL_COMBINE
    # an oracle result from OTHER weights is refused
    bad = dict(ora, weights={k: v + 1.0 for k, v in ora["weights"].items()})
    with pytest.raises(RuntimeError, match="other weights"):
        bench.edit_depth_step(args, cfg, model, ids, threads=2, vae=vae, oracle_out=bad)
    # the understanding leg's three phases the same way
    uargs = argparse.Namespace(und_image=side)
    monkeypatch.setattr(bench, "und_request", lambda a: (torch.rand(3, side, side, generator=torch.Generator().manual_seed(2)) * 2 - 1,      # (the bench's prompt ids are drawn from the 7B vocabulary)
                                                          bench.FixedTokenizer(torch.randint(8, 500, (32,), generator=torch.Generator().manual_seed(1)).tolist())))
    uin = bench.und_depth_inputs(uargs, model, ids)
    uW = {k: v for k, v in model.state_dict().items() if k.startswith(bench.UND_KEEP)}
    uora = bench.und_depth_oracle(cfg, uW, uin, threads=2, n_tokens=3)
    monkeypatch.setattr(bench, "UND_DEPTH_TOL_KV", 3e-2)
    monkeypatch.setattr(bench, "UND_DEPTH_TOL_LOGITS", 3e-2)
    gt = model.generate_text
    monkeypatch.setattr(model, "generate_text", lambda **kw: gt(use_graph=False, **kw))           # no hipGraph on the host path
    uout = bench.understanding_full_depth(uargs, cfg, model, ids, threads=2, oracle_out=uora)
    assert uout["kv_rel_l2_max"] <= 3e-2 and uout["first_step_logits_rel_l2"] <= 3e-2 and uout["context_tokens"] == 9 + 2 + 34, uout


def test_bench_steady_bracket_and_budget_guard_without_a_gpu():
    import argparse
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module_steady", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with bench.Steady("cpu") as m:
        pass
    assert m.report["steady"] is True and m.report["peak_mem_gb"] is None
    r, dt, rep = bench.timed_steady(lambda: 7, "cpu", lambda: None)
    assert r == 7 and dt >= 0 and rep["attempts"] == 1
    leg = bench.unsteady({"value": 1.0}, {"steady": False, "device_allocs_in_timed_region": 2, "device_frees_in_timed_region": 0, "attempts": 2})
    assert "error" in leg and "hipMalloc" in leg["error"]
    a = argparse.Namespace(wall_budget_s=10.0)
    assert bench.over_budget(a, 1e9, "x")["skipped"].startswith("x:") and bench.over_budget(argparse.Namespace(wall_budget_s=1e9), 5, "x") is None
    nodes = bench.numa_nodes()
    assert isinstance(nodes, list) and all(isinstance(n, list) for n in nodes)
    assert bench._cpu_tree({"a": (torch.ones(2), [1, 2]), "b": 3})["a"][1] == [1, 2]
    # a worker process (anything whose command line carries --oracle-job) is STOPPED for the duration of a timed region and resumed after it; a pid that is not
    # (or no longer) a worker is never signalled (a finished worker's pid may have been recycled)
    import subprocess
    import sys
    import time
    p = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(60)", "--oracle-job"])
    state = lambda: open(f"/proc/{p.pid}/stat").read().rsplit(")", 1)[1].split()[0]  # noqa: E731
    try:
        time.sleep(0.3)
        bench.Steady.pause_pids[:] = [p.pid, os.getpid()]
        with bench.Steady("cpu"):
            time.sleep(0.1)
            assert state() == "T" and bench.Steady.pause_pids == [p.pid]        # (this test process was dropped from the list, not stopped)
        time.sleep(0.1)
        assert state() in "SR"
    finally:
        p.kill()
        p.wait()
    with bench.Steady("cpu"):
        pass
    assert bench.Steady.pause_pids == []
