"""GPU parity of the NF4 weight-only option of the decode projections -- the 4-bit load mode the reference itself ships (app.py:114-125:
bitsandbytes quant_type "nf4", blocks of 64, fp32 absmax, no double quantisation, bf16 compute) -- against its CPU restatement
(oracle/nf4.py, written from the library's published algorithm; the library itself is an un-vendored CUDA dependency):
the quantiser bit for bit (codes, nibble order, absmax), the W4A16 projection against the DE-QUANTISED BF16 PRODUCT (what the
reference's Linear4bit computes) for every epilogue and shape of the decode layer, and the whole decode step against the oracle's
decode with the scheme switched into the same seven linears per layer."""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import nf4 as NF
from tests.test_decode_gpu import ref_rmsnorm
from tests.test_ops_gpu import BF16, DEV, close, ops, ref_gemm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(5, 64), (37, 640), (300, 3584), (33, 18944)])
def test_quantize_nf4_bit_exact(rows, cols):
    g = torch.Generator().manual_seed(rows)
    w = (torch.randn(rows, cols, generator=g) * torch.rand(rows, 1, generator=g) * 0.2).to(BF16)
    w[0, :64] = 0                                                    # an all-zero block: absmax 0, every code 0
    w[1, :16] = (NF.NF4_CODE * 0.75).to(BF16)                        # near code-book values
    w[2, :64] = 3.0e38
    if rows > 3:
        t = NF.NF4_THRESH                                            # values on and next to the decision thresholds (absmax = 1 in the block)
        w[3, 0] = 1.0
        w[3, 1:16] = t.to(BF16)
        w[3, 16:31] = torch.nextafter(t, torch.ones(15)).to(BF16)
    q, a = ops().quantize_nf4(w.to(DEV))
    qr, ar = NF.quantize_nf4(w)
    assert torch.equal(a.cpu(), ar), "block absmax differs"
    assert torch.equal(q.cpu(), qr), f"{(q.cpu() != qr).sum().item()} of {qr.numel()} code bytes differ"


@pytest.mark.parametrize("mode", ["plain", "bias", "residual", "swiglu", "norm", "norm_swiglu"])
@pytest.mark.parametrize("M,N,K", [(1, 64, 128), (1, 4608, 3584), (2, 3584, 3584), (1, 3584, 18944), (1, 37888, 3584), (4, 512, 1024), (3, 130, 192)])
def test_gemv_nf4_matches_dequantised_bf16_product(M, N, K, mode):
    if N == 37888 and mode not in ("swiglu", "norm_swiglu"):
        pytest.skip("the gate+up shape only carries the SwiGLU epilogues in the model")
    swiglu = mode.endswith("swiglu")
    if swiglu and N % 32:
        pytest.skip("SwiGLU16 pairing needs N % 32 == 0")
    g = torch.Generator().manual_seed(N + K + M)
    x = (torch.randn(M, K, generator=g) * 1.5).to(BF16)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF16)
    o = ops()
    q, a = o.quantize_nf4(W.to(DEV))
    qr, ar = NF.quantize_nf4(W)
    assert torch.equal(q.cpu(), qr) and torch.equal(a.cpu(), ar)
    Wd = NF.dequantize_nf4(qr, ar)                                   # bf16: the weight the reference's Linear4bit multiplies with
    bias = (torch.randn(N, generator=g) * 0.1).to(BF16) if mode == "bias" else None
    nw = (1.0 + 0.1 * torch.randn(K, generator=g)).to(BF16) if mode.startswith("norm") else None
    Nout = N // 2 if swiglu else N
    R = torch.randn(M, Nout, generator=g).to(BF16) if mode == "residual" else None
    C = R.to(DEV).clone() if R is not None else torch.full((M, Nout), float("nan"), dtype=BF16, device=DEV)
    o.gemv_nf4(x.to(DEV), q, a, C, bias=None if bias is None else bias.to(DEV), residual=C if R is not None else None,
               epilogue=o.EPI_SWIGLU16 if swiglu else o.EPI_NONE, norm_w=None if nw is None else nw.to(DEV), eps=1e-6)
    h = ref_rmsnorm(x, nw, 1e-6) if nw is not None else x
    ref = ref_gemm(h, Wd, bias, 3 if swiglu else 0, R)
    # (norm + projection + SwiGLU = three chained bf16 roundings: the rel-L2 budget of two chained ops, as the skinny / gemv tests use)
    close(C.cpu(), ref, ulps=2, rel_l2=8e-3 if (swiglu or nw is not None) else 4e-3, what=f"gemv_nf4 {mode} M={M} N={N} K={K}")
    if mode == "plain":                                              # what 4-bit weights cost against the bf16 product: ~9 %
        full = x.float() @ W.float().t()
        e = ((C.cpu().float() - full).norm() / full.norm()).item()
        assert 0.02 < e < 0.2, e


def test_generate_text_nf4_weights_option():
    """weight_quant='nf4': same decode loop, graph replay == eager bit for bit, the first step's logits against the oracle's decode step
    with oracle/nf4.py switched into the same seven linears per layer (bf16 prefill on both sides), model-level switch."""
    from oracle import bagel_oracle as O
    from oracle.configs import TINY_D128 as cfg, NEW_TOKEN_IDS_TINY, StubTokenizer
    from tests.test_decode_gpu import _context
    from tests.util_models import oracle_weights, product_model
    model, _ = product_model(cfg)
    cache, lens, ropes, start = _context(model, cfg, ["a small red cube"])
    ref = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=1, end_token_id=None, **start)
    ref_logits = model._last_decode_session.logits.float().clone()
    one = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=1, end_token_id=None, weight_quant="nf4", **start)
    q_logits = model._last_decode_session.logits.float().clone()
    assert one.shape == ref.shape and torch.isfinite(q_logits).all()
    err = ((q_logits - ref_logits).norm() / ref_logits.norm()).item()
    assert 1e-3 < err < 0.4, f"nf4-weight logits vs bf16-weight logits: rel_l2 {err:.3g}"
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    gi, l2, r2 = model.prepare_prompts([0], [0], ["a small red cube"], StubTokenizer(cfg["llm"]["vocab_size"]), NEW_TOKEN_IDS_TINY)
    oc = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi)
    O.NF4_WEIGHT_PTRS = O.mxfp4_decode_weight_ptrs(W)
    try:
        st = {k: torch.as_tensor(v).cpu() for k, v in start.items()}
        _, ologits = O.generate_text(W, cfg, oc, st["packed_key_value_indexes"], st["key_values_lens"], st["packed_start_tokens"],
                                     st["packed_query_position_ids"], 1, return_logits=True)
    finally:
        O.NF4_WEIGHT_PTRS = set()
    ol = torch.as_tensor(ologits[0]).float().reshape(q_logits.shape)
    e2 = ((q_logits.cpu() - ol).norm() / ol.norm()).item()
    e_bf16 = ((ref_logits.cpu() - ol).norm() / ol.norm()).item()
    # the quantised weights are IDENTICAL on both sides (the quantiser is bit-exact), activations stay bf16: the tolerance is the plain
    # decode-step tolerance of the bf16 path (no activation-code noise as in the fp8 / mxfp4 options)
    assert e2 < 2e-2 and e2 < 0.5 * e_bf16, f"nf4 decode step vs its restatement: rel_l2 {e2:.3g} (bf16 path vs the same restatement: {e_bf16:.3g})"
    n = 8
    a = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, weight_quant="nf4", use_graph=True, **start)
    sess = model._last_decode_session
    assert sess.weight_quant == "nf4" and sess.graph is not None
    b = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, weight_quant="nf4", use_graph=False, **start)
    assert torch.equal(a, b)
    model.decode_weight_quant = "nf4"
    try:
        c = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, **start)
    finally:
        model.decode_weight_quant = None
    assert torch.equal(c, a)


@pytest.mark.parametrize("rows,cols", [(5, 64), (37, 640), (300, 3584), (33, 18944)])
def test_dequantize_kernels_bit_exact(rows, cols):
    """bagel_dequantize_nf4_bf16 = oracle/nf4.py's dequantize_4bit with a bf16 output (code_book[code] * absmax in fp32, ONE rounding), also into a
    wider destination; bagel_dequantize_rows_i8_bf16 = bf16((q - 128) * scale[row])."""
    g = torch.Generator().manual_seed(rows + 1)
    w = (torch.randn(rows, cols, generator=g) * torch.rand(rows, 1, generator=g) * 0.2).to(BF16)
    w[0, :64] = 0
    q, a = ops().quantize_nf4(w.to(DEV))
    d = ops().dequantize_nf4(q, a)
    assert torch.equal(d.cpu().view(torch.int16), NF.dequantize_nf4(q.cpu(), a.cpu()).view(torch.int16))
    wide = torch.full((rows + 3, cols + 64), 7.0, dtype=BF16, device=DEV)
    d2 = ops().dequantize_nf4(q, a, wide)
    assert torch.equal(d2, d) and (wide[:, cols:] == 7.0).all() and (wide[rows:] == 7.0).all()
    q8, s8 = ops().quantize_rows_i8(w.to(DEV))
    d8 = ops().dequantize_rows_i8(q8, s8)
    ref8 = ((q8.cpu().float() - 128.0) * s8.cpu()[:, None]).to(BF16)
    assert torch.equal(d8.cpu().view(torch.int16), ref8.view(torch.int16))


@pytest.mark.parametrize("kind", ["nf4", "int8_rowwise"])
def test_whole_model_quantised_load_mode(kind):
    """Bagel.quantize_language_model (app.py:114-131's load modes over the whole forward path): prefill KV and denoise latents of the quantised engine
    are BIT-IDENTICAL to the bf16 engine on the de-quantised weights (a layer's matrices are materialised with the dequantise kernel right before
    its GEMMs: bitsandbytes' dequantise-then-F.linear), the decode streams the stored codes (tokens equal up to near ties: the gemv expands codes in
    fp32), and the bf16 projection weights are released."""
    from oracle.configs import TINY_D128 as cfg, NEW_TOKEN_IDS_TINY, StubTokenizer
    from tests.test_model_gpu import cfg_kwargs, new_cache
    from tests.util_models import _product
    g = torch.load(__file__.rsplit("/", 1)[0] + "/golden/tiny_d128_t2i.pt", weights_only=False)
    qm, _ = _product.__wrapped__(cfg["name"])                     # two FRESH models: product_model() hands every test the same cached instance
    bm, _ = _product.__wrapped__(cfg["name"])
    o = ops()
    for L in bm.language_model.model.layers:                      # W <- dequantise(quantise(W)) on every decoder projection
        mods = [getattr(L.self_attn, n + s) for n in ("q_proj", "k_proj", "v_proj", "o_proj") for s in ("", "_moe_gen")]
        for s in ("", "_moe_gen"):
            m = getattr(L, "mlp" + s)
            mods += [m.gate_proj, m.up_proj, m.down_proj]
        for m in mods:
            w = m.weight.data
            m.weight.data = (o.dequantize_nf4(*o.quantize_nf4(w)) if kind == "nf4" else o.dequantize_rows_i8(*o.quantize_rows_i8(w))).contiguous()
    bm.language_model.invalidate_packed()
    free0 = torch.cuda.memory_allocated()
    resident = qm.quantize_language_model(kind)
    full = sum(p.numel() * 2 for n, p in bm.language_model.model.layers.named_parameters() if "proj" in n and n.endswith("weight"))
    assert resident < (0.30 if kind == "nf4" else 0.52) * full
    assert all(p.numel() == 0 for n, p in qm.language_model.model.layers.named_parameters() if "proj" in n and n.endswith("weight"))
    assert torch.cuda.memory_allocated() < free0, "releasing the bf16 projection weights must free more than the codes take"
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    outs = []
    for model in (qm, bm):
        gi, newlens, newrope = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
        cache = model.forward_cache_update_text(new_cache(cfg), **gi)
        kv = [(cache.key_cache[i].clone(), cache.value_cache[i].clone()) for i in range(cfg["llm"]["num_hidden_layers"])]
        lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **g["gen_kwargs"], **g["latent_inputs"])
        st = model.prepare_start_tokens(newlens, newrope, NEW_TOKEN_IDS_TINY)
        toks = model.generate_text(past_key_values=cache, max_length=6, do_sample=False, end_token_id=None, **st)
        outs.append((kv, lat, toks, model._last_decode_session))
    (ka, la, ta, sa), (kb, lb, tb, sb) = outs
    for (k1, v1), (k2, v2) in zip(ka, kb):
        assert torch.equal(k1, k2) and torch.equal(v1, v2), "prefill KV differs"
    for a, b in zip(la, lb):
        assert torch.equal(a, b), "latents differ"
    assert sa.weight_quant == kind and sa.graph is not None and sb.weight_quant is None
    assert ta.shape == tb.shape and torch.equal(ta[:2], tb[:2])          # first decoded token from the same prefill: the codes expanded in fp32 vs bf16 weights
    lg = (sa.logits.float() - sb.logits.float()).norm() / sb.logits.float().norm()
    assert torch.equal(ta, tb) or lg < 3e-2, f"decode on the stored codes drifts from the bf16 decode on de-quantised weights: logits rel-L2 {lg:.3g}"
    with pytest.raises(NotImplementedError):
        qm.language_model.engine().refresh()


@pytest.mark.parametrize("kind", ["nf4", "int8_rowwise"])
def test_whole_model_quantised_load_mode_at_7b_width(kind, golden):
    """The same contract at BAGEL-7B-MoT WIDTH (hidden 3584, intermediate 18944, 28 / 4 heads of 128, 2 MoT layers): the row lengths, the 64-weight
    NF4 blocks per row (56 / 296) and the scratch-set rotation of ``_StoredLayers`` are the real model's.  Request = the 512^2 text->image of tests/golden/wide7b_options.pt
    (4 timesteps, CFG 4.0) + 6 greedy tokens."""
    from oracle.configs import WIDE7B as cfg, NEW_TOKEN_IDS_TINY
    from tests.test_model_gpu import cfg_kwargs, new_cache
    from tests.test_wide_gpu import _options_context, _wide_model
    g = golden("wide7b_options")
    qm, bm = _wide_model(), _wide_model()
    o = ops()
    for L in bm.language_model.model.layers:                      # W <- dequantise(quantise(W)) on every decoder projection
        mods = [getattr(L.self_attn, n + s) for n in ("q_proj", "k_proj", "v_proj", "o_proj") for s in ("", "_moe_gen")]
        for s in ("", "_moe_gen"):
            m = getattr(L, "mlp" + s)
            mods += [m.gate_proj, m.up_proj, m.down_proj]
        for m in mods:
            w = m.weight.data
            m.weight.data = (o.dequantize_nf4(*o.quantize_nf4(w)) if kind == "nf4" else o.dequantize_rows_i8(*o.quantize_rows_i8(w))).contiguous()
    bm.language_model.invalidate_packed()
    resident = qm.quantize_language_model(kind)
    full = sum(p.numel() * 2 for n, p in bm.language_model.model.layers.named_parameters() if "proj" in n and n.endswith("weight"))
    assert resident < (0.30 if kind == "nf4" else 0.52) * full
    with pytest.raises(RuntimeError, match="release_bf16"):
        qm.state_dict()                                            # (no 0-element tensors in a saved state dict)
    kw, c = g["fp8"]["gen_kwargs"], g["cfg_inputs"]
    outs = []
    for model in (qm, bm):
        cache = _options_context(model, cfg, g)
        kv = [(cache.key_cache[i].clone(), cache.value_cache[i].clone()) for i in range(cfg["llm"]["num_hidden_layers"])]
        lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), c), **kw, **g["latent_inputs"])
        lens, ropes = [cache.seq_lens], [cache.seq_lens]
        st = model.prepare_start_tokens(lens, ropes, NEW_TOKEN_IDS_TINY)
        toks = model.generate_text(past_key_values=cache, max_length=6, do_sample=False, end_token_id=None, **st)
        outs.append((kv, lat, toks, model._last_decode_session))
    (ka, la, ta, sa), (kb, lb, tb, sb) = outs
    for (k1, v1), (k2, v2) in zip(ka, kb):
        assert torch.equal(k1, k2) and torch.equal(v1, v2), "prefill KV differs"
    for a, b in zip(la, lb):
        assert torch.isfinite(a).all() and torch.equal(a, b), "latents differ"
    assert sa.weight_quant == kind and sb.weight_quant is None
    lg = (sa.logits.float() - sb.logits.float()).norm() / sb.logits.float().norm()
    assert torch.equal(ta, tb) or lg < 3e-2, f"decode on the stored codes drifts from the bf16 decode on de-quantised weights: logits rel-L2 {lg:.3g}"
    print(f"whole-model {kind} at 7B width: resident {resident / 1e6:.0f} MB of {full / 1e6:.0f} MB, prefill KV and latents bit-identical, decode logits rel-L2 {lg:.2e}")
