"""GPU parity of the FP8 (OCP e4m3) option of the gen-expert GEMMs against its CPU restatement (oracle/fp8.py): the row quantiser
bit for bit, the fp8 MFMA GEMM (persistent ping-pong kernel, every epilogue the layer uses, MoT row list) to fp32-accumulation accuracy."""
import pytest
import torch

from oracle import fp8 as F8
from tests.test_ops_gpu import BF16, DEV, close, ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(5, 128), (300, 3584), (64, 18944)])
def test_quantize_rows_fp8_bit_exact(rows, cols):
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, cols, generator=g) * torch.rand(rows, 1, generator=g) * 4).to(BF16)
    x[0, :8] = 0
    if rows > 4:
        x[3] = 0                                              # an all-zero row: scale 1, codes 0
    q, s = ops().quantize_rows_fp8(x.to(DEV))
    qr, sr = F8.quantize_rows_fp8(x)
    assert torch.equal(s.cpu(), sr), (s.cpu() - sr).abs().max()
    same = (q.cpu() == qr)
    # +0 / -0 are distinct codes with the same value
    val_same = q.cpu().view(torch.float8_e4m3fn).float() == qr.view(torch.float8_e4m3fn).float()
    assert bool(val_same.all()), f"{(~val_same).sum().item()} of {val_same.numel()} codes differ ({(~same).sum().item()} bytes)"


@pytest.mark.parametrize("mode", ["plain", "bias", "residual", "swiglu"])
@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (2500, 1024, 3584), (4096 + 77, 256, 18944), (32768, 512, 3584)])
def test_gemm_fp8_matches_dequantised_product(M, N, K, mode):
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M + 6, K, generator=g).to(BF16)            # the row list skips a few rows of the buffer
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF16)
    rows = torch.tensor([i for i in range(M + 6) if i not in (0, 7, 100, 101, 255, 256)][:M], dtype=torch.int32)
    qa, sa = F8.quantize_rows_fp8(A)
    qw, sw = F8.quantize_rows_fp8(W)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF16) if mode == "bias" else None
    Nout = N // 2 if mode == "swiglu" else N
    R = torch.randn(M + 6, Nout, generator=g).to(BF16) if mode == "residual" else None
    C = R.to(DEV).clone() if R is not None else torch.full((M + 6, Nout), float("nan"), dtype=BF16, device=DEV)
    o = ops()
    o.gemm_fp8(qa.to(DEV), sa.to(DEV), qw.to(DEV), sw.to(DEV), C, bias=None if bias is None else bias.to(DEV), rows=rows.to(DEV),
               residual=C if R is not None else None, epilogue=o.EPI_SWIGLU16 if mode == "swiglu" else o.EPI_NONE)
    torch.cuda.synchronize()
    r = rows.long()
    ref = F8.gemm_fp8(qa[r], sa[r], qw, sw, bias=bias, residual=None if R is None else R[r], swiglu=mode == "swiglu")
    close(C.cpu()[r], ref, ulps=2, what=f"gemm_fp8 {mode} M={M} N={N} K={K}")
    untouched = [i for i in range(M + 6) if i not in set(rows.tolist())]
    if R is None:
        assert torch.isnan(C.cpu()[untouched].float()).all(), "rows outside the row list were written"
    # and the quantisation itself costs what e4m3 costs: a few percent against the bf16 product
    if mode == "plain":
        full = (A[r].float() @ W.float().t())
        e = ((C.cpu()[r].float() - full).norm() / full.norm()).item()
        assert e < 6e-2, e


@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (2500, 1024, 3584), (4096 + 77, 37888 // 8, 3584), (32768, 512, 3584)])
def test_gemm_fp8_swiglu_with_fp8_output_and_delayed_scales(M, N, K):
    """bagel_gemm_fp8_swiglu_q8 + bagel_fp8_delayed_scales against the restatement (oracle/fp8.py DelayedScales): given the previous step's row maxima, the scale
    kernel's output bit for bit; the e4m3 codes of the SwiGLU result equal to the restatement's applied to the bf16 SwiGLU result of bagel_gemm_fp8_bf16 (the SAME
    kernel's bf16 output: an exact per-element statement of the epilogue) except where that bf16 value sits on a rounding boundary of the code; the collected row
    maxima exact; rows outside the row list untouched; values past the delayed scale's headroom SATURATE at +-448 (no NaN codes)."""
    g = torch.Generator().manual_seed(M + N + 1)
    A = torch.randn(M + 6, K, generator=g).to(BF16)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF16)
    rows = torch.tensor([i for i in range(M + 6) if i not in (0, 7, 100, 101, 255, 256)][:M], dtype=torch.int32)
    r = rows.long()
    qa, sa = F8.quantize_rows_fp8(A)
    qw, sw = F8.quantize_rows_fp8(W)
    o = ops()
    # the kernel's own bf16 SwiGLU output = what the fp8-output epilogue quantises
    act = torch.zeros((M + 6, N // 2), dtype=BF16, device=DEV)
    o.gemm_fp8(qa.to(DEV), sa.to(DEV), qw.to(DEV), sw.to(DEV), act, rows=rows.to(DEV), epilogue=o.EPI_SWIGLU16)
    act = act.cpu()
    true_amax = act.float().abs().amax(1)
    # "previous step": maxima off by random factors in [0.3, 1.6] -> some rows exceed the 2x headroom (saturation), row 3 of the list has no history (0 -> scale 1)
    prev = true_amax * (0.3 + 1.3 * torch.rand(M + 6, generator=g))
    prev[r[3]] = 0.0
    amax = prev.clone().to(DEV)
    scale = torch.full((M + 6,), -1.0, device=DEV)
    o.fp8_delayed_scales(amax, scale, rows=rows.to(DEV), margin=2.0)
    k = torch.tensor(2.0) / torch.tensor(448.0)
    want_scale = torch.where(prev > 0, prev * k, torch.ones_like(prev))
    assert torch.equal(scale.cpu()[r], want_scale[r]) and (scale.cpu()[[0, 7]] == -1.0).all() and (amax.cpu()[r] == 0).all() and torch.equal(amax.cpu()[[0, 7]], prev[[0, 7]])
    Cq = torch.full((M + 6, N // 2), 0x7f, dtype=torch.uint8, device=DEV)           # 0x7f = an e4m3 NaN code: must disappear from every listed row
    o.gemm_fp8_swiglu_q8(qa.to(DEV), sa.to(DEV), qw.to(DEV), sw.to(DEV), Cq, scale, amax, rows=rows.to(DEV))
    torch.cuda.synchronize()
    got = Cq.cpu()
    inv = 1.0 / want_scale
    y = (act.float() * inv[:, None]).clamp(-448.0, 448.0)
    want = y.to(torch.float8_e4m3fn)
    gv, wv = got.view(torch.float8_e4m3fn).float()[r], want.float()[r]
    assert torch.isfinite(gv).all(), "NaN codes in the fp8 output"
    assert torch.equal(gv, wv), f"{int((gv != wv).sum())} of {gv.numel()} codes differ from the restatement (max |d| {float((gv - wv).abs().max())})"
    assert torch.equal(amax.cpu()[r], true_amax[r]), "collected row maxima"
    assert (got[[0, 7, 100]] == 0x7f).all(), "rows outside the row list were written"
    sat = (act.float()[r].abs() * inv[r][:, None] > 448.0)
    assert sat.any() and (gv[sat].abs() == 448.0).all(), "values past the headroom must saturate"
    # and a second pass of the scale kernel turns the collected maxima into the next scales
    o.fp8_delayed_scales(amax, scale, rows=rows.to(DEV), margin=2.0)
    assert torch.equal(scale.cpu()[r], torch.where(true_amax > 0, true_amax * k, torch.ones_like(true_amax))[r])


@pytest.mark.parametrize("rows,cols", [(7, 128), (257, 3584)])
def test_rmsnorm_fp8_equals_rmsnorm_then_quantise(rows, cols):
    g = torch.Generator().manual_seed(cols)
    x = (torch.randn(rows, cols, generator=g) * 3).to(BF16).to(DEV)
    w = (1 + 0.1 * torch.randn(cols, generator=g)).to(BF16).to(DEV)
    o = ops()
    y = torch.empty_like(x)
    o.rmsnorm(x, w, y, 1e-6)
    q_ref, s_ref = o.quantize_rows_fp8(y)
    q = torch.empty((rows, cols), dtype=torch.uint8, device=DEV)
    s = torch.empty((rows,), dtype=torch.float32, device=DEV)
    o.rmsnorm_fp8(x, w, q, s, 1e-6)
    assert torch.equal(s, s_ref) and torch.equal(q, q_ref)


def _fp8_oracle_latents(cfg, W, g, kw, delayed=False):
    from oracle import bagel_oracle as O
    from oracle import packers as P
    from oracle.configs import NEW_TOKEN_IDS_TINY, StubTokenizer
    L = cfg["llm"]["num_hidden_layers"]
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    prompts = g["prompts"] if "prompts" in g else [g["prompt"]]
    n = len(prompts)
    gi, _, _ = P.prepare_prompts([0] * n, [0] * n, prompts, tok, NEW_TOKEN_IDS_TINY)
    cache = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi)       # the context prefill is und mode: bf16
    ci = g["cfg_inputs"]
    cfgd = dict(cache=O.OracleCache(L), position_ids=ci["cfg_packed_position_ids"], query_indexes=ci["cfg_packed_query_indexes"],
                key_values_lens=ci["cfg_key_values_lens"], key_value_indexes=ci["cfg_packed_key_value_indexes"])
    O.FP8_WEIGHT_PTRS = O.fp8_gen_weight_ptrs(W)
    assert len(O.FP8_WEIGHT_PTRS) == 7 * L
    if delayed:
        from oracle import fp8 as F8o
        O.FP8_DELAYED = F8o.DelayedScales(F8o.down_proj_gen_ptrs(W))
        assert len(O.FP8_DELAYED.ptrs) == L
    try:
        return O.generate_image(W, cfg, g["latent_inputs"], cache, cfg_text=cfgd, **kw)
    finally:
        O.FP8_WEIGHT_PTRS = set()
        O.FP8_DELAYED = None


def test_model_fp8_gen_expert_matches_its_restatement(golden):
    """text->image on the tiny D=128 model with model.gen_weight_quant='fp8' against the oracle with the SAME quantisation scheme
    switched into its gen-expert linears (oracle/fp8.py) -- parity of the option with its own CPU statement -- and, for the record,
    how far the option moves the result from the bf16 reference (~5e-2 on this model).
    Tolerance: the operators are pinned above (quantiser bit for bit, GEMM to fp32-accumulation accuracy); through the sampler the
    bf16-level differences between GPU and CPU activations (~1e-3) flip e4m3 codes next to a rounding boundary (step 2^-3 relative), a
    noise source the bf16 path does not have: measured 4.3e-2 after 4 Euler steps with CFG 4.0, frozen at 8e-2 (2x)."""
    from oracle.configs import TINY_D128 as cfg, NEW_TOKEN_IDS_TINY, StubTokenizer
    from tests.test_model_gpu import cfg_kwargs, new_cache, rel_l2
    from tests.util_models import oracle_weights, product_model
    g = golden("tiny_d128_t2i")
    W, _ = oracle_weights(cfg)
    kw = g["gen_kwargs"]
    ref8 = {False: _fp8_oracle_latents(cfg, W, g, kw), True: _fp8_oracle_latents(cfg, W, g, kw, delayed=True)}
    print("restatement: delayed scales vs exact row scales", [f"{rel_l2(a, b):.3e}" for a, b in zip(ref8[True], ref8[False])])
    model, _ = product_model(cfg)
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(new_cache(cfg), **gi)
    try:
        # both schemes of the SwiGLU output's row scales -- delayed (the default inside a denoise loop, round 6) and exact -- each against ITS restatement
        for delayed in (True, False):
            for batched in (True, False):
                model.gen_weight_quant, model.cfg_batched, model.fp8_delayed_scaling = "fp8", batched, delayed
                lat = model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **kw, **g["latent_inputs"])
                for a, b, c in zip(lat, ref8[delayed], g["latents"]):
                    assert torch.isfinite(a).all()
                    e8, e16 = rel_l2(a, b), rel_l2(a, c)
                    print(f"fp8 gen expert (delayed_scales={delayed}, cfg_batched={batched}): vs its CPU restatement {e8:.3e}; vs the bf16 reference {e16:.3e}")
                    assert e8 <= 8e-2, e8
                    assert 1e-3 < e16 < 0.5, e16            # it IS a different result, and not a wild one
    finally:
        model.gen_weight_quant, model.cfg_batched, model.fp8_delayed_scaling = None, True, True


def test_fp8_single_layer_at_7b_shapes_matches_its_restatement_tightly():
    """ONE MoT decoder layer at BAGEL-7B shapes (hidden 3584, 28 / 4 heads of 128, MLP 18 944), 1 026 query tokens on a 32-token context, gen mode: the UPDATE the layer
    adds to the residual stream (y - x: nothing of the input hides a defect) through the HIP engine with ``gen_quant="fp8"`` against the oracle layer with the same
    scheme in its gen-expert linears -- before any depth, CFG or sampler amplifies the e4m3 code flips, so the gate can be an order of magnitude tighter than the
    end-to-end ones (round-5 verdict, weak 4: a 5 % scheme bug passed those).  Both schemes of the SwiGLU output's scale: exact row scales, and the DELAYED scales on
    the second forward of a stream (state carried over from the first, as inside ``generate_image``).  The bf16 layer on the same inputs is the yardstick."""
    import argparse
    import bench
    from bagel_amd.factory import BAGEL_7B_MOT as cfg, build_bagel
    from bagel_amd.modeling.bagel.qwen2_navit import Fp8DelayedScales, NaiveCache
    from oracle import bagel_oracle as O
    llm = cfg["llm"]
    nkv, hd = llm["num_key_value_heads"], llm["hidden_size"] // llm["num_attention_heads"]
    k = {}
    bench.cpu_port_layer(argparse.Namespace(cpu_layers=1, resolution=512, prompt_tokens=30), cfg, bench.physical_cores(), k)      # the bf16 oracle layer: k["x"]
    W, x0, Lq, C = k["W"], k["x0"], k["Lq"], k["C"]
    cos_sin = O.rope_tables(torch.full((Lq,), C, dtype=torch.long), hd, llm["rope_theta"], torch.bfloat16)
    qlens, kvlens = torch.tensor([Lq], dtype=torch.int), torch.tensor([C], dtype=torch.int)
    layer = lambda x: O.mot_layer(W, llm, 0, x, qlens, cos_sin, k["q_idx"], k["cache"], kvlens, k["kv_idx"], False, False, "gen", k["vae_idx"], k["text_idx"])  # noqa: E731
    # a second input = the first moved a little, as a latent row is between two Euler steps (the delayed scale of step 2 comes from step 1's row maxima)
    x1 = (x0.float() * 1.05 + 0.02 * torch.randn(x0.shape, generator=torch.Generator().manual_seed(5))).to(torch.bfloat16)
    O.FP8_WEIGHT_PTRS = O.fp8_gen_weight_ptrs(W)
    try:
        assert len(O.FP8_WEIGHT_PTRS) == 7
        o8_exact = [layer(x0), layer(x1)]
        O.FP8_DELAYED = F8.DelayedScales(F8.down_proj_gen_ptrs(W))
        O.FP8_DELAYED.begin_step(); layer(x0)
        O.FP8_DELAYED.begin_step(); o8_delayed = layer(x1)
    finally:
        O.FP8_WEIGHT_PTRS, O.FP8_DELAYED = set(), None
    o16 = layer(x1)
    m1, _ = build_bagel(cfg, device=DEV, num_layers=1, with_vae=False)
    m1.load_state_dict(dict(W), strict=False)
    eng = m1.language_model.engine()
    c1 = NaiveCache(1)
    c1.store(0, k["cache"].key_cache[0].reshape(C, nkv * hd).to(DEV), k["cache"].value_cache[0].reshape(C, nkv * hd).to(DEV), [C], [0], nkv, hd, eng.dp)
    plan = eng.plan([Lq], torch.full((Lq,), C, dtype=torch.long), packed_query_indexes=k["q_idx"], key_values_lens=[C], packed_key_value_indexes=k["kv_idx"],
                    text_indexes=k["text_idx"], vae_indexes=k["vae_idx"])
    fwd = lambda x, **kw: eng.forward(x.to(DEV), plan, "gen", c1, update=False, causal=False, num_layers=1, final_norm=False, **kw).float().cpu()  # noqa: E731
    upd = lambda y, x, ry: float(((y - x.float()) - (ry.float() - x.float())).norm() / (ry.float() - x.float()).norm())  # noqa: E731
    e16 = upd(fwd(x1), x1, o16)
    e8 = [upd(fwd(x, gen_quant="fp8"), x, r) for x, r in ((x0, o8_exact[0]), (x1, o8_exact[1]))]
    st = Fp8DelayedScales()
    fwd(x0, gen_quant="fp8", fp8_state=st)
    assert st.primed and float(st.amax[0].max()) > 0
    y8d = fwd(x1, gen_quant="fp8", fp8_state=st)
    e8d = upd(y8d, x1, o8_delayed)
    moved = upd(o8_exact[1], x1, o16)                 # how far the option moves the layer's update from bf16: the scale of what a scheme bug would do
    print(f"7B-shape layer update, product vs restatement: bf16 {e16:.3e}; fp8 exact scales {e8[0]:.3e} / {e8[1]:.3e}; fp8 delayed scales (2nd forward) {e8d:.3e}; "
          f"the option itself moves the update {moved:.3e} from bf16; delayed vs exact restatement {upd(o8_delayed, x1, o8_exact[1]):.3e}")
    # measured on MI355X (round 6): bf16 4.9e-3; fp8 2.68e-2 / 2.63e-2 (exact scales) and 2.67e-2 (delayed) -- e4m3 codes next to a rounding boundary flip on the
    # bf16-level differences between the two sides' activations, the noise floor of ANY two executions of this scheme -- while the option moves the update 6.6e-2 from
    # bf16 and a wrong or missing scale moves it by O(1).  Frozen at 1.5 x the measured floor: 4e-2 (the end-to-end gates of this option sit at 8e-2 / 0.11).
    assert e16 <= 1e-2, e16
    assert max(e8) <= 4e-2 and e8d <= 4e-2, (e8, e8d)
    assert 2e-2 < moved < 0.2
