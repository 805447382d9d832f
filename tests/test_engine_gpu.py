"""GPU parity of the persistent decode engine (csrc/engine.hip, bagel_decode_engine_bf16) through the C ABI.

The engine replaces a CHAIN of batch-1 projections -- o_proj(+residual) -> RMSNorm + gate/up (SwiGLU) -> down(+residual) -> RMSNorm + the
next layer's qkv (bagel.py:930-1000 at Lq = 1; qwen2_navit.py:591-594,515-517; modeling_qwen2.py:200-201) -- by ONE launch whose
workgroups hand the activation vectors to each other inside the launch.  Its contract is BIT-IDENTITY with the chain of
``bagel_gemv_bf16`` launches it replaces (same chunk -> lane map, same accumulation order, same roundings), so every comparison here is
``torch.equal`` against that chain -- whose own parity against fp32 torch / the oracle lives in tests/test_ops_gpu.py and
tests/test_decode_gpu.py -- at the 7B decode shapes, at ragged / tiny geometries (fewer units than workgroups, a last group of fewer than
64 chunks, split-K quarters of different length), under hipGraph replay with the flag words re-cleared per replay, and end to end through
``generate_text`` (engine on vs off: same tokens, same logits)."""
import pytest
import torch

from tests.test_ops_gpu import BF16, DEV, ops, rnd

pytestmark = pytest.mark.gpu


def make_chain(H, I, NQKV, *, seed=0, last="qkv", vocab=None):
    """Weights and buffers of one decoder layer's chain at hidden size H, MLP width I, fused q|k|v width NQKV."""
    d = lambda t: t.to(DEV)  # noqa: E731
    w = dict(wo=d(rnd(H, H, seed=seed + 1, scale=H ** -0.5)), wgu=d(rnd(2 * I, H, seed=seed + 2, scale=H ** -0.5)),
             wd=d(rnd(H, I, seed=seed + 3, scale=I ** -0.5)),
             ln_post=d((1.0 + 0.1 * rnd(H, seed=seed + 5).float()).to(BF16)), ln_next=d((1.0 + 0.1 * rnd(H, seed=seed + 6).float()).to(BF16)))
    if last == "qkv":
        w["wlast"], w["blast"] = d(rnd(NQKV, H, seed=seed + 4, scale=H ** -0.5)), d(rnd(NQKV, seed=seed + 7, scale=0.1))
    else:
        w["wlast"], w["blast"] = d(rnd(vocab, H, seed=seed + 4, scale=H ** -0.5)), None
    return w


def chain_phases(w, att, x, act, out):
    return [dict(A=att, W=w["wo"], C=x, residual=x),
            dict(A=x, W=w["wgu"], C=act, norm_w=w["ln_post"], epilogue=ops().EPI_SWIGLU16),
            dict(A=act, W=w["wd"], C=x, residual=x),
            dict(A=x, W=w["wlast"], C=out, norm_w=w["ln_next"], bias=w["blast"])]


def run_launch_form(phases, eps):
    for ph in phases:
        ops().gemv(ph["A"].view(1, -1), ph["W"], ph["C"].view(1, -1), bias=ph.get("bias"),
                   residual=None if ph.get("residual") is None else ph["residual"].view(1, -1), epilogue=ph.get("epilogue", 0),
                   norm_w=ph.get("norm_w"), eps=eps)


def buffers(H, I, Nlast, seed):
    att = rnd(H, seed=seed + 11).to(DEV)
    x0 = rnd(H, seed=seed + 12).to(DEV)
    nan = lambda n: torch.full((n,), float("nan"), dtype=BF16, device=DEV)  # noqa: E731
    return att, x0, nan(I), nan(Nlast)


def both_forms(H, I, NQKV, *, seed=0, last="qkv", vocab=None, nph=4):
    w = make_chain(H, I, NQKV, seed=seed, last=last, vocab=vocab)
    Nlast = w["wlast"].shape[0]
    eps = 1e-6
    att, x0, act_a, out_a = buffers(H, I, Nlast, seed)
    xa = x0.clone()
    run_launch_form(chain_phases(w, att, xa, act_a, out_a)[:nph], eps)
    _, _, act_b, out_b = buffers(H, I, Nlast, seed)
    xb = x0.clone()
    phases = chain_phases(w, att, xb, act_b, out_b)[:nph]
    assert ops().decode_engine_supported(phases)
    sync = torch.zeros(ops().decode_engine_sync_words(nph), dtype=torch.int32, device=DEV)
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    ops().decode_engine(phases, eps, sync, status)
    torch.cuda.synchronize()
    assert int(status[0]) == 0, f"engine gave up a bounded wait: code 0x{int(status[0]) & 0xff:x} workgroup {int(status[0]) >> 8}"
    return (xa, act_a, out_a), (xb, act_b, out_b)


def assert_identical(a, b, what, nph=4):
    names = ["x (residual stream)", "act (SwiGLU output)", "last projection"]
    live = [True, nph >= 2, nph >= 4]
    for ta, tb, n, on in zip(a, b, names, live):
        if not on:
            continue
        assert torch.isfinite(ta.float()).all(), f"{what}: launch form left non-finite values in {n}"
        same = torch.equal(ta, tb)
        if not same:
            d = (ta.float() - tb.float()).abs()
            raise AssertionError(f"{what}: {n} differs from the gemv chain: {int((d > 0).sum())} of {d.numel()} elements, max |d| = {d.max().item():.4g}, "
                                 f"non-finite in engine output: {int((~torch.isfinite(tb.float())).sum())}")


def test_engine_7b_layer_chain_bit_identical():
    """o 3584x3584 (+x) -> gate/up 37888x3584 (norm, SwiGLU16) -> down 3584x18944 (+x, four K quarters of 10/10/10/7 groups) -> qkv 4608x3584 (norm,
    bias): 7 + 74 + 28 + 9 units per workgroup on 256 CUs."""
    a, b = both_forms(3584, 18944, 4608, seed=3)
    assert_identical(a, b, "7B layer chain")


def test_engine_7b_last_layer_lm_head():
    """The last layer's chain ends in the final norm + lm_head (152064 x 3584: 297 row pairs per workgroup)."""
    a, b = both_forms(3584, 18944, 0, seed=5, last="lm_head", vocab=152064)
    assert_identical(a, b, "7B last layer + lm_head")


@pytest.mark.parametrize("nph", [1, 2, 3])
def test_engine_shorter_chains(nph):
    a, b = both_forms(3584, 18944, 4608, seed=7, nph=nph)
    assert_identical(a, b, f"{nph}-phase chain", nph=nph)


@pytest.mark.parametrize("H,I,NQKV", [(128, 256, 192), (64, 8192, 96), (1096, 1008, 40), (520, 9216, 1024), (4096, 11008, 6144), (2048, 8208, 2560)])
def test_engine_geometry_edges(H, I, NQKV):
    """Fewer units than workgroups (most workgroups only hand over), rows whose last 64-chunk group is ragged (K = 1096, 520, 1008, 8208), split-K
    rows whose quarters differ in length (K = 8208: 17 groups -> 5/5/5/2; 9216: 18 -> 5/5/5/3; 11008: 22 -> 6/6/6/4), K < one group."""
    a, b = both_forms(H, I, NQKV, seed=H + I)
    assert_identical(a, b, f"chain H={H} I={I} NQKV={NQKV}")


def test_engine_graph_replay_and_fresh_flags():
    """30 replays of [clear flags, engine launch] from one hipGraph with a new activation vector each time: every replay must equal the launch form
    (stale flag words or stale activations from the previous replay would show here), status stays 0."""
    H, I, NQKV = 3584, 18944, 4608
    w = make_chain(H, I, NQKV, seed=21)
    eps = 1e-6
    att, x0, act, out = buffers(H, I, NQKV, 21)
    x = x0.clone()
    phases = chain_phases(w, att, x, act, out)
    sync = torch.zeros(ops().decode_engine_sync_words(4), dtype=torch.int32, device=DEV)
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with ops().HipGraph.capture(side) as g:
        sync.zero_()
        ops().decode_engine(phases, eps, sync, status)
    for it in range(30):
        a_in, x_in = rnd(H, seed=100 + it).to(DEV), rnd(H, seed=200 + it).to(DEV)
        xr, actr, outr = x_in.clone(), torch.empty_like(act), torch.empty_like(out)
        run_launch_form(chain_phases(w, a_in, xr, actr, outr), eps)
        att.copy_(a_in); x.copy_(x_in)
        act.fill_(float("nan")); out.fill_(float("nan"))
        torch.cuda.synchronize()
        g.launch()
        side.synchronize()
        assert int(status[0]) == 0
        assert_identical((xr, actr, outr), (x, act, out), f"replay {it}")


def test_engine_refuses_what_it_cannot_serve():
    H = 128
    w = make_chain(H, 256, 192)
    att, x, act, out = buffers(H, 256, 192, 0)
    sync = torch.zeros(ops().decode_engine_sync_words(4), dtype=torch.int32, device=DEV)
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    ph = chain_phases(w, att, x, act, out)
    with pytest.raises(ops().BagelHipError):
        ops().decode_engine([ph[0], ph[2]], 1e-6, sync, status)               # phase 1 does not read phase 0's output
    with pytest.raises(ops().BagelHipError):
        ops().decode_engine(ph, 1e-6, sync[:8], status)                          # flag words too few
    big = dict(A=torch.zeros(8192, dtype=BF16, device=DEV), W=torch.zeros(64, 8192, dtype=BF16, device=DEV), C=torch.zeros(64, dtype=BF16, device=DEV),
               norm_w=torch.ones(8192, dtype=BF16, device=DEV))
    assert not ops().decode_engine_supported([big])                              # fused RMSNorm over K > 4096
    torch.cuda.synchronize()
    assert int(status[0]) == 0


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_generate_text_engine_on_equals_off(monkeypatch, name):
    """End to end on the tiny models: the same prefill, then 12 greedy tokens with the engine (3 launches per layer) and with the launch form (6 per
    layer), hipGraph replay and eager: identical token ids, identical last-step logits, identical K/V rows written back."""
    import copy
    from oracle.configs import TINY, TINY_D128
    from tests.test_decode_gpu import _context
    from tests.util_models import product_model
    cfg = {"tiny": TINY, "tiny_d128": TINY_D128}[name]
    model, _ = product_model(cfg)
    cache, lens, ropes, start = _context(model, cfg, ["a small red cube"])
    outs = []
    for flag, graph in (("1", True), ("1", False), ("0", True)):
        monkeypatch.setenv("BAGEL_DECODE_ENGINE", flag)
        c = copy.deepcopy(cache)
        toks = model.generate_text(past_key_values=c, max_length=12, end_token_id=None, use_graph=graph, **start)
        sess = model._last_decode_session
        assert sess.engine_mode == (flag == "1"), "the engine form was not selected / not switched off"
        if graph:
            assert sess.graph is not None, f"hipGraph capture failed: {sess.graph_error}"
        L = cfg["llm"]["num_hidden_layers"]
        outs.append((toks.clone(), sess.logits.clone(), [c.key_cache[i].clone() for i in range(L)], [c.value_cache[i].clone() for i in range(L)]))
    ref = outs[-1]
    for got, what in zip(outs[:-1], ("engine + graph", "engine eager")):
        assert torch.equal(got[0], ref[0]), f"{what}: different tokens than the launch form"
        assert torch.equal(got[1], ref[1]), f"{what}: different last-step logits"
        for a, b in zip(got[2] + got[3], ref[2] + ref[3]):
            assert torch.equal(a, b), f"{what}: different K/V rows written back"
