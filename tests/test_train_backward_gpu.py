"""GPU parity of the training backward (``loss.backward()`` over Bagel.forward; train/pretrain_unified_navit.py:683-735), through the C ABI:

(a) every reverse kernel of csrc/backward.hip against fp64 torch on the CPU (the stand-ins of tests/mock_ops.py state the contract of
    include/bagel_hip.h in torch; they are the reference here, never the product);
(b) the block-masked attention reverse (csrc/attention_bwd.hip) against torch AUTOGRAD of dense masked attention (fp32 softmax, the
    reference's own per-sample masks, data/data_utils.py:72-103) -- an independent statement of the same gradients -- on causal / full /
    noise split structures with splits longer than one 128-row work item, GQA 28 / 4 and 4 / 2, head_dim 128 and 64;
(c) the whole step: every trainable parameter's ``.grad`` after ``loss.backward()`` on the tiny models and on a 7B-WIDTH model (hidden
    3584, 28 / 4 heads of 128, intermediate 18 944, 2 MoT layers) against the oracle's autograd, which is pinned bit for bit to the
    unmodified reference's backward (oracle/make_golden_train_grads.py, tests/test_reference_crosscheck.py);
(d) determinism: the same batch twice gives bit-identical gradients (no atomics anywhere in the reverse).

Tolerances (rel-L2 per tensor): kernels 4e-3 (one bf16 rounding of an fp32 result; 1e-2 where two bf16 roundings chain); attention
reverse 1e-2 (P and dS are rounded to bf16 before their products, as flash-attn's backward does); parameter gradients 4e-2 on the fixed
batches.  That bound is set against the REFERENCE'S OWN noise: tools/train_grad_noise_floor.py re-runs the oracle's autograd with
fp32-accumulating linears (same bf16 operands and rounding points, another summation order -- what another CPU backend or a GPU does) and
the reference's gradients move by up to 1.9e-2 at 7B width, 8.7e-3 on tiny_d128, 2.1e-3 on tiny (profiles/r03_train_grad_noise_floor.log);
the product adds flash-attn's bf16 P / dS and its own rounding points on top: measured 2.8e-2 (tiny), 1.7e-2 (tiny_d128), 2.4e-2 (7B width),
2.9e-2 (SigLIP width).  Round 4 (the last review called 4e-2 loose for 107 tensors): per tensor <= 3.5e-2 (measured worst 2.4e-2 - 2.9e-2: the tail is
1-D norm weights / biases) AND the median over the tensors <= 1.6e-2 (measured 1.1e-2 - 1.25e-2; a wrong kernel moves the bulk, not the tail); 6e-2 for
the text-only / RoPE-variant packs, 8e-2 in the CPU fuzz."""
import pytest
import torch

from tests import mock_ops as M
from tests.test_ops_gpu import rnd

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"


def ops():
    from bagel_amd import ops as o
    return o


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def i32(x):
    return torch.tensor(x, dtype=torch.int32)


# ------------------------------------------------------------------------------------------------------------
# (a) kernels
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols,gather", [(64, 64, False), (130, 72, True), (1000, 3584, False), (1, 8, False), (517, 4608, True)])
def test_transpose(rows, cols, gather):
    o = ops()
    src = rnd(rows + 7, cols, seed=1)
    idx = torch.randperm(rows + 7, generator=torch.Generator().manual_seed(2))[:rows].to(torch.int32) if gather else None
    got = o.transpose(src.to(DEV), rows=idx.to(DEV) if gather else None, n=rows)
    want = M.transpose(src, rows=idx, n=rows)
    assert got.shape == want.shape and torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("rows,cols,two,acc", [(70, 64, True, True), (257, 512, False, False), (300, 3584, True, True), (5, 1152, True, False)])
def test_rmsnorm_bwd(rows, cols, two, acc):
    o = ops()
    x, dy, g = rnd(rows, cols, seed=1), rnd(rows, cols, seed=2), rnd(rows, cols, seed=3)
    w0, w1 = (1 + 0.1 * rnd(cols, seed=4).float()).to(BF16), (1 + 0.1 * rnd(cols, seed=5).float()).to(BF16) if two else None
    ex = (torch.rand(rows, generator=torch.Generator().manual_seed(6)) < 0.4).to(torch.int32) if two else None
    gd = g.to(DEV).clone()
    dw0, dw1 = o.rmsnorm_bwd(x.to(DEV), dy.to(DEV), w0.to(DEV), gd, 1e-6, w1=w1.to(DEV) if two else None, expert=ex.to(DEV) if two else None,
                             accumulate=acc)
    gr = g.clone()
    r0, r1 = M.rmsnorm_bwd(x, dy, w0, gr, 1e-6, w1=w1, expert=ex, accumulate=acc)
    assert rel(gd, gr) < 4e-3
    assert rel(dw0, r0) < 4e-3
    if two:
        assert rel(dw1, r1) < 4e-3
    else:
        assert dw1 is None


@pytest.mark.parametrize("rows,cols,acc", [(70, 64, True), (300, 1152, True), (5, 512, False), (1000, 3584, False)])
def test_layernorm_bwd(rows, cols, acc):
    o = ops()
    x, dy, g = rnd(rows, cols, seed=1, scale=2.0), rnd(rows, cols, seed=2), rnd(rows, cols, seed=3)
    x = (x.float() + 0.7).to(BF16)                                   # a non-zero mean
    w = (1 + 0.1 * rnd(cols, seed=4).float()).to(BF16)
    gd = g.to(DEV).clone()
    dw, db = o.layernorm_bwd(x.to(DEV), dy.to(DEV), w.to(DEV), gd, 1e-6, accumulate=acc)
    gr = g.clone()
    rw, rb = M.layernorm_bwd(x, dy, w, gr, 1e-6, accumulate=acc)
    assert rel(gd, gr) < 4e-3 and rel(dw, rw) < 4e-3 and rel(db, rb) < 4e-3


@pytest.mark.parametrize("nq,nkv,hd,dp,use_norm,two", [(28, 4, 128, 128, True, True), (4, 2, 64, 64, True, False), (4, 2, 72, 128, True, True),
                                                        (8, 8, 32, 64, False, False)])
def test_qknorm_rope_bwd(nq, nkv, hd, dp, use_norm, two):
    o = ops()
    rows = 150
    W = (nq + 2 * nkv) * dp
    d, raw = rnd(rows, W, seed=1), rnd(rows, W, seed=2)
    pos = torch.randint(0, 900, (rows,), generator=torch.Generator().manual_seed(3))
    inv = 1.0 / (1e6 ** (torch.arange(0, hd, 2).float() / hd))
    cos, sin = M.rope_table(pos, inv)
    ws = [(1 + 0.1 * rnd(hd, seed=10 + i).float()).to(BF16) for i in range(4)]
    ex = (torch.rand(rows, generator=torch.Generator().manual_seed(6)) < 0.5).to(torch.int32)
    t = lambda x: None if x is None else x.to(DEV)  # noqa: E731
    args = (ws[0] if use_norm else None, ws[1] if use_norm else None, ws[2] if (use_norm and two) else None, ws[3] if (use_norm and two) else None,
            ex if two else None)
    dd = d.to(DEV).clone()
    got = o.qknorm_rope_bwd(dd, raw.to(DEV), cos.to(DEV), sin.to(DEV), *[t(a) for a in args], nq, nkv, hd, dp, 1e-6, use_norm)
    dr = d.clone()
    want = M.qknorm_rope_bwd(dr, raw, cos, sin, *args, nq, nkv, hd, dp, 1e-6, use_norm)
    assert torch.equal(dd[:, (nq + nkv) * dp:].cpu(), d[:, (nq + nkv) * dp:]), "v columns must stay untouched"
    assert rel(dd, dr) < 1e-2
    for a, b in zip(got, want):
        assert (a is None) == (b is None)
        if a is not None:
            assert rel(a, b) < 1e-2


def test_elementwise_reverse_kernels():
    o = ops()
    rows, I = 77, 18944 // 8
    gu, da = rnd(rows, 2 * I, seed=1, scale=2.0), rnd(rows, I, seed=2)
    got = o.swiglu_bwd(gu.to(DEV).clone(), da.to(DEV))
    assert rel(got, M.swiglu_bwd(gu.clone(), da)) < 4e-3
    for kind in (1, 2):
        pre, d = rnd(rows, 1152, seed=3, scale=2.0), rnd(rows, 1152, seed=4)
        assert rel(o.act_bwd(pre.to(DEV).clone(), d.to(DEV), kind), M.act_bwd(pre.clone(), d, kind)) < 4e-3
    for cols in (1000, 152064):
        logits = rnd(6, cols, seed=5, scale=3.0)
        labels = torch.tensor([0, cols - 1, 17, 500, -100, 64])
        dl = torch.rand(6, generator=torch.Generator().manual_seed(7))
        got = o.cross_entropy_bwd(logits.to(DEV).clone(), labels.to(DEV), dl.to(DEV))
        want = M.cross_entropy_bwd(logits.clone(), labels, dl)
        assert rel(got, want) < 4e-3 and float(got[4].abs().max()) == 0.0
    n, cols = 37, 64
    clean, noise = rnd(n, cols, seed=1, dtype=torch.float32), rnd(n, cols, seed=2, dtype=torch.float32)
    src, pred, dl = i32([0, 2, 9, 36]), rnd(4, cols, seed=7), torch.rand(4, cols, generator=torch.Generator().manual_seed(8))
    got = o.mse_rows_bwd(pred.to(DEV), noise.to(DEV), clean.to(DEV), src.to(DEV), dl.to(DEV))
    assert rel(got, M.mse_rows_bwd(pred, noise, clean, src, dl)) < 4e-3


def test_swiglu_kernel_equals_the_fused_epilogue():
    """gemm (plain) + swiglu_fwd == gemm with the SwiGLU16 epilogue, bit for bit (the tape's KEEP_GATE_UP path)."""
    o = ops()
    Mr, K, I = 300, 512, 1024
    A, W = rnd(Mr, K, seed=1), rnd(2 * I, K, seed=2, scale=K ** -0.5)
    fused = torch.empty(Mr, I, dtype=BF16, device=DEV)
    o.gemm(A.to(DEV), W.to(DEV), fused, epilogue=o.EPI_SWIGLU16)
    gu = torch.empty(Mr, 2 * I, dtype=BF16, device=DEV)
    o.gemm(A.to(DEV), W.to(DEV), gu)
    act = o.swiglu_fwd(gu, torch.empty_like(fused))
    assert torch.equal(act.view(torch.int16), fused.view(torch.int16))


def test_segment_sum_and_colsum():
    o = ops()
    src = rnd(500, 256, seed=1)
    ids = torch.randint(0, 40, (300,), generator=torch.Generator().manual_seed(2))
    rows = torch.randperm(500, generator=torch.Generator().manual_seed(3))[:300]
    order = torch.argsort(ids, stable=True)
    uniq, counts = torch.unique_consecutive(ids[order], return_counts=True)
    seg = torch.zeros(uniq.numel() + 1, dtype=torch.int32)
    seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
    dst = torch.zeros(64, 256, dtype=BF16)
    got = o.rows_segment_sum(src.to(DEV), rows[order].to(torch.int32).to(DEV), seg.to(DEV), uniq.to(torch.int32).to(DEV), dst.to(DEV))
    want = M.rows_segment_sum(src, rows[order].to(torch.int32), seg, uniq.to(torch.int32), dst.clone())
    assert rel(got, want) < 4e-3
    untouched = torch.ones(64, dtype=torch.bool); untouched[uniq] = False
    assert float(got[untouched.to(DEV)].abs().max()) == 0.0
    for n, cols, gather in ((500, 256, False), (130, 4608, True), (1, 64, False)):
        s = rnd(max(n, 200), cols, seed=4)
        idx = torch.randperm(s.shape[0], generator=torch.Generator().manual_seed(5))[:n].to(torch.int32) if gather else None
        got = o.colsum(s.to(DEV), idx.to(DEV) if gather else None, n)
        assert rel(got, M.colsum(s, idx, n)) < 4e-3


def test_weight_and_input_gradient_products_on_the_forward_gemm():
    """dW = dY^T X per expert (row-gathered transposes) and dX = dY W through the MoT routing, on bagel_gemm_bf16."""
    from bagel_amd.modeling.bagel import train_step as TS
    o = ops()
    Mr, N, K = 700, 512, 384
    dY, X = rnd(Mr, N, seed=1), rnd(Mr, K, seed=2)
    W0, W1 = rnd(N, K, seed=3, scale=K ** -0.5), rnd(N, K, seed=4, scale=K ** -0.5)
    perm = torch.randperm(Mr, generator=torch.Generator().manual_seed(5))
    r0, r1 = perm[:450].sort().values.to(torch.int32), perm[450:].sort().values.to(torch.int32)
    for rows in (r0, r1):
        dW = TS._wgrad(dY.to(DEV), X.to(DEV), rows.to(DEV), rows.numel())
        want = dY[rows.long()].double().t() @ X[rows.long()].double()
        assert rel(dW, want) < 4e-3
    dX = torch.empty(Mr, K, dtype=BF16, device=DEV)
    o.gemm(dY.to(DEV), TS._wt(W0.to(DEV)), dX, a_rows0=r0.to(DEV), c_rows0=r0.to(DEV), M0=r0.numel(), W1=TS._wt(W1.to(DEV)), a_rows1=r1.to(DEV),
           c_rows1=r1.to(DEV), M1=r1.numel())
    want = torch.empty(Mr, K, dtype=torch.float64)
    want[r0.long()] = dY[r0.long()].double() @ W0.double()
    want[r1.long()] = dY[r1.long()].double() @ W1.double()
    assert rel(dX, want) < 4e-3


# ------------------------------------------------------------------------------------------------------------
# (b) attention reverse
# ------------------------------------------------------------------------------------------------------------
SPLITS = {
    "mixed": [([6, 70, 9], ["causal", "full", "causal"]), ([5, 130, 140, 4, 66], ["causal", "full", "noise", "causal", "noise"])],
    "long_causal": [([300], ["causal"]), ([1, 257], ["noise", "causal"])],
    "noise_first": [([64, 64, 200], ["noise", "full", "causal"])],
    "single_rows": [([1], ["causal"]), ([1, 1, 1], ["full", "noise", "causal"])],
}


@pytest.mark.parametrize("nq,nkv,D", [(4, 2, 128), (28, 4, 128), (4, 2, 64)])
@pytest.mark.parametrize("case", list(SPLITS))
def test_attention_backward_block_mask(case, nq, nkv, D):
    from bagel_amd.modeling.bagel.train_step import AttnBackwardPlan
    from oracle import bagel_oracle as O
    o = ops()
    samples = SPLITS[case]
    sample_lens = [sum(s[0]) for s in samples]
    Mr, G, scale = sum(sample_lens), nq // nkv, D ** -0.5
    q, k, v, do = rnd(Mr, nq * D, seed=1), rnd(Mr, nkv * D, seed=2), rnd(Mr, nkv * D, seed=3), rnd(Mr, nq * D, seed=4)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    outs, r0 = [], 0
    for (lens, modes), n in zip(samples, sample_lens):
        mask = O.attention_mask_per_sample(lens, modes)
        qs = qf[r0:r0 + n].view(n, nq, D).transpose(0, 1)
        ks = kf[r0:r0 + n].view(n, nkv, D).repeat_interleave(G, dim=1).transpose(0, 1)
        vs = vf[r0:r0 + n].view(n, nkv, D).repeat_interleave(G, dim=1).transpose(0, 1)
        p = torch.softmax(qs @ ks.transpose(1, 2) * scale + mask[None], dim=-1)
        outs.append((p @ vs).transpose(0, 1).reshape(n, nq * D))
        r0 += n
    out = torch.cat(outs)
    out.backward(do.float())
    bp = AttnBackwardPlan(DEV, sample_lens, samples)
    dq, dk, dv = (torch.full(t.shape, float("nan"), dtype=BF16, device=DEV) for t in (q, k, v))
    o.attn_bwd_blockmask(q.to(DEV), k.to(DEV), v.to(DEV), out.detach().to(BF16).to(DEV), do.to(DEV), dq, dk, dv, bp.q_items, bp.k_items, bp.noise_bits,
                         nq, nkv, D, scale)
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        assert torch.isfinite(got.float()).all(), (case, name)
        assert rel(got, want) < 1e-2, (case, name, rel(got, want))
    # the same with the row statistics handed over by the forward (log2 of the softmax denominators) instead of the first kernel's own pass
    lse, r0 = torch.empty(nq, Mr), 0
    for (lens, modes), n in zip(samples, sample_lens):
        mask = O.attention_mask_per_sample(lens, modes)
        qs = q[r0:r0 + n].float().view(n, nq, D).transpose(0, 1)
        ks = k[r0:r0 + n].float().view(n, nkv, D).repeat_interleave(G, dim=1).transpose(0, 1)
        lse[:, r0:r0 + n] = torch.logsumexp(qs @ ks.transpose(1, 2) * scale + mask[None], dim=-1) * 1.4426950408889634
        r0 += n
    dq2, dk2, dv2 = (torch.full(t.shape, float("nan"), dtype=BF16, device=DEV) for t in (q, k, v))
    o.attn_bwd_blockmask(q.to(DEV), k.to(DEV), v.to(DEV), out.detach().to(BF16).to(DEV), do.to(DEV), dq2, dk2, dv2, bp.q_items, bp.k_items, bp.noise_bits,
                         nq, nkv, D, scale, lse=lse.to(DEV))
    for name, got, want in (("dq", dq2, qf.grad), ("dk", dk2, kf.grad), ("dv", dv2, vf.grad)):
        assert rel(got, want) < 1e-2, (case, name, "lse from the forward", rel(got, want))


# ------------------------------------------------------------------------------------------------------------
# (c), (d) the whole step
# ------------------------------------------------------------------------------------------------------------
FROZEN = ("vit_pos_embed.", "latent_pos_embed.")


def _trainable(model):
    names = []
    for n, p in model.named_parameters():
        p.requires_grad_(not n.startswith(FROZEN))
        if p.requires_grad:
            names.append(n)
    return names


def _step(model, batch, noise, w_ce):
    from oracle import bagel_oracle as O
    for p in model.parameters():
        p.grad = None
    out = model(noise=noise, **batch)
    loss = O.training_step_loss(out, None if w_ce is None else w_ce.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


def _compare(grads, ref, names, tol, what, key_bias_is_zero=True, median_tol=None):
    worst = ("", 0.0)
    every = []
    for n in names:
        rn = float(ref[n].float().norm())
        if key_bias_is_zero and n.startswith("vit_model.") and n.endswith("k_proj.bias"):
            # the exact gradient of SigLIP's key bias is zero (softmax is invariant to a constant added to every key of a row): both sides
            # hold rounding noise -- it has to BE noise next to the query bias of the same layer
            qn = float(ref[n.replace("k_proj", "q_proj")].float().norm())
            assert rn <= 0.1 * qn and (n not in grads or float(grads[n].float().norm()) <= 0.1 * qn), (what, n)
            continue
        if rn == 0.0:
            assert n not in grads or float(grads[n].float().norm()) == 0.0, (what, n)
            continue
        assert n in grads, (what, n, "no gradient")
        assert torch.isfinite(grads[n].float()).all(), (what, n)
        d = rel(grads[n], ref[n])
        every.append(d)
        worst = max(worst, (n, d), key=lambda x: x[1])
        assert d < tol, (what, n, d)
    if every:
        med = sorted(every)[len(every) // 2]
        print(f"{what}: {len(every)} gradient tensors, rel-L2 median {med:.2e}, worst {worst[1]:.2e} ({worst[0]})")
        # the per-tensor bound is set by the noisiest tensors (1-D norm weights and biases that sum few, large terms); the BULK has to sit well
        # below it -- a wrong kernel moves every tensor, not the tail
        assert median_tol is None or med < median_tol, (what, "median", med)
    return worst


@pytest.mark.parametrize("keep_gate_up", [False, True], ids=["recompute_gate_up", "keep_gate_up"])
@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_dense", "tiny_moe"])
def test_training_step_gradients_match_the_oracle(golden, monkeypatch, name, keep_gate_up):
    """(tiny_dense / tiny_moe: forward_train + its reverse for Qwen2DecoderLayer and Qwen2MoEDecoderLayer, qwen2_navit.py:620-646,852-883 -- the oracle's
    training forward of these kinds is bit-exact vs the live reference, tests/test_reference_crosscheck.py.)"""
    from bagel_amd.modeling.bagel import train_step as TS
    from oracle import bagel_oracle as O
    monkeypatch.setattr(TS, "KEEP_GATE_UP", keep_gate_up)
    from oracle.configs import TINY, TINY_D128, TINY_DENSE, TINY_MOE
    from tests.util_models import oracle_weights, product_model
    cfg = {"tiny": TINY, "tiny_d128": TINY_D128, "tiny_dense": TINY_DENSE, "tiny_moe": TINY_MOE}[name]
    g = golden(f"{name}_train")
    batch, noise = g["batch"], g["noise"]
    w_ce = torch.rand(g["ce"].shape[0], generator=torch.Generator().manual_seed(5)) + 0.5
    model, _ = product_model(cfg)
    try:
        names = _trainable(model)
        W, _ = oracle_weights(cfg)
        rloss, rgrads, _ = O.training_step_grads(W, cfg, batch, noise, w_ce, names=set(names))
        loss, grads = _step(model, batch, noise, w_ce)
        assert abs(loss - rloss) < 2e-2 * abs(rloss)
        worst = _compare(grads, rgrads, names, 3.5e-2, name, median_tol=1.6e-2)
        print(f"[{name}] {len(names)} gradients, worst rel-L2 {worst[1]:.2e} at {worst[0]}")
        if name == "tiny":
            _compare(grads, golden("tiny_train_grads")["grads"], names, 3.5e-2, "reference fixture", median_tol=1.6e-2)
        loss2, grads2 = _step(model, batch, noise, w_ce)               # (d) no atomics: bit-identical on a second run
        assert loss2 == loss
        for n in grads:
            assert torch.equal(grads[n], grads2[n]), n
    finally:
        for p in model.parameters():
            p.requires_grad_(False)
            p.grad = None


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
def test_text_only_pack_takes_the_single_expert_path(name):
    """No image in the pack: no gen rows, the engines run without expert routing (one row group per GEMM, plain norms) and only the und
    expert, the embeddings, the norms and lm_head receive gradients -- vs the oracle's primitives under autograd."""
    from oracle import bagel_oracle as O
    from oracle.configs import TINY, TINY_D128
    from tests.util_models import oracle_weights, pack_training_batch, product_model, text_only_training_grads
    cfg = {"tiny": TINY, "tiny_d128": TINY_D128}[name]
    batch, _, _, _ = pack_training_batch(cfg, [[("text", 5, True), ("text", 3, False), ("text", 4, True)], [("text", 150, True)]], 3)
    w_ce = torch.rand(batch["ce_loss_indexes"].numel(), generator=torch.Generator().manual_seed(2)) + 0.5
    W, _ = oracle_weights(cfg)
    rloss, rgrads = text_only_training_grads(W, cfg, batch, w_ce)
    model, _ = product_model(cfg)
    try:
        _trainable(model)
        for p in model.parameters():
            p.grad = None
        out = model(**batch)
        assert out["mse"] is None
        loss = O.training_step_loss(out, w_ce.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss.detach()) - rloss) < 2e-2 * abs(rloss)
        grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        assert set(grads) == set(rgrads), sorted(set(grads) ^ set(rgrads))[:6]
        for n, r in rgrads.items():
            if float(r.float().norm()) > 0:
                assert rel(grads[n], r) < 6e-2, (n, rel(grads[n], r))
    finally:
        for p in model.parameters():
            p.requires_grad_(False)
            p.grad = None


def test_siglip_2d_rope_variant_backward():
    """The tower with 2-D RoPE (config.rope) instead of the learned position table: the rotation's reverse on the q / k gradient heads."""
    from oracle import bagel_oracle as O
    from oracle.configs import TINY_ROPE
    from tests.util_models import oracle_weights, pack_training_batch, product_model
    cfg = TINY_ROPE
    samples = [[("text", 3, True), ("vit", 28, 42), ("text", 4, True)], [("text", 2, False), ("vit", 42, 14), ("vae", 32, 48, True)]]
    batch, noise, _, _ = pack_training_batch(cfg, samples, 21)
    w_ce = torch.rand(batch["ce_loss_indexes"].numel(), generator=torch.Generator().manual_seed(4)) + 0.5
    W, _ = oracle_weights(cfg)
    model, _ = product_model(cfg)
    try:
        names = _trainable(model)
        rloss, rgrads, _ = O.training_step_grads(W, cfg, batch, noise, w_ce, names=set(names))
        loss, grads = _step(model, batch, noise, w_ce)
        assert abs(loss - rloss) < 2e-2 * abs(rloss)
        _compare(grads, rgrads, names, 6e-2, "tiny_rope", key_bias_is_zero=False)
    finally:
        for p in model.parameters():
            p.requires_grad_(False)
            p.grad = None


def test_training_step_gradients_at_7b_width():
    """Hidden 3584, 28 / 4 heads of 128, intermediate 18 944, 2 MoT layers: an understanding sample (prompt + 28 x 42 ViT image + answer
    with CE loss) and a generation sample (prompt + clean 64 x 64 image + two noised images, one of them 320 x 256 = 320 latent tokens,
    longer than one work item) against the oracle's autograd on the host."""
    from oracle import bagel_oracle as O
    from oracle.configs import WIDE7B
    from oracle.shapes import bagel_shapes
    from oracle.weights import synth_state_dict
    from bagel_amd.factory import build_bagel
    from tests.util_models import pack_training_batch
    cfg = WIDE7B
    W = {k: v.to(BF16) for k, v in synth_state_dict(bagel_shapes(cfg), 0).items()}
    H = cfg["llm"]["hidden_size"]
    W["latent_pos_embed.pos_embed"] = O.sincos_2d_table(H, cfg["bagel"]["max_latent_size"]).to(BF16)
    W["vit_pos_embed.pos_embed"] = O.sincos_2d_table(H, cfg["bagel"]["vit_max_num_patch_per_side"]).to(BF16)
    model, _ = build_bagel(cfg, device="cuda", with_vae=False)
    model.load_state_dict(W, strict=True)
    samples = [[("text", 6, False), ("vit", 28, 42), ("text", 9, True)],
               [("text", 5, False), ("vae", 64, 64, False), ("vae", 320, 256, True), ("text", 3, True), ("vae", 64, 48, True)]]
    batch, noise, _, _ = pack_training_batch(cfg, samples, 11)
    w_ce = torch.rand(batch["ce_loss_indexes"].numel(), generator=torch.Generator().manual_seed(5)) + 0.5
    names = _trainable(model)
    rloss, rgrads, _ = O.training_step_grads(W, cfg, batch, noise, w_ce, names=set(names))
    loss, grads = _step(model, batch, noise, w_ce)
    assert abs(loss - rloss) < 2e-2 * abs(rloss)
    worst = _compare(grads, rgrads, names, 3.5e-2, "wide7b", median_tol=1.6e-2)
    print(f"[wide7b] {len(names)} gradients, worst rel-L2 {worst[1]:.2e} at {worst[0]}")


def test_training_step_gradients_at_siglip_width():
    """The SigLIP tower at so400m WIDTH (1152-d, 16 heads of 72 padded to 128 lanes, MLP 4304, patch 14; 2 layers) under a 7B-width 2-layer
    LLM: a 448 x 448 image (1024 patches) in an understanding sample + a small generation sample -- every gradient of the tower (patch and
    position embeddings, LayerNorms, attention with its padded heads, MLP) and of the rest against the oracle's autograd."""
    from oracle import bagel_oracle as O
    from oracle.configs import WIDE7B_UND
    from oracle.shapes import bagel_shapes
    from oracle.weights import synth_state_dict
    from bagel_amd.factory import build_bagel
    from tests.util_models import pack_training_batch
    cfg = dict(WIDE7B_UND, llm=dict(WIDE7B_UND["llm"]))
    W = {k: v.to(BF16) for k, v in synth_state_dict(bagel_shapes(cfg), 0).items()}
    H = cfg["llm"]["hidden_size"]
    W["latent_pos_embed.pos_embed"] = O.sincos_2d_table(H, cfg["bagel"]["max_latent_size"]).to(BF16)
    W["vit_pos_embed.pos_embed"] = O.sincos_2d_table(H, cfg["bagel"]["vit_max_num_patch_per_side"]).to(BF16)
    model, _ = build_bagel(cfg, device="cuda", with_vae=False)
    model.load_state_dict(W, strict=True)
    samples = [[("text", 4, False), ("vit", 448, 448), ("text", 6, True)], [("text", 3, False), ("vae", 64, 64, True)]]
    batch, noise, _, _ = pack_training_batch(cfg, samples, 13)
    w_ce = torch.rand(batch["ce_loss_indexes"].numel(), generator=torch.Generator().manual_seed(5)) + 0.5
    names = _trainable(model)
    assert any(n.startswith("vit_model.") for n in names)
    rloss, rgrads, _ = O.training_step_grads(W, cfg, batch, noise, w_ce, names=set(names))
    loss, grads = _step(model, batch, noise, w_ce)
    assert abs(loss - rloss) < 2e-2 * abs(rloss)
    worst = _compare(grads, rgrads, names, 4.5e-2, "wide7b_und")
    print(f"[wide7b_und] {len(names)} gradients, worst rel-L2 {worst[1]:.2e} at {worst[0]}")
