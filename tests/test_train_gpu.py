"""GPU parity of the training forward (Bagel.forward, bagel.py:101-229) against the reference's golden
losses, plus the kernels it adds: the block mask as per-split sequences of the varlen attention kernel, the noised-latent
mix, the per-image timestep add, the MSE and cross-entropy heads."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_ops_gpu import close, rnd

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"


def ops():
    from bagel_amd import ops as o
    return o


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("nq,nkv,D", [(4, 2, 128), (28, 4, 128), (4, 2, 64)])
def test_block_mask_attention_as_split_sequences(nq, nkv, D):
    """Two packed samples with causal / full / noise splits: the TrainPlan decomposition on bagel_attn_varlen_ranges_bf16
    equals masked SDPA (fp32 softmax) with the reference's per-sample masks."""
    from bagel_amd.modeling.bagel.qwen2_navit import TrainPlan, _ceil_to
    from oracle import bagel_oracle as O
    o = ops()
    samples = [([6, 70, 9], ["causal", "full", "causal"]), ([5, 130, 140, 4, 66], ["causal", "full", "noise", "causal", "noise"])]
    sample_lens = [sum(s[0]) for s in samples]
    M = sum(sample_lens)
    tp = TrainPlan(DEV, sample_lens, samples, torch.zeros(M, dtype=torch.long), list(range(M)), [],
                   (1.0 / (1e6 ** (torch.arange(0, D, 2).float() / D))).to(DEV))
    q, k, v = rnd(M, nq * D, seed=1), rnd(M, nkv * D, seed=2), rnd(M, nkv * D, seed=3)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    kw_ = nkv * D
    vt = torch.zeros((kw_, _ceil_to(tp.vt_cols, 256)), dtype=BF16, device=DEV)
    k_clean = torch.zeros((_ceil_to(tp.n_clean + 64, 64), kw_), dtype=BF16, device=DEV)
    v_clean = torch.zeros_like(k_clean)
    vt_clean = torch.zeros((kw_, _ceil_to(tp.vt_clean_cols, 256)), dtype=BF16, device=DEV)
    o.v_transpose(vd, vt, tp.cu_splits, tp.new_col, tp.n_splits, tp.max_split, nkv, D)
    o.copy_rows(kd, k_clean, tp.n_clean, kw_, src_rows=tp.clean_rows)
    o.copy_rows(vd, v_clean, tp.n_clean, kw_, src_rows=tp.clean_rows)
    o.v_transpose(v_clean, vt_clean, tp.cu_clean, tp.clean_col, tp.n_samples, tp.max_clean, nkv, D)
    out = torch.full((M, nq * D), float("nan"), dtype=BF16, device=DEV)
    lse = torch.full((nq, M), float("nan"), dtype=torch.float32, device=DEV)
    scale = D ** -0.5
    for g in tp.groups:
        o.attn_varlen_ranges(qd, kd, vt, out, g["qs"], g["qe"], g["ncol"], g["n"], g["max_lq"], nq, nkv, D, g["causal"], scale,
                             k_ctx=k_clean, vt_ctx=vt_clean, ctx_start=g["cs"], ctx_end=g["ce"], vt_ctx_col=g["ccol"], lse=lse)
    # reference: masked attention per sample, fp32 softmax, GQA by head repeat
    ref, ref_lse, r0 = [], [], 0
    G = nq // nkv
    for (lens, modes), n in zip(samples, sample_lens):
        mask = O.attention_mask_per_sample(lens, modes)
        qs = q[r0:r0 + n].float().view(n, nq, D).transpose(0, 1)
        ks = k[r0:r0 + n].float().view(n, nkv, D).repeat_interleave(G, dim=1).transpose(0, 1)
        vs = v[r0:r0 + n].float().view(n, nkv, D).repeat_interleave(G, dim=1).transpose(0, 1)
        sc = qs @ ks.transpose(1, 2) * scale + mask[None]
        p = torch.softmax(sc, dim=-1)
        ref.append((p @ vs).transpose(0, 1).reshape(n, nq * D))
        ref_lse.append(torch.logsumexp(sc, dim=-1) * 1.4426950408889634)
        r0 += n
    close(out, torch.cat(ref).to(BF16), ulps=2, what="block-mask attention")
    # the row statistics for the training backward: log2 of the softmax denominator (P is summed after its bf16 rounding: 2^-8 relative)
    d = (lse.cpu() - torch.cat(ref_lse, dim=1)).abs().max().item()
    assert d < 2e-2, f"lse: max |d| {d}"


def test_training_glue_kernels():
    o = ops()
    n, cols = 37, 64
    clean, noise = rnd(n, cols, seed=1, dtype=torch.float32), rnd(n, cols, seed=2, dtype=torch.float32)
    t = torch.rand(n, generator=torch.Generator().manual_seed(3))
    t[5:9] = 0.0
    x = o.flow_mix(clean.to(DEV), noise.to(DEV), t.to(DEV))
    ref = ((1 - t[:, None]) * clean + t[:, None] * noise).to(BF16)
    assert torch.equal(x.cpu().view(torch.int16), ref.view(torch.int16)), "flow_mix must be bit-exact"
    assert torch.equal(x[5:9].cpu(), clean[5:9].to(BF16)), "t = 0 keeps the clean latent"
    # per-image timestep embedding + position add
    H = 128
    seq = rnd(50, H, seed=4)
    rows = torch.tensor([3, 4, 5, 20, 21, 40], dtype=torch.int32)
    temb, tid = rnd(2, H, seed=5), torch.tensor([0, 0, 0, 1, 1, 1], dtype=torch.int32)
    table, pid = rnd(9, H, seed=6), torch.tensor([0, 1, 2, 0, 8, 4], dtype=torch.long)
    got = seq.to(DEV).clone()
    o.flow_add_rows(got, rows.to(DEV), temb.to(DEV), tid.to(DEV), table.to(DEV), pid.to(DEV))
    want = seq.clone()
    want[rows.long()] = (seq[rows.long()] + temb[tid.long()]) + table[pid]
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16)), "flow_add_rows must be bit-exact"
    # MSE head
    src = torch.tensor([0, 2, 9, 36], dtype=torch.int32)
    pred = rnd(4, cols, seed=7)
    mse = o.mse_rows(pred.to(DEV), noise.to(DEV), clean.to(DEV), src.to(DEV))
    assert torch.equal(mse.cpu(), (pred - (noise - clean)[src.long()]) ** 2), "mse_rows must be bit-exact"
    # cross entropy
    logits = rnd(6, 1000, seed=8, scale=3.0)
    labels = torch.tensor([0, 999, 17, 500, 3, 64])
    ce = o.cross_entropy(logits.to(DEV), labels.to(DEV)).cpu()
    ref = F.cross_entropy(logits.float(), labels, reduction="none")
    assert (ce - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("name", ["tiny", "tiny_d128", "tiny_dense", "tiny_moe"])
@pytest.mark.parametrize("mask_api", ["nested", "splits"])
def test_training_forward_matches_reference(golden, name, mask_api):
    """Bagel.forward on the hand-packed two-sample batch vs the reference's per-token losses.  Tolerances: CE rel-L2 <= 2e-2
    (logits carry one bf16 rounding per op over L layers); MSE rel-L2 <= 5e-2 (a squared bf16 prediction error).  tiny_dense / tiny_moe =
    the training forward of the dense and MoE layer kinds (qwen2_navit.py:620-646,852-883; their backward: tests/test_train_backward_gpu.py)."""
    from oracle.configs import TINY, TINY_D128, TINY_DENSE, TINY_MOE
    from tests.util_models import product_model
    cfg = {"tiny": TINY, "tiny_d128": TINY_D128, "tiny_dense": TINY_DENSE, "tiny_moe": TINY_MOE}[name]
    g = golden(f"{name}_train")
    model, _ = product_model(cfg)
    batch = dict(g["batch"])
    if mask_api == "splits":
        batch.pop("nested_attention_masks")
        batch.update(split_lens=g["split_lens"], attn_modes=g["attn_modes"])
    out = model(noise=g["noise"], **batch)
    assert out["mse"].dtype == torch.float32 and out["mse"].shape == g["mse"].shape
    assert out["ce"].dtype == torch.float32 and out["ce"].shape == g["ce"].shape
    assert torch.isfinite(out["mse"]).all() and torch.isfinite(out["ce"]).all()
    assert rel_l2(out["ce"], g["ce"]) <= 2e-2, f"ce rel_l2 {rel_l2(out['ce'], g['ce']):.4g}"
    assert rel_l2(out["mse"], g["mse"]) <= 5e-2, f"mse rel_l2 {rel_l2(out['mse'], g['mse']):.4g}"
    # the mean losses a training step would log
    assert abs(out["ce"].mean().item() - g["ce"].mean().item()) <= 1e-2 * abs(g["ce"].mean().item())
    assert abs(out["mse"].mean().item() - g["mse"].mean().item()) <= 2e-2 * abs(g["mse"].mean().item())
