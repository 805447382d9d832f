"""GPU parity of the autoregressive decode path (Bagel.generate_text, bagel.py:930-1000): the skinny weight-streaming
GEMM, the paged KV cache + Lq=1 split attention, the device-side step bookkeeping, hipGraph replay, and the whole
DecodeSession against the (already golden-pinned) packed prefill engine.

Tolerances: as tests/test_ops_gpu.py (one bf16 op: max|d| <= 2^-7 max|ref|); token ids / bookkeeping / copies bit-exact.
"""
import copy
import math

import pytest
import torch
import torch.nn.functional as F

from tests.test_ops_gpu import close, ref_gemm, rnd

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16
DEV = "cuda"


def ops():
    from bagel_amd import ops as o
    return o


def ref_rmsnorm(x, w, eps):
    xf = x.float()
    return w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(BF16)


# ------------------------------------------------------------------------------------------------------------
# skinny GEMM
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(1, 256, 3584), (1, 64, 18944), (2, 130, 256), (3, 96, 512), (4, 72, 1024), (8, 64, 264),
                                   (5, 34, 8), (1, 512, 64), (2, 48, 18944), (4, 32, 7168), (1, 40, 8200), (3, 24, 18952),
                                   (1, 3584, 18944), (1, 4608, 3584)])
def test_gemv_bias_residual(M, N, K):
    A, W, b, R = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1), rnd(M, N, seed=4)
    C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    ops().gemv(A.to(DEV), W.to(DEV), C, bias=b.to(DEV))
    close(C, ref_gemm(A, W, b), what=f"gemv {M}x{N}x{K}")
    X = R.to(DEV).clone()
    ops().gemv(A.to(DEV), W.to(DEV), X, bias=b.to(DEV), residual=X)        # in place, as the decode layer uses it
    close(X, ref_gemm(A, W, b, 0, R), ulps=2, what=f"gemv residual {M}x{N}x{K}")


@pytest.mark.parametrize("epi", [1, 2])
def test_gemv_activations(epi):
    M, N, K = 3, 136, 320
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    C = torch.empty((M, N), dtype=BF16, device=DEV)
    ops().gemv(A.to(DEV), W.to(DEV), C, bias=b.to(DEV), epilogue=epi)
    close(C, ref_gemm(A, W, b, epi), ulps=2, what=f"gemv epi{epi}")


@pytest.mark.parametrize("M,I,K", [(1, 416, 256), (2, 64, 3584), (4, 32, 512)])
def test_gemv_swiglu_with_fused_rmsnorm(M, I, K):
    from bagel_amd.modeling.bagel.qwen2_navit import interleave_gate_up
    A, Wg, Wu = rnd(M, K, seed=1, scale=3.0), rnd(I, K, seed=2, scale=K ** -0.5), rnd(I, K, seed=3, scale=K ** -0.5)
    w = (1.0 + 0.1 * rnd(K, seed=4).float()).to(BF16)
    Wi = interleave_gate_up(Wg, Wu)
    C = torch.empty((M, I), dtype=BF16, device=DEV)
    ops().gemv(A.to(DEV), Wi.to(DEV), C, epilogue=3)
    ref = F.silu((A.float() @ Wg.float().t()).to(BF16)) * (A.float() @ Wu.float().t()).to(BF16)
    close(C, ref, ulps=2, what="gemv swiglu")
    h = ref_rmsnorm(A, w, 1e-6)
    ops().gemv(A.to(DEV), Wi.to(DEV), C, epilogue=3, norm_w=w.to(DEV), eps=1e-6)
    ref = F.silu((h.float() @ Wg.float().t()).to(BF16)) * (h.float() @ Wu.float().t()).to(BF16)
    close(C, ref, ulps=2, what="gemv rmsnorm+swiglu")


def test_gemv_fused_rmsnorm_matches_two_kernel_path():
    M, N, K = 2, 200, 3584
    A, W, b = rnd(M, K, seed=1, scale=2.0), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    w = (1.0 + 0.1 * rnd(K, seed=4).float()).to(BF16)
    C = torch.empty((M, N), dtype=BF16, device=DEV)
    ops().gemv(A.to(DEV), W.to(DEV), C, bias=b.to(DEV), norm_w=w.to(DEV), eps=1e-6)
    close(C, ref_gemm(ref_rmsnorm(A, w, 1e-6), W, b), what="gemv fused rmsnorm")
    h = torch.empty((M, K), dtype=BF16, device=DEV)
    ops().rmsnorm(A.to(DEV), w.to(DEV), h, 1e-6)
    C2 = torch.empty_like(C)
    ops().gemm(h, W.to(DEV), C2, bias0=b.to(DEV), variant=0)     # the MFMA tile on the normalised rows
    close(C, C2, what="gemv fused rmsnorm vs rmsnorm kernel + MFMA gemm")


@pytest.mark.parametrize("M,N,K", [(2, 64, 64), (5, 48, 3584), (16, 3584, 3584), (17, 208, 512), (34, 4608, 3584), (64, 96, 18944),
                                   (33, 1040, 96), (8, 3584, 18944), (3, 16, 32)])
def test_gemm_skinny_bias_residual(M, N, K):
    A, W, b, R = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1), rnd(M, N, seed=4)
    C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    ops().gemm_skinny(A.to(DEV), W.to(DEV), C, bias=b.to(DEV))
    close(C, ref_gemm(A, W, b), what=f"skinny {M}x{N}x{K}")
    X = R.to(DEV).clone()
    ops().gemm_skinny(A.to(DEV), W.to(DEV), X, bias=b.to(DEV), residual=X)
    close(X, ref_gemm(A, W, b, 0, R), ulps=2, what=f"skinny residual {M}x{N}x{K}")
    # the MFMA tile kernel on the same operands (different summation order only)
    C2 = torch.empty_like(C)
    ops().gemm(A.to(DEV), W.to(DEV), C2, bias0=b.to(DEV), variant=0)
    close(C, C2, what=f"skinny vs 128x128 tile {M}x{N}x{K}")


@pytest.mark.parametrize("epi", [1, 2])
def test_gemm_skinny_activations(epi):
    M, N, K = 20, 144, 320
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    C = torch.empty((M, N), dtype=BF16, device=DEV)
    ops().gemm_skinny(A.to(DEV), W.to(DEV), C, bias=b.to(DEV), epilogue=epi)
    close(C, ref_gemm(A, W, b, epi), ulps=2, what=f"skinny epi{epi}")


@pytest.mark.parametrize("M,I,K", [(2, 416, 256), (34, 18944, 3584), (50, 64, 512)])
def test_gemm_skinny_swiglu(M, I, K):
    from bagel_amd.modeling.bagel.qwen2_navit import interleave_gate_up
    A, Wg, Wu = rnd(M, K, seed=1), rnd(I, K, seed=2, scale=K ** -0.5), rnd(I, K, seed=3, scale=K ** -0.5)
    Wi = interleave_gate_up(Wg, Wu)
    C = torch.full((M, I), float("nan"), dtype=BF16, device=DEV)
    ops().gemm_skinny(A.to(DEV), Wi.to(DEV), C, epilogue=3)
    ref = F.silu((A.float() @ Wg.float().t()).to(BF16)) * (A.float() @ Wu.float().t()).to(BF16)
    close(C, ref, ulps=2, what="skinny swiglu")


def test_gemm_routes_by_row_count():
    """ops.gemm: 1 row -> lane-FMA gemv, 2..64 rows -> skinny MFMA, index lists / odd shapes -> tile kernels; all agree."""
    K, N = 512, 96
    W, b = rnd(N, K, seed=2, scale=K ** -0.5).to(DEV), rnd(N, seed=3, scale=0.1).to(DEV)
    for M in (1, 2, 9, 64, 65):
        A = rnd(M, K, seed=10 + M)
        C = torch.empty((M, N), dtype=BF16, device=DEV)
        ops().gemm(A.to(DEV), W, C, bias0=b)
        close(C, ref_gemm(A, W.cpu(), b.cpu()), what=f"gemm routing M={M}")
    A = rnd(5, 40, seed=1)                       # K % 32 != 0: not the skinny kernel, still correct
    W2 = rnd(24, 40, seed=2, scale=0.2)
    C = torch.empty((5, 24), dtype=BF16, device=DEV)
    ops().gemm(A.to(DEV), W2.to(DEV), C)
    close(C, ref_gemm(A, W2), what="gemm routing K=40")


def test_gemm_routes_few_rows_to_gemv_and_strided_views():
    """ops.gemm with M <= 8 and no index lists takes the skinny kernel; operands may be strided row views."""
    K, N = 512, 96
    buf = rnd(3, 3 * K, seed=1).to(DEV)
    A = buf[:, K:2 * K]                                  # row stride 3K
    W = rnd(N, K, seed=2, scale=K ** -0.5).to(DEV)
    out = torch.zeros((3, 2 * N), dtype=BF16, device=DEV)
    ops().gemm(A, W, out[:, N:])
    close(out[:, N:], ref_gemm(A.cpu(), W.cpu()), what="gemm->gemv strided")
    assert (out[:, :N] == 0).all()


# ------------------------------------------------------------------------------------------------------------
# paged KV cache + Lq = 1 attention
# ------------------------------------------------------------------------------------------------------------
def ref_decode_attention(q, ks, vs, nq, nkv, D, scale):
    """flash-attn definition at Lq=1: fp32 softmax(q k^T scale) v, GQA by head repeat; q:[B,nq*D]."""
    outs = []
    G = nq // nkv
    for b in range(q.shape[0]):
        qb = q[b].float().view(nq, D)
        k = ks[b].float().view(-1, nkv, D).repeat_interleave(G, dim=1)      # [L, nq, D]
        v = vs[b].float().view(-1, nkv, D).repeat_interleave(G, dim=1)
        s = torch.einsum("hd,lhd->hl", qb, k) * scale
        p = torch.softmax(s, dim=-1)
        outs.append(torch.einsum("hl,lhd->hd", p, v).reshape(nq * D))
    return torch.stack(outs).to(BF16)


@pytest.mark.parametrize("lens,nq,nkv,D", [([700, 1, 129], 28, 4, 128), ([127, 128], 4, 2, 64), ([5], 8, 1, 128),
                                            ([300, 64, 65, 1000], 6, 2, 64), ([4932], 28, 4, 128), ([33], 3, 3, 64)])
def test_paged_append_and_decode_attention(lens, nq, nkv, D):
    from bagel_amd.modeling.bagel.decode import PagedKVCache
    o = ops()
    B, width = len(lens), nkv * D
    cap = max(lens) + 3
    g = torch.Generator().manual_seed(7)
    pages = B * ((cap + 63) // 64)
    order = torch.randperm(pages, generator=g).tolist()                  # physically scattered pages
    pg = PagedKVCache(1, B, width, cap, DEV, order=order)
    ks = [rnd(n, width, seed=10 + b) for b, n in enumerate(lens)]
    vs = [rnd(n, width, seed=20 + b) for b, n in enumerate(lens)]

    class FakeCache:       # what PagedKVCache.adopt reads from a NaiveCache: merged rows per layer
        _k = {0: torch.cat([k[:-1] for k in ks]).to(DEV)}
        _v = {0: torch.cat([v[:-1] for v in vs]).to(DEV)}
    if sum(n - 1 for n in lens) > 0:
        pg.adopt(FakeCache, [n - 1 for n in lens])
    else:
        pg.kv_len.zero_()
    # the last token of every sample arrives through the append kernel (slot kv_len[b])
    new = torch.zeros((B, (nq + 2 * nkv) * D), dtype=BF16)
    for b in range(B):
        new[b, nq * D:nq * D + width] = ks[b][-1]
        new[b, nq * D + width:] = vs[b][-1]
    q = rnd(B, nq * D, seed=3)
    new[:, :nq * D] = q
    new = new.to(DEV)
    o.kv_append_paged(new[:, nq * D:nq * D + width], new[:, nq * D + width:], pg.k[0], pg.v[0], pg.block_table, pg.kv_len, B, width)
    for b, n in enumerate(lens):                                          # bit-exact placement
        rows = torch.tensor(pg.physical_rows(b, 0, n), device=DEV)
        assert torch.equal(pg.k[0][rows].cpu(), ks[b]) and torch.equal(pg.v[0][rows].cpu(), vs[b])
    scale = D ** -0.5
    ref = ref_decode_attention(q, ks, vs, nq, nkv, D, scale)
    for max_len in (max(lens), max(lens) + 200):                          # grid larger than the live range is fine
        po, pml = o.attn_decode_workspace(B, nq, D, max_len, DEV)
        out = torch.full((B, nq * D), float("nan"), dtype=BF16, device=DEV)
        o.attn_decode_paged(new, pg.k[0], pg.v[0], pg.block_table, pg.kv_len, 1, max_len, po, pml, out, B, nq, nkv, D, scale)
        close(out, ref, what=f"decode attention lens={lens} max_len={max_len}")
    # len_add = 0 sees only the adopted context
    if min(lens) > 1:
        ref0 = ref_decode_attention(q, [k[:-1] for k in ks], [v[:-1] for v in vs], nq, nkv, D, scale)
        po, pml = o.attn_decode_workspace(B, nq, D, max(lens), DEV)
        out = torch.empty((B, nq * D), dtype=BF16, device=DEV)
        o.attn_decode_paged(new, pg.k[0], pg.v[0], pg.block_table, pg.kv_len, 0, max(lens), po, pml, out, B, nq, nkv, D, scale)
        close(out, ref0, what="decode attention len_add=0")


@pytest.mark.parametrize("hd,dp,nq,nkv,use_norm", [(128, 128, 28, 4, True), (32, 64, 4, 2, True), (64, 64, 3, 1, True), (128, 128, 5, 5, False)])
def test_decode_qkv_post_equals_qknorm_rope_plus_append(hd, dp, nq, nkv, use_norm):
    """The fused decode epilogue is bit-identical to the packed-prefill kernels it replaces (qknorm_rope, und cast points,
    + kv_append), pad lanes of the page rows are zeroed, and physically scattered pages are honoured."""
    from bagel_amd.modeling.bagel.decode import PagedKVCache
    o = ops()
    B = 3
    width = nkv * dp
    ld = (nq + 2 * nkv) * dp
    qkv = torch.zeros((B, nq + 2 * nkv, dp), dtype=BF16)
    qkv[:, :, :hd] = rnd(B, nq + 2 * nkv, hd, seed=1)
    qkv = qkv.view(B, ld).to(DEV)
    qw, kw = (1.0 + 0.1 * rnd(hd, seed=2).float()).to(BF16).to(DEV), (1.0 + 0.1 * rnd(hd, seed=3).float()).to(BF16).to(DEV)
    pos = torch.tensor([5, 77, 4000], dtype=torch.long, device=DEV)
    inv = (1.0 / (1e6 ** (torch.arange(0, hd, 2).float() / hd))).to(DEV)
    cos, sin = o.rope_table(pos, inv)
    lens = [3, 64, 130]
    order = torch.randperm(B * 3, generator=torch.Generator().manual_seed(1)).tolist()
    pg1 = PagedKVCache(1, B, width, 192, DEV, order=order)
    pg2 = PagedKVCache(1, B, width, 192, DEV, order=order)
    for pg in (pg1, pg2):
        pg.k.fill_(float("nan")); pg.v.fill_(float("nan"))
        pg.kv_len.copy_(torch.tensor(lens, dtype=torch.int32))
    a = qkv.clone()
    o.qknorm_rope(a, cos, sin, qw if use_norm else None, kw if use_norm else None, None, None, None, nq, nkv, hd, dp, 1e-6,
                  gen_mode=False, use_norm=use_norm)
    o.kv_append_paged(a[:, nq * dp:nq * dp + width], a[:, nq * dp + width:], pg1.k[0], pg1.v[0], pg1.block_table, pg1.kv_len, B, width)
    b = qkv.clone()
    o.decode_qkv_post(b, cos, sin, qw if use_norm else None, kw if use_norm else None, pg2.k[0], pg2.v[0], pg2.block_table,
                      pg2.kv_len, B, nq, nkv, hd, dp, 1e-6, use_norm)
    assert torch.equal(a[:, :nq * dp].view(torch.int16), b[:, :nq * dp].view(torch.int16)), "q rows differ"
    assert torch.equal(b[:, nq * dp:], qkv[:, nq * dp:]), "k/v of the projection buffer must stay untouched"
    for i, n in enumerate(lens):
        r = pg1.physical_rows(i, n, n + 1)[0]
        assert torch.equal(pg1.k[0][r].view(torch.int16), pg2.k[0][r].view(torch.int16)), "K page row differs"
        assert torch.equal(pg1.v[0][r].view(torch.int16), pg2.v[0][r].view(torch.int16)), "V page row differs"
        assert torch.isfinite(pg2.k[0][r].float()).all() and torch.isfinite(pg2.v[0][r].float()).all()
    untouched = torch.ones(pg2.k.shape[1], dtype=torch.bool)
    untouched[[pg2.physical_rows(i, n, n + 1)[0] for i, n in enumerate(lens)]] = False
    assert torch.isnan(pg2.k[0][untouched.to(DEV)].float()).all(), "only the slot kv_len[b] may be written"


@pytest.mark.parametrize("hd,dp,nq,nkv,use_norm", [(128, 128, 28, 4, True), (32, 64, 4, 2, True), (64, 64, 3, 1, True), (128, 128, 8, 8, False)])
def test_fused_decode_attention_equals_post_plus_attention(hd, dp, nq, nkv, use_norm):
    """bagel_attn_decode_fused_bf16 == decode_qkv_post + attn_decode_paged, bit for bit: attention output and page rows,
    with never-written (NaN) page slots around the live range."""
    from bagel_amd.modeling.bagel.decode import PagedKVCache
    o = ops()
    lens = [3, 129, 500]
    B, width = len(lens), nkv * dp
    qkv = torch.zeros((B, nq + 2 * nkv, dp), dtype=BF16)
    qkv[:, :, :hd] = rnd(B, nq + 2 * nkv, hd, seed=1)
    qkv = qkv.view(B, -1).to(DEV)
    qw, kw = (1.0 + 0.1 * rnd(hd, seed=2).float()).to(BF16).to(DEV), (1.0 + 0.1 * rnd(hd, seed=3).float()).to(BF16).to(DEV)
    pos = torch.tensor([5, 77, 4000], dtype=torch.long, device=DEV)
    inv = (1.0 / (1e6 ** (torch.arange(0, hd, 2).float() / hd))).to(DEV)
    cos, sin = o.rope_table(pos, inv)
    cap = max(lens) + 2
    pages = B * ((cap + 63) // 64)
    order = torch.randperm(pages, generator=torch.Generator().manual_seed(1)).tolist()
    pgs = [PagedKVCache(1, B, width, cap, DEV, order=order) for _ in range(2)]
    ctx_k = [torch.zeros(n, nkv, dp, dtype=BF16) for n in lens]
    ctx_v = [torch.zeros(n, nkv, dp, dtype=BF16) for n in lens]
    for b, n in enumerate(lens):
        ctx_k[b][:, :, :hd] = rnd(n, nkv, hd, seed=10 + b)
        ctx_v[b][:, :, :hd] = rnd(n, nkv, hd, seed=20 + b)

    class Ctx:
        _k = {0: torch.cat([k.view(-1, width) for k in ctx_k]).to(DEV)}
        _v = {0: torch.cat([v.view(-1, width) for v in ctx_v]).to(DEV)}
    for pg in pgs:
        pg.k.fill_(float("nan")); pg.v.fill_(float("nan"))
        pg.adopt(Ctx, lens)
    scale = hd ** -0.5
    max_len = max(lens) + 1
    po, pml = o.attn_decode_workspace(B, nq, dp, max_len, DEV)
    a, out_a = qkv.clone(), torch.full((B, nq * dp), float("nan"), dtype=BF16, device=DEV)
    o.decode_qkv_post(a, cos, sin, qw if use_norm else None, kw if use_norm else None, pgs[0].k[0], pgs[0].v[0], pgs[0].block_table,
                      pgs[0].kv_len, B, nq, nkv, hd, dp, 1e-6, use_norm)
    o.attn_decode_paged(a, pgs[0].k[0], pgs[0].v[0], pgs[0].block_table, pgs[0].kv_len, 1, max_len, po, pml, out_a, B, nq, nkv, dp, scale)
    b_, out_b = qkv.clone(), torch.full((B, nq * dp), float("nan"), dtype=BF16, device=DEV)
    po2, pml2 = o.attn_decode_workspace(B, nq, dp, max_len, DEV)
    o.attn_decode_fused(b_, cos, sin, qw if use_norm else None, kw if use_norm else None, pgs[1].k[0], pgs[1].v[0], pgs[1].block_table,
                        pgs[1].kv_len, max_len, po2, pml2, out_b, B, nq, nkv, hd, dp, 1e-6, use_norm, scale)
    assert torch.isfinite(out_b.float()).all()
    assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16)), "fused attention output differs"
    assert torch.equal(b_, qkv), "the fused kernel must leave the projection buffer untouched"
    for i, n in enumerate(lens):
        r = pgs[0].physical_rows(i, n, n + 1)[0]
        assert torch.equal(pgs[0].k[0][r].view(torch.int16), pgs[1].k[0][r].view(torch.int16)), "K page row differs"
        assert torch.equal(pgs[0].v[0][r].view(torch.int16), pgs[1].v[0][r].view(torch.int16)), "V page row differs"


def test_decode_attention_sharp_softmax():
    """One key dominates by a huge margin in a late split: the split merge must not lose it or overflow."""
    from bagel_amd.modeling.bagel.decode import PagedKVCache
    o = ops()
    nq = nkv = 1
    D, n = 128, 520
    k, v = rnd(n, D, seed=1, scale=0.1), rnd(n, D, seed=2)
    q = rnd(1, D, seed=3)
    k[400] = (q[0].float() * 40).to(BF16)
    pg = PagedKVCache(1, 1, D, n, DEV)

    class FakeCache:
        _k = {0: k.to(DEV)}
        _v = {0: v.to(DEV)}
    pg.adopt(FakeCache, [n])
    po, pml = o.attn_decode_workspace(1, nq, D, n, DEV)
    out = torch.empty((1, D), dtype=BF16, device=DEV)
    o.attn_decode_paged(q.to(DEV), pg.k[0], pg.v[0], pg.block_table, pg.kv_len, 0, n, po, pml, out, 1, nq, nkv, D, 1.0)
    close(out, ref_decode_attention(q, [k], [v], nq, nkv, D, 1.0), what="sharp softmax")


def test_decode_advance_bookkeeping():
    o = ops()
    B, steps = 3, 5
    nxt = torch.tensor([11, 22, 33], dtype=torch.long, device=DEV)
    cur = torch.zeros(B, dtype=torch.int32, device=DEV)
    toks = torch.zeros((steps, B), dtype=torch.long, device=DEV)
    pos = torch.tensor([5, 6, 7], dtype=torch.long, device=DEV)
    kvl = torch.tensor([50, 60, 70], dtype=torch.int32, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    for s in range(steps):       # one call more than rows: the last write is dropped, the counters still advance
        nxt.add_(1)
        o.decode_advance(nxt, cur, toks, pos, kvl, step, B, steps)
    assert step.item() == steps
    assert pos.tolist() == [10, 11, 12] and kvl.tolist() == [55, 65, 75]
    assert cur.tolist() == [16, 27, 38]
    assert toks[0].tolist() == [0, 0, 0]
    for s in range(1, steps):
        assert toks[s].tolist() == [11 + s, 22 + s, 33 + s]


def test_hip_graph_replay():
    """A captured launch sequence replays with device-side state: out = W (x + step) via copy + gemv, step advanced by a kernel."""
    o = ops()
    K, N = 256, 64
    W = rnd(N, K, seed=1, scale=K ** -0.5).to(DEV)
    table = rnd(N, K, seed=2).to(DEV)             # one row per possible token: the argmax over N logits indexes it (16 rows read out of bounds for tokens >= 16)
    x = torch.zeros((1, K), dtype=BF16, device=DEV)
    y = torch.zeros((1, N), dtype=BF16, device=DEV)
    cur = torch.zeros(1, dtype=torch.int32, device=DEV)
    nxt = torch.zeros(1, dtype=torch.long, device=DEV)
    toks = torch.zeros((8, 1), dtype=torch.long, device=DEV)
    pos = torch.zeros(1, dtype=torch.long, device=DEV)
    kvl = torch.zeros(1, dtype=torch.int32, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)

    def launches():
        o.copy_rows(table, x, 1, K, src_rows=cur)
        o.gemv(x, W, y)
        o.argmax_into(y, nxt)
        o.decode_advance(nxt, cur, toks, pos, kvl, step, 1, 8)
    def reset():
        for t in (cur, nxt, toks, pos, kvl, step):
            t.zero_()
    reset()
    for _ in range(6):                           # the recurrence, launched eagerly
        launches()
    torch.cuda.synchronize()
    eager = toks.clone()
    assert step.item() == 6 and len(set(eager[1:7, 0].tolist())) > 1, "degenerate recurrence: the test would prove nothing"
    yy = ref_gemm(table[0:1].cpu(), W.cpu()).float()
    assert yy[0, int(eager[1, 0])] >= yy.max() - 2.0 ** -6 * yy.abs().max(), "first token is not the (near-)argmax of W x"
    reset()
    launches()                                   # step 0 eager (warm-up), steps 1..5 from the graph
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with o.HipGraph.capture(side) as g:
        launches()
    torch.cuda.synchronize()
    assert step.item() == 1, "capture must record, not execute"
    for _ in range(5):
        g.launch()
    side.synchronize()
    assert step.item() == 6 and pos.item() == 6 and kvl.item() == 6
    assert torch.equal(toks, eager), "graph replay must reproduce the eager launches bit for bit"


# ------------------------------------------------------------------------------------------------------------
# DecodeSession / generate_text against the packed prefill engine (MFMA path, golden-pinned in test_model_gpu.py)
# ------------------------------------------------------------------------------------------------------------
def _context(model, cfg, prompts):
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from oracle.configs import NEW_TOKEN_IDS_TINY, StubTokenizer
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    B = len(prompts)
    gi, lens, ropes = model.prepare_prompts([0] * B, [0] * B, prompts, tok, NEW_TOKEN_IDS_TINY)
    cache = model.forward_cache_update_text(NaiveCache(cfg["llm"]["num_hidden_layers"]), **gi)
    return cache, lens, ropes, model.prepare_start_tokens(lens, ropes, NEW_TOKEN_IDS_TINY)


def _prefill_engine_decode(model, cache, start, lens, tokens_in):
    """Teacher-forced stepping through MoTEngine.forward at Lq=1 (varlen MFMA attention, contiguous cache):
    returns the logits of every step."""
    o = ops()
    lm = model.language_model
    eng = lm.engine()
    B = len(lens)
    table, head = lm.model.embed_tokens.weight.data, lm.lm_head.weight.data
    pos = start["packed_query_position_ids"].clone()
    kv = list(lens)
    out = []
    for s in range(tokens_in.shape[0]):
        x = torch.empty((B, model.hidden_size), dtype=BF16, device=DEV)
        o.copy_rows(table, x, B, model.hidden_size, src_rows=tokens_in[s].to(torch.int32))
        plan = eng.plan([1] * B, pos, key_values_lens=kv)
        h = eng.forward(x, plan, "und", cache, update=True, causal=True)
        logits = torch.empty((B, head.shape[0]), dtype=BF16, device=DEV)
        o.gemm(h, head, logits, M0=B, variant=0)
        out.append(logits)
        kv = [k + 1 for k in kv]
        pos = pos + 1
    return torch.stack(out)


@pytest.mark.parametrize("name", ["tiny", "tiny_d128"])
@pytest.mark.parametrize("prompts", [["a small red cube"], ["sky", "a much longer prompt about nothing in particular"],
                                     # 20 requests: the two-block (17..32 rows) form of the batched projection + one RMSNorm launch in front of it (round 6)
                                     [("word " * (1 + i % 7)).strip() + f" {i}" for i in range(20)]],
                         ids=["one", "two", "twenty"])
def test_generate_text_graph_eager_and_prefill_engine(name, prompts):
    from oracle.configs import TINY, TINY_D128
    from tests.util_models import product_model
    cfg = {"tiny": TINY, "tiny_d128": TINY_D128}[name]
    model, _ = product_model(cfg)
    cache, lens, ropes, start = _context(model, cfg, prompts)
    n = 12
    c_graph, c_eager, c_ref = copy.deepcopy(cache), copy.deepcopy(cache), copy.deepcopy(cache)
    t_graph = model.generate_text(past_key_values=c_graph, max_length=n, end_token_id=None, use_graph=True, **start)
    sess = model._last_decode_session
    assert sess.graph is not None, f"hipGraph capture failed: {sess.graph_error}"
    last_logits = sess.logits.clone()
    t_eager = model.generate_text(past_key_values=c_eager, max_length=n, end_token_id=None, use_graph=False, **start)
    assert t_graph.shape == (n, len(prompts)) and t_graph.dtype == torch.int64
    assert torch.equal(t_graph, t_eager), "graph replay and eager stepping must be bit-identical"
    assert torch.equal(t_graph[0].cpu(), start["packed_start_tokens"])
    # the caller's cache received the new rows (reference: in-place update), identically on both paths
    L = cfg["llm"]["num_hidden_layers"]
    assert c_graph.seq_lens == cache.seq_lens + n * len(prompts)
    for i in range(L):
        assert torch.equal(c_graph.key_cache[i], c_eager.key_cache[i]) and torch.equal(c_graph.value_cache[i], c_eager.value_cache[i])
    # teacher-forced comparison with the packed prefill engine on the same tokens
    ref_logits = _prefill_engine_decode(model, c_ref, start, lens, t_graph)
    close(last_logits, ref_logits[-1], ulps=4, rel_l2=2e-2, what="last-step logits vs prefill engine")
    for i in range(L):
        close(c_graph.key_cache[i], c_ref.key_cache[i], ulps=4, rel_l2=1e-2, what=f"K cache layer {i} after decode")
        close(c_graph.value_cache[i], c_ref.value_cache[i], ulps=4, rel_l2=1e-2, what=f"V cache layer {i} after decode")
    # greedy ids: equal to the prefill engine's argmax up to the first near tie (same rule as test_model_gpu.py)
    for s in range(1, n):
        ours, row = t_graph[s].cpu(), ref_logits[s - 1].float().cpu()
        theirs = row.argmax(-1)
        if torch.equal(ours, theirs):
            continue
        gap = row.max(-1).values - row.gather(-1, ours.view(-1, 1)).squeeze(-1)
        assert (gap <= row.abs().max().item() * 2.0 ** -6).all(), f"step {s}: tokens differ without a near tie"


def test_generate_text_end_token_and_scattered_pages():
    from bagel_amd.modeling.bagel.decode import DecodeSession
    from oracle.configs import TINY_D128 as cfg
    from tests.util_models import product_model
    model, _ = product_model(cfg)
    cache, lens, ropes, start = _context(model, cfg, ["a small red cube"])
    full = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=10, end_token_id=None, **start)
    stop = int(full[4, 0])
    first = next(s for s in range(1, 10) if int(full[s, 0]) == stop)      # the loop stops when this id is first produced
    cut = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=10, end_token_id=stop, **start)
    assert cut.shape[0] == first and torch.equal(cut, full[:first])
    # physically scattered pages give the same tokens
    lm = model.language_model
    pages = (lens[0] + 10 + 63) // 64
    order = list(reversed(range(pages)))
    sess = DecodeSession(lm.engine(), lm.model.embed_tokens.weight.data, lm.lm_head.weight.data, copy.deepcopy(cache), lens,
                         start["packed_start_tokens"], start["packed_query_position_ids"], 10, page_order=order)
    for _ in range(10):
        sess.step()
    assert torch.equal(sess.tokens_so_far(), full)


def test_generate_text_sampling_runs_and_is_seeded(monkeypatch):
    """do_sample=True: the draw happens on the device INSIDE the captured step (Gumbel-max, bagel_sample_gumbel_bf16), so the sampled decode replays from the
    hipGraph; reproducible under torch.manual_seed, graph == eager, a new seed gives new tokens; BAGEL_DECODE_SAMPLER=torch keeps torch.multinomial (eager steps)."""
    from oracle.configs import TINY as cfg
    from tests.util_models import product_model
    model, _ = product_model(cfg)
    cache, lens, ropes, start = _context(model, cfg, ["sky"])
    outs = {}
    for tag, seed, graph in (("a", 5, True), ("b", 5, True), ("eager", 5, False), ("other", 6, True)):
        torch.manual_seed(seed)
        outs[tag] = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=12, do_sample=True, temperature=1.3, end_token_id=None, use_graph=graph, **start)
        sess = model._last_decode_session
        assert sess.sampler is not None and (sess.graph is not None) == graph
        if graph:
            assert sess._graph_has_advance, "a device-sampled step is captured WITH its bookkeeping"
    assert outs["a"].shape == (12, 1) and torch.equal(outs["a"], outs["b"]) and torch.equal(outs["a"], outs["eager"])
    assert not torch.equal(outs["a"], outs["other"])
    assert (outs["a"] >= 0).all() and (outs["a"] < cfg["llm"]["vocab_size"]).all()
    monkeypatch.setenv("BAGEL_DECODE_SAMPLER", "torch")
    t = []
    for _ in range(2):
        torch.manual_seed(5)
        t.append(model.generate_text(past_key_values=copy.deepcopy(cache), max_length=6, do_sample=True, temperature=0.7, end_token_id=None, **start))
    assert model._last_decode_session.sampler is None and torch.equal(t[0], t[1])


def test_sample_gumbel_kernel_matches_its_restatement_and_the_softmax_distribution():
    """bagel_sample_gumbel_bf16 vs oracle/sampling.py (Philox4x32-10 pinned to the Random123 known-answer vectors on the CPU side) at the lm_head width and at ragged widths:
    identical ids except where the two top perturbed scores are within float rounding of each other; and the empirical distribution of 40 960 draws over a 13-way
    head within 4.5 sigma of softmax(logits / T) (bagel.py:980-983)."""
    import numpy as np
    from oracle import sampling as S
    o = ops()
    g = torch.Generator().manual_seed(3)
    for rows, cols, T in ((3, 152064, 0.7), (5, 1001, 1.0), (4, 7, 2.5)):
        logits = (torch.randn(rows, cols, generator=g) * 3).to(torch.bfloat16)
        for step in (0, 1, 77):
            ctr = torch.tensor([step], dtype=torch.int32, device=DEV)
            out = torch.full((rows,), -1, dtype=torch.int64, device=DEV)
            o.sample_gumbel_into(logits.to(DEV), out, T, 0x0123456789abcdef & (2 ** 62 - 1), ctr)
            keys = S.gumbel_keys(logits.float().numpy(), T, 0x0123456789abcdef & (2 ** 62 - 1), step)
            want = keys.argmax(1)
            got = out.cpu().numpy()
            for b in range(rows):
                if got[b] != want[b]:                           # a float-rounding near tie between the two best perturbed scores (logf differs by an ulp)
                    assert abs(keys[b, got[b]] - keys[b, want[b]]) <= 1e-5 * max(1.0, abs(keys[b, want[b]])), (rows, cols, step, b)
    V, T, rows = 13, 0.7, 8192
    lg = (torch.randn(V, generator=g) * 2).to(torch.bfloat16)
    counts = np.zeros(V)
    x = lg.to(DEV).repeat(rows, 1).contiguous()
    out = torch.empty(rows, dtype=torch.int64, device=DEV)
    for step in range(5):
        o.sample_gumbel_into(x, out, T, 99, torch.tensor([step], dtype=torch.int32, device=DEV))
        counts += np.bincount(out.cpu().numpy(), minlength=V)
    z = S.bf16_round(lg.float().numpy() / np.float32(T)).astype(np.float64)
    p = np.exp(z - z.max()); p /= p.sum()
    n = counts.sum()
    assert (np.abs(counts - n * p) <= 4.5 * np.sqrt(n * p * (1 - p)) + 1).all(), (counts, n * p)


def test_gumbel_value_on_the_device_is_finite_on_the_edge_draws():
    """The sampler's uniform -> Gumbel map on the draws no seed can be steered to (bagel_debug_gumbel_of_u32): x = 0xFFFFFFFF used to give u == 1.0 and a
    Gumbel value of +inf (2^-24 per column = ~0.9 % of the decode steps at the lm_head width picked a uniformly random token).  Also == the restatement."""
    import numpy as np
    from oracle import sampling as S
    xs = np.array([0, 1, 0x1ff, 0x200, 0x7fffffff, 0x80000000, 0xfffffdff, 0xfffffe00, 0xffffffff] + list(range(0xffff0000, 0xffffffff, 0x101)), dtype=np.uint32)
    g = ops().debug_gumbel_of_u32(torch.from_numpy(xs.view(np.int32)).to(DEV)).cpu().numpy()
    assert np.isfinite(g).all(), g[:9]
    np.testing.assert_allclose(g, S.gumbel_of(xs), rtol=3e-6, atol=3e-6)
    assert (np.diff(g[:9]) >= 0).all()


@pytest.mark.parametrize("max_length", [1, 2, 3])
def test_generate_text_short_runs_and_empty_context(max_length):
    """1-2 steps never reach the graph capture; a decode from an EMPTY cache (no prefill) works and matches the eager path."""
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from oracle.configs import TINY_D128 as cfg, NEW_TOKEN_IDS_TINY
    from tests.util_models import product_model
    model, _ = product_model(cfg)
    cache, lens, ropes, start = _context(model, cfg, ["sky"])
    a = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=max_length, end_token_id=None, use_graph=True, **start)
    b = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=max_length, end_token_id=None, use_graph=False, **start)
    assert a.shape == (max_length, 1) and torch.equal(a, b)
    L = cfg["llm"]["num_hidden_layers"]
    empty = NaiveCache(L)
    st = model.prepare_start_tokens([0], [0], NEW_TOKEN_IDS_TINY)
    t = model.generate_text(past_key_values=empty, max_length=4, end_token_id=None, **st)
    assert t.shape == (4, 1) and empty.seq_lens == 4, "the decoded rows must land in the (previously empty) cache"
    assert model.generate_text(past_key_values=NaiveCache(L), max_length=0, end_token_id=None, **st).shape == (0, 1)


# ------------------------------------------------------------------------------------------------------------
# weight-only INT8 decode option (MI355X analogue of the reference's quantised load modes, app.py:114-131)
# ------------------------------------------------------------------------------------------------------------
def test_quantize_rows_i8_exact():
    W = rnd(37, 272, seed=1, scale=0.05)
    W[5] = 0                                                        # an all-zero row must not divide by zero
    q, s = ops().quantize_rows_i8(W.to(DEV))
    amax = W.float().abs().amax(dim=1)
    s_ref = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
    q_ref = (torch.round(W.float() / s_ref[:, None]).clamp(-127, 127) + 128).to(torch.uint8)
    assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu(), q_ref)
    deq = (q.cpu().float() - 128) * s.cpu()[:, None]
    assert (deq - W.float()).abs().max() <= 0.5 * s_ref.max() + 1e-9


@pytest.mark.parametrize("M,N,K", [(1, 256, 3584), (1, 64, 18944), (2, 130, 256), (3, 96, 512), (4, 72, 1040), (1, 4608, 3584), (5, 34, 16)])
def test_gemv_w8_matches_dequantised_reference(M, N, K):
    """y = s_n (sum u8 x - 128 sum x) equals the fp32 product with the DEQUANTISED weights to fp32 rounding (then one bf16 rounding)."""
    o = ops()
    A, W, b, R = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1), rnd(M, N, seed=4)
    q, s = o.quantize_rows_i8(W.to(DEV))
    Wd = ((q.cpu().float() - 128) * s.cpu()[:, None])
    ref = (A.float() @ Wd.t() + b.float()).to(BF16)
    C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    o.gemv_w8(A.to(DEV), q, s, C, bias=b.to(DEV))
    close(C, ref, what=f"gemv_w8 {M}x{N}x{K}")
    X = R.to(DEV).clone()
    o.gemv_w8(A.to(DEV), q, s, X, bias=b.to(DEV), residual=X)
    close(X, R + ref, ulps=2, what="gemv_w8 residual")
    # and it is a faithful approximation of the bf16 product: row-wise absmax INT8 keeps ~0.5 % relative error
    full = (A.float() @ W.float().t() + b.float())
    err = ((C.float().cpu() - full).norm() / full.norm()).item()
    assert err < 2e-2, f"int8 weights: rel_l2 {err:.3g} vs the bf16-weight product"


def test_gemv_w8_swiglu_and_fused_norm():
    from bagel_amd.modeling.bagel.qwen2_navit import interleave_gate_up
    o = ops()
    M, I, K = 2, 96, 512
    A, Wg, Wu = rnd(M, K, seed=1, scale=2.0), rnd(I, K, seed=2, scale=K ** -0.5), rnd(I, K, seed=3, scale=K ** -0.5)
    w = (1.0 + 0.1 * rnd(K, seed=4).float()).to(BF16)
    Wi = interleave_gate_up(Wg, Wu)
    q, s = o.quantize_rows_i8(Wi.to(DEV))
    Wd = ((q.cpu().float() - 128) * s.cpu()[:, None]).view(I // 16, 2, 16, K)
    Wgd, Wud = Wd[:, 0].reshape(I, K), Wd[:, 1].reshape(I, K)
    h = ref_rmsnorm(A, w, 1e-6)
    ref = F.silu((h.float() @ Wgd.t()).to(BF16)) * (h.float() @ Wud.t()).to(BF16)
    C = torch.empty((M, I), dtype=BF16, device=DEV)
    o.gemv_w8(A.to(DEV), q, s, C, epilogue=3, norm_w=w.to(DEV), eps=1e-6)
    close(C, ref, ulps=2, what="gemv_w8 rmsnorm+swiglu")


def test_generate_text_int8_weights_option():
    """weight_quant='int8_rowwise': same loop, graph replay == eager, logits within the quantisation noise of the bf16 path."""
    from oracle.configs import TINY_D128 as cfg
    from tests.util_models import product_model
    model, _ = product_model(cfg)
    cache, lens, ropes, start = _context(model, cfg, ["a small red cube"])
    n = 8
    ref = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, **start)
    ref_logits = model._last_decode_session.logits.float().clone()
    a = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, weight_quant="int8_rowwise", use_graph=True, **start)
    sess = model._last_decode_session
    assert sess.weight_quant == "int8_rowwise" and sess.graph is not None
    q_logits = sess.logits.float().clone()
    b = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, weight_quant="int8_rowwise", use_graph=False, **start)
    assert torch.equal(a, b) and a.shape == ref.shape
    if torch.equal(a, ref):          # same token history -> the last-step logits are comparable
        err = ((q_logits - ref_logits).norm() / ref_logits.norm()).item()
        assert err < 5e-2, f"int8-weight logits differ from bf16-weight logits by rel_l2 {err:.3g}"
    assert torch.equal(a[0], ref[0])
    model.decode_weight_quant = "int8_rowwise"                                     # the model-level switch (load-time mode in the reference)
    try:
        c = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, **start)
    finally:
        model.decode_weight_quant = None
    assert torch.equal(c, a)


def test_generate_text_mxfp4_weights_option():
    """weight_quant='mxfp4' (OCP-MX FP4 weights x FP8 activations on the block-scaled MFMA; the projection itself is pinned to its
    restatement in tests/test_mxfp4_gpu.py): same loop, graph replay == eager bit for bit, first-step logits within the 4-bit noise of
    the bf16 path and not equal to it, model-level switch."""
    from oracle.configs import TINY_D128 as cfg
    from tests.util_models import product_model
    model, _ = product_model(cfg)
    cache, lens, ropes, start = _context(model, cfg, ["a small red cube"])
    ref = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=1, end_token_id=None, **start)
    ref_logits = model._last_decode_session.logits.float().clone()
    one = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=1, end_token_id=None, weight_quant="mxfp4", **start)
    q_logits = model._last_decode_session.logits.float().clone()
    assert one.shape == ref.shape and torch.isfinite(q_logits).all()
    err = ((q_logits - ref_logits).norm() / ref_logits.norm()).item()
    assert 1e-3 < err < 0.4, f"mxfp4-weight logits vs bf16-weight logits: rel_l2 {err:.3g}"
    # ... and against the oracle's decode step with the MXFP4 scheme switched into the same seven linears per layer (bf16 prefill on
    # both sides): the whole quantised step, not only the projection, is what the restatement says
    from oracle import bagel_oracle as O
    from oracle.configs import NEW_TOKEN_IDS_TINY, StubTokenizer
    from tests.util_models import oracle_weights
    W, _ = oracle_weights(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    gi, l2, r2 = model.prepare_prompts([0], [0], ["a small red cube"], StubTokenizer(cfg["llm"]["vocab_size"]), NEW_TOKEN_IDS_TINY)
    oc = O.forward_cache_update_text(W, cfg, O.OracleCache(L), **gi)
    O.MXFP4_WEIGHT_PTRS = O.mxfp4_decode_weight_ptrs(W)
    try:
        st = {k: torch.as_tensor(v).cpu() for k, v in start.items()}
        _, ologits = O.generate_text(W, cfg, oc, st["packed_key_value_indexes"], st["key_values_lens"], st["packed_start_tokens"],
                                     st["packed_query_position_ids"], 1, return_logits=True)
    finally:
        O.MXFP4_WEIGHT_PTRS = set()
    ol = torch.as_tensor(ologits[0]).float().reshape(q_logits.shape)
    e2 = ((q_logits.cpu() - ol).norm() / ol.norm()).item()
    e_bf16 = ((ref_logits.cpu() - ol).norm() / ol.norm()).item()
    # tolerance: as for the FP8 gen expert (tests/test_fp8_gpu.py) -- an e4m3 activation code next to a rounding boundary flips on a
    # bf16-level difference of its input (one code step = 6 % of the element), a noise source the bf16 path does not have.  Measured
    # 4.4e-2 on MI355X, with the bf16 path 2.1e-1 away from the same restatement.
    assert e2 < 8e-2 and e2 < 0.5 * e_bf16, f"mxfp4 decode step vs its restatement: rel_l2 {e2:.3g} (bf16 path vs the same restatement: {e_bf16:.3g})"
    n = 8
    a = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, weight_quant="mxfp4", use_graph=True, **start)
    sess = model._last_decode_session
    assert sess.weight_quant == "mxfp4" and sess.graph is not None
    b = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, weight_quant="mxfp4", use_graph=False, **start)
    assert torch.equal(a, b)
    model.decode_weight_quant = "mxfp4"
    try:
        c = model.generate_text(past_key_values=copy.deepcopy(cache), max_length=n, end_token_id=None, **start)
    finally:
        model.decode_weight_quant = None
    assert torch.equal(c, a)


@pytest.mark.parametrize("cpw", ["2", "3", "8"])
def test_decode_attention_with_several_chunks_per_workgroup(cpw):
    """The batched-decode form of the attention kernel: a workgroup walks `cpw` consecutive 128-key chunks with a running (max, sum, O) and
    the next chunk's page rows prefetched (BAGEL_DEC_CPW; chosen automatically once requests x KV heads x chunks exceed one round of
    workgroups).  The attention tests of this file -- ragged lengths, scattered pages, NaN-filled never-written slots, the fused q/k-norm +
    RoPE + append prologue bit-identical to the two-kernel form, the sharp-softmax merge -- re-run in a process with the loop forced on
    (the knob is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BAGEL_DEC_CPW=cpw)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_decode_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "paged_append_and_decode_attention or fused_decode_attention_equals or sharp_softmax"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------------------
# NaiveCache.__deepcopy__ is copy-on-write (inferencer.py:189,230-231,244,253 deep-copies the context per request stream)
# ------------------------------------------------------------------------------------------------------------
def test_naive_cache_deepcopy_is_copy_on_write_and_isolated():
    """A 1 100-token context on the tiny model: ``copy.deepcopy`` allocates (almost) nothing; the copy and the original stay independent -- a decode that
    appends to the COPY leaves the original's rows and length untouched and vice versa -- and the first write pays for ONE clone of the layers it touches.
    ``NaiveCache.concat`` with a single live stream (text->image: the CFG stream has no context) shares instead of copying."""
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from oracle.configs import TINY_D128 as cfg
    from tests.util_models import product_model
    model, _ = product_model(cfg)
    L = cfg["llm"]["num_hidden_layers"]
    cache, lens, ropes, start = _context(model, cfg, ["w " * 1100])
    assert lens[0] >= 1000
    torch.cuda.synchronize()
    snap_k = [cache.key_cache[i].clone() for i in range(L)]
    per_cache = sum(cache._k[i].numel() + cache._v[i].numel() for i in range(L)) * 2
    m0 = torch.cuda.memory_allocated()
    copies = [copy.deepcopy(cache) for _ in range(3)]
    m1 = torch.cuda.memory_allocated()
    assert m1 - m0 < 0.05 * per_cache, f"deepcopy x3 allocated {m1 - m0} bytes for caches of {per_cache} bytes: not copy-on-write"
    for c in copies:
        assert c.seq_lens == cache.seq_lens
        assert all(torch.equal(c.key_cache[i], snap_k[i]) for i in range(L))
    shared = NaiveCache.concat([copies[2], None], [1, 1])
    assert torch.cuda.memory_allocated() - m1 < 0.05 * per_cache, "concat with one live stream must share its buffers"
    assert shared.lens(0) == [lens[0], 0]
    # decode on copy 0: it must clone (once), everybody else keeps the context bit for bit
    t0 = model.generate_text(past_key_values=copies[0], max_length=5, end_token_id=None, **start)
    m2 = torch.cuda.memory_allocated()
    assert copies[0].seq_lens == cache.seq_lens + 5 and cache.seq_lens == lens[0] and copies[1].seq_lens == lens[0]
    for c in (cache, copies[1], copies[2]):
        assert all(torch.equal(c.key_cache[i], snap_k[i]) for i in range(L)), "a write through one copy reached another sharer"
    assert all(torch.equal(copies[0].key_cache[i][: lens[0]], snap_k[i]) for i in range(L))
    # the same decode through the ORIGINAL gives the same tokens and does not disturb copy 0's extra rows
    k0 = [copies[0].key_cache[i].clone() for i in range(L)]
    t1 = model.generate_text(past_key_values=cache, max_length=5, end_token_id=None, **start)
    assert torch.equal(t0, t1)
    assert all(torch.equal(copies[0].key_cache[i], k0[i]) for i in range(L))
    assert all(torch.equal(cache.key_cache[i], k0[i]) for i in range(L)), "same request, same cache contents"
    del m2
    # copy.copy == deepcopy (a shallow copy would alias the owner counters), and unshare() gives private storage for in-place edits through the views
    sh = copy.copy(copies[1])
    assert sh._own[0] is copies[1]._own[0] and sh._own[0][0] >= 2 and sh._k[0] is copies[1]._k[0]
    sh.unshare()
    assert sh._own[0][0] == 1 and sh._k[0] is not copies[1]._k[0]
    sh.key_cache[0].zero_()
    assert float(sh.key_cache[0].float().abs().max()) == 0.0
    assert all(torch.equal(copies[1].key_cache[i], snap_k[i]) for i in range(L)), "an in-place edit after unshare() reached a sharer"
    del sh
    assert copies[1]._own[1][0] >= 1
