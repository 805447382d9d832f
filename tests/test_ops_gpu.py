"""GPU parity of every HIP op against the CPU oracle / plain fp32 torch on the same seeded inputs.

Tolerance convention: a bf16 output may differ from the reference by accumulation order
only, i.e. by <= 1 bf16 ulp of the tensor's magnitude per op: max|d| <= 2^-7 * max|ref| (x2 for two chained roundings),
and the relative L2 error stays ~1e-3.  Integer / copy ops are bit-exact.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16
DEV = "cuda"


def ops():
    from bagel_amd import ops as o
    return o


def O():
    from oracle import bagel_oracle
    return bagel_oracle


def close(got, ref, ulps=1.0, rel_l2=4e-3, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    scale = ref.abs().max().item() + 1e-30
    d = (got - ref).abs()
    l2 = (got - ref).norm().item() / (ref.norm().item() + 1e-30)
    bad = d.max().item() > ulps * 2 ** -7 * scale or l2 > rel_l2
    if bad:
        idx = d.argmax().item()
        raise AssertionError(f"{what}: max|d|={d.max().item():.4g} (allowed {ulps * 2 ** -7 * scale:.4g}), rel_l2={l2:.3g} "
                             f"(allowed {rel_l2}), worst flat index {idx}: got {got.flatten()[idx].item()} ref {ref.flatten()[idx].item()}, "
                             f"frac exact={(d == 0).float().mean().item():.3f}")


def rnd(*shape, seed=0, scale=1.0, dtype=BF16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


# ------------------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------------------
def ref_gemm(A, W, bias=None, epi=0, residual=None):
    acc = A.float() @ W.float().t()
    if epi == 3:
        N = W.shape[0]
        a = acc.view(acc.shape[0], N // 32, 2, 16)
        g, u = a[:, :, 0].reshape(acc.shape[0], -1).to(BF16), a[:, :, 1].reshape(acc.shape[0], -1).to(BF16)
        return F.silu(g) * u
    if bias is not None:
        acc = acc + bias.float()
    c = acc.to(BF16)
    if epi == 1:
        c = F.gelu(c, approximate="tanh")
    elif epi == 2:
        c = F.silu(c)
    if residual is not None:
        c = residual + c
    return c


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (7, 136, 128), (300, 520, 200), (1000, 256, 144), (513, 384, 3584), (130, 64, 512)])
def test_gemm_plain_bias(M, N, K, variant):
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    ops().gemm(A.to(DEV), W.to(DEV), C, bias0=b.to(DEV), variant=variant)
    close(C, ref_gemm(A, W, b), what=f"gemm {M}x{N}x{K} v{variant}")


@pytest.mark.parametrize("epi", [1, 2])
def test_gemm_activation_and_residual(epi):
    M, N, K = 200, 264, 320
    A, W, b, R = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1), rnd(M, N, seed=4)
    C = torch.empty((M, N), dtype=BF16, device=DEV)
    ops().gemm(A.to(DEV), W.to(DEV), C, bias0=b.to(DEV), epilogue=epi)
    close(C, ref_gemm(A, W, b, epi), ulps=2, what=f"gemm epi{epi}")
    X = R.to(DEV).clone()
    ops().gemm(A.to(DEV), W.to(DEV), X, bias0=b.to(DEV), residual=X)      # in-place residual, as the layer loop uses it
    close(X, ref_gemm(A, W, b, 0, R), ulps=2, what="gemm residual in place")


@pytest.mark.parametrize("variant", [3, 5, None])
@pytest.mark.parametrize("N,K,mode", [(1152, 4352, "res"), (1152, 2048, "res"), (4352, 1152, "gelu"), (1152, 1152, "res")])
def test_gemm_vit_epilogues_at_the_tower_shapes(N, K, mode, variant):
    """SigLIP's out / fc2 (bias + residual, in place) and fc1 (bias + GELU-tanh) at a 980^2 image (M = 4900; siglip_navit.py:216-258) against fp32: on the
    one-tile ping-pong kernel (3), on the persistent kernel's round-6 epilogue modes 5 / 6 incl. the K-split of its leftover tiles (5) and as ops.gemm routes them."""
    M = 4900
    A, W, b, R = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1), rnd(M, N, seed=4)
    if mode == "res":
        C = R.to(DEV).clone()
        ops().gemm(A.to(DEV), W.to(DEV), C, bias0=b.to(DEV), residual=C, variant=variant)
        close(C, ref_gemm(A, W, b, 0, R), ulps=2, what=f"gemm bias + residual {M}x{N}x{K} v{variant}")
    else:
        C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
        ops().gemm(A.to(DEV), W.to(DEV), C, bias0=b.to(DEV), epilogue=1, variant=variant)
        close(C, ref_gemm(A, W, b, 1), ulps=2, what=f"gemm bias + gelu {M}x{N}x{K} v{variant}")


@pytest.mark.parametrize("variant", [0, 1, 3, 4, 5])
def test_gemm_swiglu(variant):
    from bagel_amd.modeling.bagel.qwen2_navit import interleave_gate_up
    M, I, K = 333, 416, 256
    A, Wg, Wu = rnd(M, K, seed=1), rnd(I, K, seed=2, scale=K ** -0.5), rnd(I, K, seed=3, scale=K ** -0.5)
    Wi = interleave_gate_up(Wg, Wu)
    C = torch.empty((M, I), dtype=BF16, device=DEV)
    ops().gemm(A.to(DEV), Wi.to(DEV), C, epilogue=3, variant=variant)
    ref = F.silu((A.float() @ Wg.float().t()).to(BF16)) * (A.float() @ Wu.float().t()).to(BF16)
    close(C, ref, ulps=2, what="gemm swiglu")


@pytest.mark.parametrize("variant", [0, 1, 3, 4, 5])
def test_gemm_two_expert_groups(variant):
    """MoT routing: text rows -> W0, latent rows -> W1, rows interleaved like <start> latents <end> per sample."""
    H, N = 256, 392
    lens = [3, 70, 260]
    rows_t, rows_v, base = [], [], 0
    for n in lens:
        rows_t += [base, base + n + 1]
        rows_v += list(range(base + 1, base + n + 1))
        base += n + 2
    M = base
    A, W0, W1 = rnd(M, H, seed=1), rnd(N, H, seed=2, scale=H ** -0.5), rnd(N, H, seed=3, scale=H ** -0.5)
    b0, b1, R = rnd(N, seed=4, scale=0.1), rnd(N, seed=5, scale=0.1), rnd(M, N, seed=6)
    it = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)  # noqa: E731
    C = R.to(DEV).clone()
    ops().gemm(A.to(DEV), W0.to(DEV), C, bias0=b0.to(DEV), a_rows0=it(rows_t), c_rows0=it(rows_t), M0=len(rows_t),
               W1=W1.to(DEV), bias1=b1.to(DEV), a_rows1=it(rows_v), c_rows1=it(rows_v), M1=len(rows_v), residual=C, variant=variant)
    ref = torch.empty(M, N, dtype=BF16)
    ref[rows_t] = ref_gemm(A[rows_t], W0, b0, 0, R[rows_t])
    ref[rows_v] = ref_gemm(A[rows_v], W1, b1, 0, R[rows_v])
    close(C, ref, ulps=2, what="grouped gemm")


def test_gemm_pingpong_race_screen():
    """variant 3 (two-group ping-pong, hand-counted DMA waits) on a multi-tile problem: every repeat must be BIT-IDENTICAL
    to the first (a missed vmcnt/barrier shows up as a sporadic mismatch) and within 1 bf16 ulp of the plain tile kernel
    (the two use different MFMA shapes, hence a different summation association)."""
    M, N, K = 1500, 1280, 2048
    A, W, b = rnd(M, K, seed=11).to(DEV), rnd(N, K, seed=12, scale=K ** -0.5).to(DEV), rnd(N, seed=13, scale=0.1).to(DEV)
    C0 = torch.empty((M, N), dtype=BF16, device=DEV)
    ops().gemm(A, W, C0, bias0=b, variant=0)
    first = None
    for rep in range(6):
        C3 = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
        ops().gemm(A, W, C3, bias0=b, variant=3)
        if first is None:
            first = C3
            close(C3, C0.cpu(), ulps=1, what="ping-pong vs tile kernel")
        else:
            assert torch.equal(first.view(torch.int16), C3.view(torch.int16)), f"ping-pong GEMM is not deterministic (repeat {rep})"


@pytest.mark.parametrize("wgs", ["8", "24", ""])
def test_gemm_persistent_identical_to_pingpong(wgs):
    """variant 4 (persistent ping-pong: tile loop inside the kernel, next tile's DMA under the current epilogue, MoT row lists
    through an LDS table ring) must reproduce variant 3 BIT FOR BIT on every epilogue mode, with and without row lists, over
    repeats -- with 8 / 24 workgroups (every workgroup walks several tiles even on these small problems) and with the default
    one-per-CU grid.  Own process: the grid size is read once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if wgs:
        env["BAGEL_GEMM_PERSIST_WGS"] = wgs
    else:
        env.pop("BAGEL_GEMM_PERSIST_WGS", None)
    env["BAGEL_GEMM_SPLITK"] = "0"        # bit identity is a property of the one-pass schedule (the K-split of leftover tiles re-associates)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gemm_persist_check.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL IDENTICAL" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("wgs", ["8", "24"])
def test_gemm_leftover_k_split_small_grids(wgs):
    """The K-split schedule of bagel_gemm_bf16_ws (full rounds, leftover tiles cut along K, reduce + epilogue) on small problems with 8 / 24
    persistent workgroups, where almost every shape leaves a partial round: every epilogue it serves (bias, residual, plain), two
    expert groups with gather / scatter row lists, ragged M / N edges -- against fp32.  Own process (the grid size is read once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BAGEL_GEMM_PERSIST_WGS=wgs, BAGEL_GEMM_SPLITK="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gemm_persist_check.py"), "--tolerance"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL WITHIN TOLERANCE" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_gemm_gather_rows_dense_out():
    """llm2vae(out)[vae rows] -> dense (bagel.py:832-833): a_rows gather, identity C rows."""
    M, K, N = 150, 192, 64
    rows = list(range(1, 60)) + list(range(70, 140))
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    C = torch.empty((len(rows), N), dtype=BF16, device=DEV)
    ops().gemm(A.to(DEV), W.to(DEV), C, bias0=b.to(DEV), a_rows0=torch.tensor(rows, dtype=torch.int32, device=DEV), M0=len(rows))
    close(C, ref_gemm(A[rows], W, b), what="gemm gather")


# ------------------------------------------------------------------------------------------------------------
# norms, rope
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(1, 128), (5, 512), (333, 3584), (64, 1152)])
def test_rmsnorm_and_expert_routing(rows, cols):
    x, w0, w1 = rnd(rows, cols, seed=1, scale=3.0), (1 + 0.1 * rnd(cols, seed=2).float()).to(BF16), (1 + 0.1 * rnd(cols, seed=3).float()).to(BF16)
    y = torch.empty((rows, cols), dtype=BF16, device=DEV)
    ops().rmsnorm(x.to(DEV), w0.to(DEV), y, 1e-6)
    close(y, O().rmsnorm(x, w0, 1e-6), what="rmsnorm")
    ex = (torch.arange(rows) % 3 == 1).to(torch.int32)
    ops().rmsnorm(x.to(DEV), w0.to(DEV), y, 1e-6, w1=w1.to(DEV), expert=ex.to(DEV))
    ref = O().rmsnorm(x, w0, 1e-6)
    ref[ex.bool()] = O().rmsnorm(x[ex.bool()], w1, 1e-6)
    close(y, ref, what="rmsnorm routed")


@pytest.mark.parametrize("rows,cols", [(3, 64), (100, 144), (257, 1152)])
def test_layernorm(rows, cols):
    x, w, b = rnd(rows, cols, seed=1, scale=2.0), (1 + 0.1 * rnd(cols, seed=2).float()).to(BF16), rnd(cols, seed=3, scale=0.1)
    y = torch.empty((rows, cols), dtype=BF16, device=DEV)
    ops().layernorm(x.to(DEV), w.to(DEV), b.to(DEV), y, 1e-6)
    close(y, F.layer_norm(x, (cols,), w, b, 1e-6), what="layernorm")


@pytest.mark.parametrize("hd", [32, 64, 128])
def test_rope_table(hd):
    pos = torch.tensor([0, 1, 2, 17, 100, 1023, 4097, 4097, 31999], dtype=torch.long)
    inv = 1.0 / (1e6 ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    cos, sin = ops().rope_table(pos.to(DEV), inv.to(DEV))
    rc, rs = O().rope_tables(pos, hd, 1e6, BF16)
    # cos/sin of fp32 angles may differ by 1 fp32 ulp between libms -> at most a rare 1-ulp bf16 flip
    close(cos, rc[:, : hd // 2], ulps=1, rel_l2=2e-3, what="rope cos")
    close(sin, rs[:, : hd // 2], ulps=1, rel_l2=2e-3, what="rope sin")
    assert (cos.cpu() == rc[:, : hd // 2]).float().mean() > 0.98


@pytest.mark.parametrize("gen", [0, 1])
@pytest.mark.parametrize("hd,dp,nq,nkv", [(128, 128, 4, 2), (128, 128, 28, 4), (32, 64, 4, 2), (64, 64, 3, 1)])
def test_qknorm_rope(hd, dp, nq, nkv, gen):
    rows = 37
    g = torch.Generator().manual_seed(5)
    q = rnd(rows, nq, hd, seed=1)
    k = rnd(rows, nkv, hd, seed=2)
    v = rnd(rows, nkv, hd, seed=3)
    pos = torch.randint(0, 3000, (rows,), generator=g)
    cos, sin = O().rope_tables(pos, hd, 1e6, BF16)
    w = [(1 + 0.1 * rnd(hd, seed=10 + i).float()).to(BF16) for i in range(4)]   # q0 k0 q1 k1
    ex = (torch.arange(rows) % 4 != 0).to(torch.int32) if gen else None

    def pad(t, n):
        out = torch.zeros(rows, n, dp, dtype=BF16)
        out[:, :, :hd] = t
        return out.reshape(rows, n * dp)
    qkv = torch.cat([pad(q, nq), pad(k, nkv), pad(v, nkv)], 1).to(DEV)
    ops().qknorm_rope(qkv, cos[:, : hd // 2].contiguous().to(DEV), sin[:, : hd // 2].contiguous().to(DEV), w[0].to(DEV), w[1].to(DEV),
                      w[2].to(DEV) if gen else None, w[3].to(DEV) if gen else None, ex.to(DEV) if gen else None, nq, nkv, hd, dp,
                      1e-6, gen, 1)
    if gen:
        qf, kf = q.float(), k.float()
        t, vv = ex == 0, ex == 1
        qf[t], qf[vv] = O().rmsnorm(qf[t], w[0], 1e-6), O().rmsnorm(qf[vv], w[2], 1e-6)
        kf[t], kf[vv] = O().rmsnorm(kf[t], w[1], 1e-6), O().rmsnorm(kf[vv], w[3], 1e-6)
        rq, rk = O().apply_rope(qf, kf, cos, sin)
    else:
        rq, rk = O().apply_rope(O().rmsnorm(q, w[0], 1e-6), O().rmsnorm(k, w[1], 1e-6), cos, sin)
    got = qkv.cpu().view(rows, nq + 2 * nkv, dp)
    close(got[:, :nq, :hd], rq.to(BF16), what="q norm+rope")
    close(got[:, nq:nq + nkv, :hd], rk.to(BF16), what="k norm+rope")
    assert torch.equal(got[:, nq + nkv:, :hd], v), "v must be untouched"
    if dp != hd:
        assert (got[:, :, hd:] == 0).all(), "padding lanes must stay zero"


# ------------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------------
def run_attention(q_lens, ctx_lens, nq, nkv, D, causal, seed=0, scale_dim=None, planned=None):
    """Builds packed q/k/v (+context), runs v_transpose + attn_varlen, returns (got, ref).
    ``planned`` = dict(n_workers=, split_min_tiles=): also run the persistent planned kernel on the same device tensors and return
    (got, ref, got_planned, plan)."""
    B = len(q_lens)
    M, Ctot = sum(q_lens), sum(ctx_lens)
    q, k, v = rnd(M, nq, D, seed=seed + 1), rnd(M, nkv, D, seed=seed + 2), rnd(M, nkv, D, seed=seed + 3)
    kc, vc = rnd(max(Ctot, 1), nkv, D, seed=seed + 4), rnd(max(Ctot, 1), nkv, D, seed=seed + 5)
    scale = (scale_dim or D) ** -0.5
    # reference on the merged layout
    mk, mv, cu_k = [], [], [0]
    qo = co = 0
    for b in range(B):
        mk += [kc[co:co + ctx_lens[b]], k[qo:qo + q_lens[b]]]
        mv += [vc[co:co + ctx_lens[b]], v[qo:qo + q_lens[b]]]
        qo += q_lens[b]; co += ctx_lens[b]
        cu_k.append(cu_k[-1] + ctx_lens[b] + q_lens[b])
    cu_q = [0]
    for n in q_lens:
        cu_q.append(cu_q[-1] + n)
    from oracle.bagel_oracle import attn_varlen
    ref = attn_varlen(q, torch.cat(mk), torch.cat(mv), torch.tensor(cu_q, dtype=torch.int32), torch.tensor(cu_k, dtype=torch.int32),
                      max(q_lens), 0, softmax_scale=scale, causal=causal)
    # device side: fused [q|k|v] rows like the layer loop produces
    o = ops()
    qkv = torch.cat([q.reshape(M, -1), k.reshape(M, -1), v.reshape(M, -1)], 1).to(DEV)
    qw, kw = nq * D, nkv * D
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)  # noqa: E731

    def cols(lens):
        out, c = [], 0
        for n in lens:
            out.append(c)
            c += (max(n, 1) + 63) // 64 * 64
        return out, c
    vcol, vtot = cols(q_lens)
    vt = torch.zeros((kw, (vtot + 255) // 256 * 256), dtype=BF16, device=DEV)
    o.v_transpose(qkv[:, qw + kw:], vt, i32(cu_q), i32(vcol), B, max(q_lens), nkv, D)
    kwargs = {}
    if Ctot > 0:
        cu_c = [0]
        for n in ctx_lens:
            cu_c.append(cu_c[-1] + n)
        ccol, ctot = cols(ctx_lens)
        vtc = torch.zeros((kw, (ctot + 255) // 256 * 256), dtype=BF16, device=DEV)
        vcd = vc.reshape(Ctot, -1).to(DEV)
        o.v_transpose(vcd, vtc, i32(cu_c), i32(ccol), B, max(ctx_lens), nkv, D)
        kwargs = dict(k_ctx=kc.reshape(Ctot, -1).to(DEV), vt_ctx=vtc, cu_ctx=i32(cu_c), vt_ctx_col=i32(ccol))
    out = torch.full((M, qw), float("nan"), dtype=BF16, device=DEV)
    o.attn_varlen(qkv[:, :qw], qkv[:, qw:qw + kw], vt, out, i32(cu_q), i32(vcol), B, max(q_lens), nq, nkv, D, causal, scale, **kwargs)
    if planned is None:
        return out.view(M, nq, D), ref
    if planned.get("n_workers") == "2xchip":      # more workers than CUs: the split-ring form of the planned kernel
        planned = dict(planned, n_workers=2 * (torch.cuda.get_device_properties(0).multi_processor_count // 8 * 8))
    ap = o.AttnPlan(cu_q[:-1], q_lens, vcol, nq, nkv, D, causal, DEV, ctx_start=cu_c[:-1] if Ctot > 0 else None,
                    ctx_len=ctx_lens if Ctot > 0 else None, vt_ctx_col=ccol if Ctot > 0 else None, **planned)
    out2 = torch.full((M, qw), float("nan"), dtype=BF16, device=DEV)
    o.attn_planned(qkv[:, :qw], qkv[:, qw:qw + kw], vt, out2, ap, scale, k_ctx=kwargs.get("k_ctx"), vt_ctx=kwargs.get("vt_ctx"))
    return out.view(M, nq, D), ref, out2.view(M, nq, D), ap


ATTN_CASES = [
    # q_lens, ctx_lens, nq, nkv, D, causal
    ([1], [0], 4, 2, 128, True),
    ([5, 64, 65], [0, 0, 0], 4, 2, 128, True),
    ([130, 7], [0, 0], 4, 2, 128, False),
    ([18, 10], [17, 5], 4, 2, 128, False),           # tiny t2i denoise: <start> 16 latents <end> on top of a text context
    ([1, 1], [40, 129], 4, 2, 128, True),            # decode step, two samples
    ([33], [200], 7, 1, 128, True),                  # prefill on top of a context, GQA 7
    ([300, 257], [64, 0], 4, 2, 64, False),          # D=64 path (tiny config, padded heads)
    ([9, 120], [3, 70], 2, 2, 64, True),
    ([1026], [32], 28, 4, 128, False),               # 7B head layout, 512^2-sized latent block
    # (sample, KV head) pair counts that do not fill the 8 XCDs: every pair is dealt out as 8 / gcd(pairs, 8) interleaved query-tile sets
    ([1400], [300], 8, 4, 128, True),                # 4 pairs -> 2 sets of the 6 query tiles; causal: heaviest tiles first
    ([700, 1100, 257], [64, 0, 500], 4, 4, 128, False),   # 12 pairs -> 2 sets, ragged tile counts (3 / 5 / 2)
    ([900], [10], 3, 1, 128, True),                  # 1 pair -> 8 sets, fewer query tiles (4) than sets
    ([600, 300, 513, 256, 1030], [0, 40, 0, 7, 0], 4, 2, 64, False),   # 10 pairs -> 4 sets, D = 64
]


@pytest.mark.parametrize("q_lens,ctx_lens,nq,nkv,D,causal", ATTN_CASES)
def test_attention_matches_flash_attn_definition(q_lens, ctx_lens, nq, nkv, D, causal):
    got, ref = run_attention(q_lens, ctx_lens, nq, nkv, D, causal)
    # P is rounded to bf16 before the PV MFMA (as flash-attn does): allow 2 ulp
    close(got, ref, ulps=2, rel_l2=6e-3, what=f"attn q={q_lens} ctx={ctx_lens} D={D} causal={causal}")


def test_attention_online_softmax_rescale_branch():
    """A spiked key late in the sequence forces the running-max rescale in the last tiles."""
    q_lens, ctx_lens = [70], [200]
    nq, nkv, D = 2, 1, 128
    M = 70
    q, k, v = rnd(M, nq, D, seed=11), rnd(M, nkv, D, seed=12), rnd(M, nkv, D, seed=13)
    kc, vc = rnd(200, nkv, D, seed=14), rnd(200, nkv, D, seed=15)
    k[60] = (q[5, 0].float() * 4).to(BF16)       # huge score for query 5 (and correlated ones) at new-key 60
    kc[150] = (q[9, 1].float() * 3).to(BF16)
    from oracle.bagel_oracle import attn_varlen
    ref = attn_varlen(q, torch.cat([kc, k]), torch.cat([vc, v]), torch.tensor([0, 70], dtype=torch.int32),
                      torch.tensor([0, 270], dtype=torch.int32), 70, 270, causal=False)
    o = ops()
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)  # noqa: E731
    vt = torch.zeros((D, 256), dtype=BF16, device=DEV)
    vtc = torch.zeros((D, 256), dtype=BF16, device=DEV)
    o.v_transpose(v.reshape(M, -1).to(DEV), vt, i32([0, 70]), i32([0]), 1, 70, nkv, D)
    o.v_transpose(vc.reshape(200, -1).to(DEV), vtc, i32([0, 200]), i32([0]), 1, 200, nkv, D)
    out = torch.empty((M, nq * D), dtype=BF16, device=DEV)
    o.attn_varlen(q.reshape(M, -1).to(DEV), k.reshape(M, -1).to(DEV), vt, out, i32([0, 70]), i32([0]), 1, 70, nq, nkv, D, False,
                  D ** -0.5, k_ctx=kc.reshape(200, -1).to(DEV), vt_ctx=vtc, cu_ctx=i32([0, 200]), vt_ctx_col=i32([0]))
    close(out.view(M, nq, D), ref, ulps=2, rel_l2=6e-3, what="attn rescale branch")


def test_v_transpose_exact():
    lens = [70, 1, 129]
    nkv, D = 2, 128
    M = sum(lens)
    v = rnd(M, nkv * D, seed=3)
    cu = [0, 70, 71, 200]
    col = [0, 128, 192]
    vt = torch.full((nkv * D, 512), 7.0, dtype=BF16, device=DEV)
    ops().v_transpose(v.to(DEV), vt, torch.tensor(cu, dtype=torch.int32, device=DEV), torch.tensor(col, dtype=torch.int32, device=DEV),
                      3, 129, nkv, D)
    vt = vt.cpu()
    for b in range(3):
        blk = v[cu[b]:cu[b + 1]].t()          # (nkv*D, len)
        assert torch.equal(vt[:, col[b]:col[b] + lens[b]], blk), f"sample {b}"
        pad_end = col[b] + (lens[b] + 63) // 64 * 64
        assert (vt[:, col[b] + lens[b]:pad_end] == 0).all(), "tail of the last 64-block must be zero-filled"


# ------------------------------------------------------------------------------------------------------------
# glue kernels
# ------------------------------------------------------------------------------------------------------------
def test_copy_rows_and_embedding_gather():
    table = rnd(50, 128, seed=1)
    ids = torch.tensor([3, 3, 49, 0, 17], dtype=torch.int32)
    dst_rows = torch.tensor([4, 0, 9, 2, 7], dtype=torch.int32)
    dst = torch.zeros((10, 128), dtype=BF16, device=DEV)
    ops().copy_rows(table.to(DEV), dst, 5, 128, src_rows=ids.to(DEV), dst_rows=dst_rows.to(DEV))
    ref = torch.zeros(10, 128, dtype=BF16)
    ref[dst_rows.long()] = table[ids.long()]
    assert torch.equal(dst.cpu(), ref)
    # strided source view (K rows inside a fused qkv buffer)
    big = rnd(6, 512, seed=2).to(DEV)
    out = torch.zeros((8, 128), dtype=BF16, device=DEV)
    ops().copy_rows(big[:, 256:384], out[2:], 6, 128)
    assert torch.equal(out[2:].cpu(), big[:, 256:384].cpu())


def test_f32_to_bf16_with_padding():
    x = torch.randn(33, 588)
    y = ops().f32_to_bf16(x.to(DEV), cols_padded=592).cpu()
    assert torch.equal(y[:, :588], x.to(BF16)) and (y[:, 588:] == 0).all()
    z = ops().f32_to_bf16(torch.randn(7, 64).to(DEV))
    assert z.shape == (7, 64)


def test_timestep_embedding_inputs():
    import math as m
    half = 128
    freqs = torch.exp(-m.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    for t in (1.0, 0.9473684430122375, 0.5, 0.0):
        out = torch.empty((1, 256), dtype=BF16, device=DEV)
        ops().timestep_sinusoid(t, freqs.to(DEV), out)
        args = torch.tensor([t])[:, None].float() * freqs[None]
        ref = torch.cat([torch.cos(args), torch.sin(args)], -1).to(BF16)
        close(out, ref, ulps=1, rel_l2=2e-3, what=f"sinusoid t={t}")


def test_flow_add_and_table_add():
    n, H = 40, 256
    seq = rnd(50, H, seed=1)
    rows = torch.randperm(50)[:n].to(torch.int32)
    temb, table = rnd(1, H, seed=2), rnd(64, H, seed=3)
    ids = torch.randint(0, 64, (n,))
    d = seq.to(DEV).clone()
    ops().flow_add(d, rows.to(DEV), temb.to(DEV), table.to(DEV), ids.to(DEV))
    ref = seq.clone()
    ref[rows.long()] = (seq[rows.long()] + temb) + table[ids]
    assert torch.equal(d.cpu(), ref)
    x = rnd(n, H, seed=4)
    d = x.to(DEV).clone()
    ops().add_table_rows(d, table.to(DEV), ids.to(DEV))
    assert torch.equal(d.cpu(), x + table[ids])


def ref_cfg(v, vct, vci, s_t, s_i, mn, mode):
    """bagel.py:873-905 verbatim on CPU bf16 tensors."""
    if mode == "text_channel":
        v_text_ = vct + s_t * (v - vct)
        scale = (torch.norm(v, dim=-1, keepdim=True) / (torch.norm(v_text_, dim=-1, keepdim=True) + 1e-8)).clamp(min=mn, max=1.0)
        v_text = v_text_ * scale
        return vci + s_i * (v_text - vci) if vci is not None else v_text
    v_text_ = vct + s_t * (v - vct)
    v_ = vci + s_i * (v_text_ - vci) if vci is not None else v_text_
    if mode == "global":
        n0, n1 = torch.norm(v), torch.norm(v_)
    else:
        n0, n1 = torch.norm(v, dim=-1, keepdim=True), torch.norm(v_, dim=-1, keepdim=True)
    return v_ * (n0 / (n1 + 1e-8)).clamp(min=mn, max=1.0)


@pytest.mark.parametrize("mode", ["global", "channel", "text_channel"])
@pytest.mark.parametrize("with_img", [False, True])
def test_cfg_renorm_euler(mode, with_img):
    n, cols = 4096 + 37, 64
    v, vct, vci = rnd(n, cols, seed=1), rnd(n, cols, seed=2), (rnd(n, cols, seed=3) if with_img else None)
    x = torch.randn(n, cols, generator=torch.Generator().manual_seed(4))
    s_t, s_i, mn, dt = 4.0, 2.0, 0.3, 0.10000002
    o = ops()
    tmp = torch.empty((n, cols), dtype=BF16, device=DEV)
    partials = torch.zeros(512, dtype=torch.float32, device=DEV)
    xd = x.to(DEV).clone()
    nparts = o.cfg_stage1(v.to(DEV), vct.to(DEV), vci.to(DEV) if with_img else None, tmp, partials, s_t, s_i, mn, o.RENORM_MODES[mode])
    o.cfg_stage2_euler(xd, tmp, partials, nparts, mn, dt, use_global_scale=(mode == "global"))
    vt = ref_cfg(v, vct, vci, s_t, s_i, mn, mode)
    ref = x - vt * torch.tensor(dt)
    # the global norm is an fp32 sum in a different order -> the bf16 scale may flip by one ulp in rare cases
    close(xd, ref, ulps=1.5, rel_l2=3e-3, what=f"cfg {mode} img={with_img}")
    # no-CFG Euler step is exact
    xd = x.to(DEV).clone()
    o.cfg_stage2_euler(xd, v.to(DEV), None, 0, 0.0, dt, use_global_scale=False)
    assert torch.equal(xd.cpu(), x - v * torch.tensor(dt))


def test_argmax_first_max_wins():
    x = rnd(3, 1000, seed=1)
    x[1, 700] = x[1, 20] = 50.0
    got = ops().argmax(x.to(DEV)).cpu()
    assert torch.equal(got, torch.argmax(x.float(), -1))
    assert got[1].item() == 20


# ------------------------------------------------------------------------------------------------------------
# VAE kernels (fp32): tolerance = fp32 accumulation-order noise
# ------------------------------------------------------------------------------------------------------------
def close32(got, ref, tol=2e-5, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
    assert err <= tol, f"{what}: max rel-to-max error {err:.3g} > {tol}"


def _vae_engine():
    from bagel_amd.modeling.autoencoder import AutoEncoder, AutoEncoderParams
    from bagel_amd.modeling.vae_engine import VaeEngine
    from oracle.configs import TINY
    ae = AutoEncoder(AutoEncoderParams(**TINY["vae"])).to(DEV)
    from oracle.weights import load_synth
    load_synth(ae, 0)
    return VaeEngine(ae), ae


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("cin,cout,H,W", [(32, 64, 9, 13), (64, 3, 16, 16), (128, 160, 6, 10)])
def test_conv3x3_modes(mode, cin, cout, H, W):
    from torch import nn
    eng, _ = _vae_engine()
    g = torch.Generator().manual_seed(1)
    if mode == 2:
        H, W = H + (H % 2), W + (W % 2)
    x = torch.randn(2, cin, H, W, generator=g)
    m = nn.Module()
    m.weight = nn.Parameter(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5)
    m.bias = nn.Parameter(torch.randn(cout, generator=g) * 0.1)
    if mode == 1:
        ref = F.conv2d(x, m.weight, m.bias, padding=1)
    elif mode == 2:
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), m.weight, m.bias, stride=2)
    else:
        ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), m.weight, m.bias, padding=1)
    res = torch.randn(ref.shape, generator=g)
    m = m.to(DEV)
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = eng.conv(xh, m, mode)
    close32(out.permute(0, 3, 1, 2), ref, what=f"conv mode {mode}")
    if cout % 4 == 0:
        out = eng.conv(xh, m, mode, residual=res.permute(0, 2, 3, 1).contiguous().to(DEV))
        close32(out.permute(0, 3, 1, 2), ref + res, what=f"conv mode {mode} + residual")


def test_conv1x1_and_nt_gemm():
    eng, _ = _vae_engine()
    g = torch.Generator().manual_seed(2)
    a, b = torch.randn(300, 132, generator=g), torch.randn(76, 132, generator=g)
    close32(eng.gemm_nt(a.to(DEV), b.to(DEV)), a @ b.t(), what="nt gemm")


@pytest.mark.parametrize("C,swish", [(32, True), (64, False), (128, True)])
def test_groupnorm_swish(C, swish):
    eng, _ = _vae_engine()
    from torch import nn
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, C, 12, 20, generator=g) * 3 + 5.0     # large mean: the shifted accumulation must not cancel
    m = nn.Module()
    m.weight = nn.Parameter(1 + 0.1 * torch.randn(C, generator=g))
    m.bias = nn.Parameter(0.1 * torch.randn(C, generator=g))
    ref = F.group_norm(x, 32, m.weight, m.bias, 1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    out = eng.gn(x.permute(0, 2, 3, 1).contiguous().to(DEV), m.to(DEV), swish)
    close32(out.permute(0, 3, 1, 2), ref, tol=3e-5, what="groupnorm")


def test_softmax_rows():
    from bagel_amd._lib import lib, check
    g = torch.Generator().manual_seed(4)
    x = torch.randn(37, 1000, generator=g) * 4
    d = x.to(DEV).clone()
    check(lib().bagel_softmax_rows_f32(d.data_ptr(), d.stride(0), 37, 1000, 0.25, torch.cuda.current_stream().cuda_stream))
    close32(d, torch.softmax(x * 0.25, -1), what="softmax rows")
