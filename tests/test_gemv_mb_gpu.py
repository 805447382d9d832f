"""GPU parity of the batched-decode projection (csrc/gemv_mb.hip, bagel_gemv_mb_bf16) through the C ABI: C[M <= 32, N] =
norm(A) W^T with gemm.hip's epilogues, against fp32 torch (tests/test_ops_gpu.py conventions: <= 1 bf16 ulp of the tensor's
magnitude per rounding, rel-L2 ~1e-3) -- at the shapes a 7B decode step launches (qkv 4608 x 3584 with bias + fused RMSNorm, o 3584 x
3584 with the in-place residual, gate+up 37888 x 3584 with SwiGLU16 + fused RMSNorm, down 3584 x 18944 through four K slices, lm_head with
the final norm) and at the edges of its geometry (K = 128 / 448 / 1024 / 1056 / 4864: the three instantiations and their ragged last wave;
M = 1, 2, 5, 16 on one block of request rows, 17, 24, 32 on two; fewer column blocks than workgroups; more than one 8-block chunk per workgroup)."""
import pytest
import torch

from tests.test_ops_gpu import BF16, DEV, close, ops, ref_gemm, rnd

pytestmark = pytest.mark.gpu


def ref_rmsnorm(x, w, eps):
    h = x.float()
    h = (h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)).to(BF16)
    return (w.float() * h.float()).to(BF16)


def run(M, N, K, *, bias=False, resid=False, epi=0, norm=False, seed=0):
    A = rnd(M, K, seed=seed + 1)
    W = rnd(N, K, seed=seed + 2, scale=K ** -0.5)
    b = rnd(N, seed=seed + 3, scale=0.1) if bias else None
    nw = (1.0 + 0.1 * rnd(K, seed=seed + 4).float()).to(BF16) if norm else None
    Nout = N // 2 if epi == 3 else N
    R = rnd(M, Nout, seed=seed + 5) if resid else None
    C = R.to(DEV).clone() if resid else torch.full((M, Nout), float("nan"), dtype=BF16, device=DEV)
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    assert ops().gemv_mb_supported(A.to(DEV), W.to(DEV), C, d(b), C if resid else None, epi, norm, M=M), "shape not served by gemv_mb"
    ops().gemv_mb(A.to(DEV), W.to(DEV), C, bias=d(b), residual=C if resid else None, epilogue=epi, norm_w=d(nw), eps=1e-6, M=M)
    torch.cuda.synchronize()
    Ain = ref_rmsnorm(A, nw, 1e-6) if norm else A
    ref = ref_gemm(Ain, W, b, epi, R)
    return C, ref


@pytest.mark.parametrize("M", [2, 5, 16, 17, 24, 32])
@pytest.mark.parametrize("proj", ["qkv", "o", "gate_up", "down", "lm_head"])
def test_gemv_mb_at_7b_decode_shapes(proj, M):
    H, I = 3584, 18944
    N, K, kw = {"qkv": (4608, H, dict(bias=True, norm=True)), "o": (H, H, dict(resid=True)),
                "gate_up": (2 * I, H, dict(epi=3, norm=True)), "down": (H, I, dict(resid=True)),
                "lm_head": (152064, H, dict(norm=True))}[proj]
    C, ref = run(M, N, K, seed=M, **kw)
    close(C, ref, ulps=2, what=f"gemv_mb {proj} M={M}")


@pytest.mark.parametrize("M", [1, 3, 16, 17, 32])
@pytest.mark.parametrize("N,K", [(64, 128), (48, 448), (272, 1024), (528, 1056), (160, 4864), (4112, 512), (16, 3584), (528, 2560)])
def test_gemv_mb_geometry_edges(M, N, K):
    """NS = 4 (K <= 1024), 10 (K = 2560), 14 (K <= 3584), 19 (K = 4864 in one slice): ragged last waves, rows shorter than 8 waves x their steps, fewer blocks
    than workgroups (N = 16: one), several chunks per workgroup (N = 4112 on 256 workgroups is one block each; see the 7B shapes for > 8).  M > 16 (two
    blocks of request rows): K = 4864 takes two K slices there (the longest two-block instantiation is 14 steps), so no fused norm / SwiGLU at that length."""
    one_slice = M <= 16 or K <= 3584
    for kw in (dict(bias=True), dict(resid=True), dict(bias=True, norm=True), dict(epi=1, bias=True), dict(epi=2)):
        if kw.get("norm") and not one_slice:
            A, W = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2).to(DEV)
            assert not ops().gemv_mb_supported(A, W, torch.empty((M, N), dtype=BF16, device=DEV), None, None, 0, True, M=M)
            continue
        C, ref = run(M, N, K, seed=N + K, **kw)
        close(C, ref, ulps=2, what=f"gemv_mb {M}x{N}x{K} {kw}")
    if N % 32 == 0 and one_slice:
        C, ref = run(M, N, K, epi=3, norm=True, seed=7)
        close(C, ref, ulps=2, what=f"gemv_mb swiglu {M}x{N}x{K}")


@pytest.mark.parametrize("M", [16, 32])
def test_gemv_mb_k_slices_are_deterministic_and_match_the_tile_kernel(M):
    """K = 18944 runs as K slices over workgroups + the reduce launch: same bits on every call (fixed summation order, no atomics), within an ulp of the
    128x128 tile kernel on the same operands, and a NaN-filled workspace from an earlier, larger call cannot leak into the result."""
    N, K = 3584, 18944
    A, W, R = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2, scale=K ** -0.5).to(DEV), rnd(M, N, seed=3).to(DEV)
    ws = torch.full((ops().mb_workspace_floats(N, K) + 1024,), float("nan"), dtype=torch.float32, device=DEV)
    outs = []
    for _ in range(3):
        C = R.clone()
        ops().gemv_mb(A, W, C, residual=C, M=M, workspace=ws)
        outs.append(C)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    C2 = R.clone()
    ops().gemm(A, W, C2, residual=C2, variant=0)
    close(outs[0], C2, ulps=2, what="gemv_mb K slices vs the 128x128 tile kernel")
    with pytest.raises(ops().BagelHipError):
        ops().gemv_mb(A, W, R.clone(), M=M, workspace=torch.empty(16, dtype=torch.float32, device=DEV))     # too small a workspace is refused


def test_gemv_mb_refuses_what_it_does_not_build():
    o = ops()
    A, W = rnd(4, 18944, seed=1).to(DEV), rnd(64, 18944, seed=2).to(DEV)
    C = torch.empty((4, 64), dtype=BF16, device=DEV)
    assert not o.gemv_mb_supported(A, W, C, None, None, 0, True)            # fused norm needs the whole row in one slice
    with pytest.raises(o.BagelHipError):
        o.gemv_mb(A, W, C, norm_w=rnd(18944, seed=3).to(DEV), eps=1e-6)
    with pytest.raises(o.BagelHipError):
        o.gemv_mb(rnd(33, 128, seed=1).to(DEV), rnd(64, 128, seed=2).to(DEV), torch.empty((33, 64), dtype=BF16, device=DEV))   # M > 32
    # two blocks of request rows: a row of more than 8 x 14 steps runs as slices, so the fused norm stops at K = 3584 there
    A, W = rnd(20, 4864, seed=1).to(DEV), rnd(64, 4864, seed=2).to(DEV)
    assert o.gemv_mb_supported(A[:16], W, torch.empty((16, 64), dtype=BF16, device=DEV), None, None, 0, True, M=16)
    assert not o.gemv_mb_supported(A, W, torch.empty((20, 64), dtype=BF16, device=DEV), None, None, 0, True, M=20)
    with pytest.raises(o.BagelHipError):
        o.gemv_mb(A, W, torch.empty((20, 64), dtype=BF16, device=DEV), norm_w=rnd(4864, seed=3).to(DEV), eps=1e-6)


@pytest.mark.parametrize("M", [16, 32])
def test_ops_gemm_routes_2_to_32_rows_to_gemv_mb(monkeypatch, M):
    o = ops()
    called = []
    real = o.gemv_mb
    monkeypatch.setattr(o, "gemv_mb", lambda *a, **k: (called.append(1), real(*a, **k))[1])
    A, W = rnd(M, 3584, seed=1), rnd(4608, 3584, seed=2, scale=3584 ** -0.5)
    C = torch.empty((M, 4608), dtype=BF16, device=DEV)
    o.gemm(A.to(DEV), W.to(DEV), C)
    assert called
    close(C, ref_gemm(A, W), what="ops.gemm -> gemv_mb")


def test_gemv_mb_rows_do_not_depend_on_their_neighbours():
    """A request row's result is a function of that row alone: the first 16 rows of a 32-row call equal the same rows of a 20-row call bit for bit (same
    two-block geometry), and rows 0..15 of a 16-row call equal those of a 3-row call (one block)."""
    o = ops()
    N, K = 4608, 3584
    A, W = rnd(32, K, seed=5).to(DEV), rnd(N, K, seed=6, scale=K ** -0.5).to(DEV)
    nw = (1.0 + 0.1 * rnd(K, seed=7).float()).to(BF16).to(DEV)

    def call(M):
        C = torch.empty((M, N), dtype=BF16, device=DEV)
        o.gemv_mb(A[:M].contiguous(), W, C, norm_w=nw, eps=1e-6, M=M)
        return C
    c32, c20, c16, c3 = call(32), call(20), call(16), call(3)
    torch.cuda.synchronize()
    assert torch.equal(c32[:20], c20) and torch.equal(c16[:3], c3)
