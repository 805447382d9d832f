"""Bitwise cross-check of the persistent ping-pong GEMM (variant 4) against the one-tile-per-workgroup ping-pong kernel
(variant 3), and of its SGPR-base-DMA form (variant 5): same MFMA order, same rounding points, so every output must be IDENTICAL.  Run with BAGEL_GEMM_PERSIST_WGS=8
to make every workgroup walk several tiles even on small problems (tests/test_ops_gpu.py does), and at the default grid.
python tools/gemm_persist_check.py [--bench]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

DEV, BF16 = "cuda", torch.bfloat16
TOLERANCE = "--tolerance" in sys.argv


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF16).to(DEV)


def it32(x):
    return torch.tensor(x, dtype=torch.int32, device=DEV)


def both(what, call, out_shape, init=None, repeats=3, same45=True):
    outs = []
    for variant in (3, 4, 5, 4, 5)[:2 + repeats]:        # 5 = variant 4 with SGPR-base DMA addresses: same arithmetic
        C = init.clone() if init is not None else torch.full(out_shape, float("nan"), dtype=BF16, device=DEV)
        call(C, variant)
        torch.cuda.synchronize()
        outs.append(C)
    if TOLERANCE:
        # K-split schedule (bagel_gemm_bf16_ws): the leftover tiles are summed in another order -- one bf16 ulp of the tensor's magnitude against
        # the one-tile kernel (itself pinned to fp32 in tests/test_ops_gpu.py), identical between repeats
        ref = outs[0].float()
        tol = 2 ** -7 * ref.abs().max().item()
        # repeats of one variant are bit-identical; variants 4 and 5 too where they run the same schedule (not for the ViT epilogues: variant 4 takes the
        # one-tile kernel for them, variant 5 the persistent kernel with its K-split)
        ok = all((o.float() - ref).abs().max().item() <= tol for o in outs[1:]) and all(torch.equal(outs[i], outs[i + 2]) for i in range(1, len(outs) - 2))
        ok = ok and (not same45 or torch.equal(outs[1], outs[2]))
        ok = ok and bool(torch.isfinite(outs[1].float()).all())
        differ = float((outs[1].float() != ref).float().mean())
        print(f"{'ok ' if ok else 'BAD'} {what}   (elements that differ from the one-pass result: {differ:.3f})", flush=True)
        return ok
    ok = all(torch.equal(outs[0].view(torch.int16), o.view(torch.int16)) for o in outs[1:]) and bool(torch.isfinite(outs[0].float()).all())
    print(f"{'ok ' if ok else 'BAD'} {what}", flush=True)
    return ok


def main():
    good = True
    # dense, no epilogue operands (mode 3); partial tiles in M and N
    for M, N, K in [(1500, 1280, 2048), (777, 392, 512), (100, 104, 128), (2050, 264, 192)]:
        A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
        good &= both(f"plain {M}x{N}x{K}", lambda C, v: ops.gemm(A, W, C, variant=v), (M, N))
    # bias (mode 2)
    for M, N, K in [(1500, 1280, 256), (777, 392, 512)]:
        A, W, b = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=K ** -0.5), rnd(N, seed=5, scale=0.1)
        good &= both(f"bias {M}x{N}x{K}", lambda C, v: ops.gemm(A, W, C, bias0=b, variant=v), (M, N))
    # residual, in place (mode 1)
    for M, N, K in [(1000, 768, 320), (1530, 520, 1024)]:
        A, W, R = rnd(M, K, seed=6), rnd(N, K, seed=7, scale=K ** -0.5), rnd(M, N, seed=8)
        good &= both(f"residual in place {M}x{N}x{K}", lambda C, v: ops.gemm(A, W, C, residual=C, variant=v), (M, N), init=R)
    # the ViT's epilogues, SGPR-base form only (variant 4 takes the one-tile kernel for them): bias + residual in place (mode 5), bias + GELU-tanh (mode 6)
    for M, N, K in [(1000, 768, 320), (1530, 520, 1024), (700, 1152, 4352)]:
        A, W, b, R = rnd(M, K, seed=19), rnd(N, K, seed=20, scale=K ** -0.5), rnd(N, seed=21, scale=0.1), rnd(M, N, seed=22)
        good &= both(f"bias + residual in place {M}x{N}x{K}", lambda C, v: ops.gemm(A, W, C, bias0=b, residual=C, variant=v), (M, N), init=R, same45=False)
    for M, N, K in [(1500, 1280, 256), (777, 392, 512), (600, 4352, 1152)]:
        A, W, b = rnd(M, K, seed=23), rnd(N, K, seed=24, scale=K ** -0.5), rnd(N, seed=25, scale=0.1)
        good &= both(f"bias + gelu {M}x{N}x{K}", lambda C, v: ops.gemm(A, W, C, bias0=b, epilogue=ops.EPI_GELU_TANH, variant=v), (M, N), same45=False)
    # SwiGLU pairing (mode 0)
    for M, N, K in [(1333, 832, 256), (600, 1536, 512)]:
        A, W = rnd(M, K, seed=9), rnd(N, K, seed=10, scale=K ** -0.5)
        good &= both(f"swiglu {M}x{N}x{K}", lambda C, v: ops.gemm(A, W, C, epilogue=ops.EPI_SWIGLU16, variant=v), (M, N // 2))
    # MoT routing: two row groups with gather/scatter lists, rows interleaved like <start> latents <end> per sample
    lens = [300, 700, 260]
    rows_t, rows_v, base = [], [], 0
    for n in lens:
        rows_t += [base, base + n + 1]
        rows_v += list(range(base + 1, base + n + 1))
        base += n + 2
    Mtot, H = base, 512
    rt, rv_ = it32(rows_t), it32(rows_v)
    A = rnd(Mtot, H, seed=11)
    for what, N, kw in [("bias", 640, dict(bias=True)), ("residual", 512, dict(res=True)), ("swiglu", 768, dict(epi=ops.EPI_SWIGLU16)), ("plain", 392, {})]:
        W0, W1 = rnd(N, H, seed=12, scale=H ** -0.5), rnd(N, H, seed=13, scale=H ** -0.5)
        b0, b1 = (rnd(N, seed=14, scale=0.1), rnd(N, seed=15, scale=0.1)) if kw.get("bias") else (None, None)
        outN = N // 2 if kw.get("epi") else N
        R = rnd(Mtot, outN, seed=16)

        def call(C, v, W0=W0, W1=W1, b0=b0, b1=b1, kw=kw):
            ops.gemm(A, W0, C, bias0=b0, a_rows0=rt, c_rows0=rt, M0=len(rows_t), W1=W1, bias1=b1, a_rows1=rv_, c_rows1=rv_, M1=len(rows_v),
                     residual=C if kw.get("res") else None, epilogue=kw.get("epi", ops.EPI_NONE), variant=v)
        good &= both(f"two expert groups, {what}, N={N}", call, (Mtot, outN), init=R)
    # gather rows, dense output (llm2vae)
    rows = list(range(1, 600)) + list(range(700, 1400))
    W, b = rnd(64, H, seed=17, scale=H ** -0.5), rnd(64, seed=18, scale=0.1)
    good &= both("gather rows -> dense out", lambda C, v: ops.gemm(A, W, C, bias0=b, a_rows0=it32(rows), M0=len(rows), variant=v), (len(rows), 64))

    if "--bench" in sys.argv:
        # the four denoise GEMMs of BASELINE configs[2] with the model's routing (16384 latent rows + 8 marker rows) and epilogues
        B, L = 4, 4096
        rows_t, rows_v = [], []
        for s in range(B):
            rows_t += [s * (L + 2), s * (L + 2) + L + 1]
            rows_v += list(range(s * (L + 2) + 1, s * (L + 2) + L + 1))
        rt, rv_ = it32(rows_t), it32(rows_v)
        M = B * (L + 2)
        tot = {3: 0.0, 4: 0.0}
        for name, N, K, kw in [("qkv", 4608, 3584, dict(bias=True)), ("o_proj", 3584, 3584, dict(res=True)),
                               ("gate+up", 37888, 3584, dict(epi=ops.EPI_SWIGLU16)), ("down", 3584, 18944, dict(res=True))]:
            A = rnd(M, K, seed=21)
            W0, W1 = rnd(N, K, seed=22, scale=K ** -0.5), rnd(N, K, seed=23, scale=K ** -0.5)
            b0, b1 = (rnd(N, seed=24, scale=0.1), rnd(N, seed=25, scale=0.1)) if kw.get("bias") else (None, None)
            outN = N // 2 if kw.get("epi") else N
            X = rnd(M, outN, seed=26)

            def call(C, v):
                ops.gemm(A, W0, C, bias0=b0, a_rows0=rt, c_rows0=rt, M0=len(rows_t), W1=W1, bias1=b1, a_rows1=rv_, c_rows1=rv_, M1=len(rows_v),
                         residual=C if kw.get("res") else None, epilogue=kw.get("epi", ops.EPI_NONE), variant=v)
            good &= both(f"denoise {name} M={M} N={N} K={K}", call, (M, outN), init=X, repeats=1)
            ms = {}
            for rep in range(2):
                for v in (3, 4):
                    C = X.clone()
                    for _ in range(2):
                        call(C, v)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        call(C, v)
                    e1.record()
                    torch.cuda.synchronize()
                    ms[v] = min(ms.get(v, 1e9), e0.elapsed_time(e1) / 10)
            fl = 2.0 * M * N * K
            tot[3] += ms[3]
            tot[4] += ms[4]
            print(f"  {name:8s} variant 3: {ms[3]:.3f} ms {fl / ms[3] / 1e9:7.1f} TFLOP/s | variant 4 (persistent): {ms[4]:.3f} ms {fl / ms[4] / 1e9:7.1f} TFLOP/s | "
                  f"time ratio {ms[4] / ms[3]:.3f}", flush=True)
        print(f"  layer total: variant 3 {tot[3]:.3f} ms, variant 4 {tot[4]:.3f} ms, ratio {tot[4] / tot[3]:.3f}", flush=True)
    print(("ALL WITHIN TOLERANCE" if TOLERANCE else "ALL IDENTICAL") if good else "MISMATCH", flush=True)
    sys.exit(0 if good else 1)


if __name__ == "__main__":
    main()
