"""Stand-alone timing of the batched-decode projection (csrc/gemv_mb.hip) against what it replaces (rmsnorm + skinny.hip) and against the
one-request gemv, at the shapes of a 7B decode step.  Every shape cycles through COPIES different weight buffers so that no launch finds its
weights in the 256 MB Infinity Cache (the decode step streams 14 GB between two uses of a matrix).
    python tools/gemv_mb_bench.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

DEV, BF16 = "cuda", torch.bfloat16
H, I, V = 3584, 18944, 152064
SHAPES = [("qkv", 4608, H, dict(bias=True, norm=True)), ("o", H, H, dict(resid=True)), ("gate_up", 2 * I, H, dict(epi=3, norm=True)),
          ("down", H, I, dict(resid=True)), ("lm_head", V, H, dict(norm=True))]


def timeit(fn, n=20):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16]
    g = torch.Generator(device=DEV).manual_seed(0)
    tot = {}
    for name, N, K, kw in SHAPES:
        copies = max(2, int(1.2e9 // (N * K * 2)))
        Ws = [(torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).to(BF16) for _ in range(copies)]
        bias = (torch.randn(N, device=DEV, generator=g) * 0.1).to(BF16) if kw.get("bias") else None
        nw = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).to(BF16) if kw.get("norm") else None
        epi = kw.get("epi", 0)
        Nout = N // 2 if epi == 3 else N
        mb = N * K * 2 / 1e6
        line = f"{name:8s} {mb:7.1f} MB |"
        # one request: the lane-FMA gemv with the fused norm
        A1 = torch.randn(1, K, device=DEV, generator=g).to(BF16)
        C1 = torch.zeros(1, Nout, device=DEV, dtype=BF16)
        t1 = timeit(lambda i: ops.gemv(A1, Ws[i % copies], C1, bias=bias, residual=C1 if kw.get("resid") else None, epilogue=epi, norm_w=nw, eps=1e-6))
        line += f" B=1 gemv {t1:6.1f} us {mb / t1:5.2f} TB/s |"
        tot.setdefault(1, [0.0, 0.0])
        tot[1][0] += t1 * (1 if name == "lm_head" else 28)
        for B in Bs:
            A = torch.randn(B, K, device=DEV, generator=g).to(BF16)
            C = torch.zeros(B, Nout, device=DEV, dtype=BF16)
            h = torch.empty_like(A)

            def old(i):
                x = A
                if nw is not None:
                    ops.rmsnorm(A, nw, h, 1e-6)
                    x = h
                ops.gemm_skinny(x, Ws[i % copies], C, bias=bias, residual=C if kw.get("resid") else None, epilogue=epi, M=B)
            t_old = timeit(old)
            t_new = timeit(lambda i: ops.gemv_mb(A, Ws[i % copies], C, bias=bias, residual=C if kw.get("resid") else None, epilogue=epi, norm_w=nw, eps=1e-6, M=B))
            line += f" B={B}: skinny {t_old:6.1f} -> mb {t_new:6.1f} us ({mb / t_new:4.2f} TB/s)"
            if B <= ops.GEMV_MAX_ROWS and K * 2 <= ops.GEMV_MAX_K_BYTES:        # the lane-FMA gemv at a few rows
                t_fma = timeit(lambda i: ops.gemv(A, Ws[i % copies], C, bias=bias, residual=C if kw.get("resid") else None, epilogue=epi, norm_w=nw, eps=1e-6, M=B))
                line += f" lane-FMA {t_fma:6.1f}"
                tot.setdefault(("fma", B), [0.0, 0.0])
                tot[("fma", B)][1] += t_fma * (1 if name == "lm_head" else 28)
            line += " |"
            tot.setdefault(B, [0.0, 0.0])
            tot[B][0] += t_old * (1 if name == "lm_head" else 28)
            tot[B][1] += t_new * (1 if name == "lm_head" else 28)
        print(line, flush=True)
        del Ws
        torch.cuda.empty_cache()
    for B, (a, b) in sorted(tot.items(), key=lambda kv: str(kv[0])):
        if isinstance(B, tuple):
            print(f"projections of one 28-layer step, B={B[1]} on the lane-FMA gemv: {b / 1e3:.3f} ms")
        else:
            print(f"projections of one 28-layer step, B={B}: {a / 1e3:.3f} ms" + (f" -> {b / 1e3:.3f} ms" if B > 1 else " (gemv)"))


if __name__ == "__main__":
    main()
