"""bagel_gemm_bf16_ws vs bagel_gemm_bf16 at the LLM-prefill shapes of the understanding request (M = 4936): one pass vs full rounds + K-split
leftovers + reduce, same tensors, same box.  python tools/gemm_splitk_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

BF16, DEV = torch.bfloat16, "cuda"


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    H, I = 3584, 18944
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV).to(BF16)  # noqa: E731
    Ms = [int(x) for x in os.environ["PROBE_M"].split(",")] if os.environ.get("PROBE_M") else (4936, 2050, 12294, 16392, 4902, 9066)
    for M in Ms:      # understanding prefill, a short prompt, the 3-stream edit forward, one denoise stream, the ViT block, the edit context
        tot = [0.0, 0.0]
        for name, N, K, bias, resid in (("qkv", 4608, H, True, False), ("o", H, H, False, True), ("down", H, I, False, True)):
            A, W = rn(M, K), rn(N, K)
            b = rn(N) if bias else None
            C = rn(M, N)
            t = []
            for sk in (False, True):
                t.append(timeit(lambda: ops.gemm(A, W, C, bias0=b, residual=C if resid else None, variant=5, splitk=sk)))
            fl = 2.0 * M * N * K
            tot[0] += t[0]; tot[1] += t[1]
            print(f"M={M} {name:5s} one pass {t[0]:7.1f} us {fl / t[0] / 1e6:7.1f} TF | k-split leftovers {t[1]:7.1f} us {fl / t[1] / 1e6:7.1f} TF  ({t[0] / t[1]:.3f}x)", flush=True)
        print(f"(BAGEL_GEMM_SPLIT_POLICY={os.environ.get('BAGEL_GEMM_SPLIT_POLICY', '0')} FORCE={os.environ.get('BAGEL_GEMM_SPLIT_FORCE', '-')}) M={M} qkv + o + down per layer: {tot[0]:.0f} -> {tot[1]:.0f} us ({(tot[0] - tot[1]) * 28 / 1e3:.1f} ms over 28 layers)")


if __name__ == "__main__":
    main()
