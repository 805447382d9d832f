"""The four gen-expert projections of a stream-batched denoise forward (M = 2 x 4 x 4096 latent rows) in bf16 (persistent ping-pong
kernel) and in FP8 (same kernel, e4m3 operands), plus the activation quantiser.  python tools/gemm_fp8_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot = {"bf16": 0.0, "fp8": 0.0, "quant": 0.0}
for name, N, K, mode in (("qkv", 4608, 3584, "bias"), ("o", 3584, 3584, "residual"), ("gate_up", 37888, 3584, "swiglu"), ("down", 3584, 18944, "residual")):
    A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.zeros(N, dtype=torch.bfloat16, device=dev) if mode == "bias" else None
    Nout = N // 2 if mode == "swiglu" else N
    C = torch.zeros((M, Nout), dtype=torch.bfloat16, device=dev)
    epi = ops.EPI_SWIGLU16 if mode == "swiglu" else ops.EPI_NONE
    res = C if mode == "residual" else None
    t16 = timeit(lambda: ops.gemm(A, W, C, bias0=bias, residual=res, epilogue=epi, variant=4))
    qw, sw = ops.quantize_rows_fp8(W)
    qa, sa = ops.quantize_rows_fp8(A)
    tq = timeit(lambda: ops.quantize_rows_fp8(A, qa, sa))
    t8 = timeit(lambda: ops.gemm_fp8(qa, sa, qw, sw, C, bias=bias, residual=res, epilogue=epi))
    fl = 2.0 * M * N * K
    print(f"{name:8s} M={M} N={N} K={K}: bf16 {t16:.3f} ms {fl / t16 / 1e9:6.0f} TFLOP/s | fp8 {t8:.3f} ms {fl / t8 / 1e9:6.0f} TFLOP/s "
          f"({t16 / t8:.2f}x) | quantise A {tq * 1e3:.0f} us ({3.0 * M * K / tq / 1e6:.0f} GB/s)", flush=True)
    tot["bf16"] += t16; tot["fp8"] += t8; tot["quant"] += tq
print(f"layer total: bf16 {tot['bf16']:.3f} ms, fp8 {tot['fp8']:.3f} ms + activation quantisers {tot['quant']:.3f} ms "
      f"-> {tot['bf16'] / (tot['fp8'] + tot['quant']):.2f}x", flush=True)
