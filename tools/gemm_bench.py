"""One GEMM shape, dense, repeated: python tools/gemm_bench.py M N K [variant] [epilogue] [iters].  For rocprofv3 --pmc."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 3
epi = int(sys.argv[5]) if len(sys.argv) > 5 else 0
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
C = torch.empty((M, N // 2 if epi == 3 else N), dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.gemm(A, W, C, epilogue=epi, variant=variant)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.gemm(A, W, C, epilogue=epi, variant=variant)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"gemm M={M} N={N} K={K} v{variant} epi{epi}: {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)
