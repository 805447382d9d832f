"""The four denoise GEMM shapes of BASELINE configs[2] (M = 16392 packed rows): this repo's hand-written kernel beside the
vendor library (torch.matmul -> hipBLASLt/rocBLAS) on the same tensors.  The library is a yardstick only -- the product
never calls it.  python tools/gemm_vs_library.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = 16392
SHAPES = [("qkv", 4608, 3584), ("o_proj", 3584, 3584), ("gate+up", 37888, 3584), ("down", 3584, 18944)]
g = torch.Generator(device="cuda").manual_seed(0)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot_o = tot_l = 0.0
for name, N, K in SHAPES:
    A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
    C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    C2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    Wt = W.t()
    ours = timed(lambda: ops.gemm(A, W, C))
    lib = timed(lambda: torch.matmul(A, Wt, out=C2))
    fl = 2.0 * M * N * K
    err = ((C.float() - C2.float()).norm() / C2.float().norm()).item()
    tot_o += ours
    tot_l += lib
    print(f"{name:8s} M={M} N={N} K={K}: ours {ours:.3f} ms {fl / ours / 1e9:7.1f} TFLOP/s | library {lib:.3f} ms {fl / lib / 1e9:7.1f} TFLOP/s | "
          f"ours/library time {ours / lib:.3f} | rel-L2 between them {err:.2e}", flush=True)
print(f"layer total (4 GEMMs): ours {tot_o:.3f} ms, library {tot_l:.3f} ms, ratio {tot_o / tot_l:.3f}", flush=True)
