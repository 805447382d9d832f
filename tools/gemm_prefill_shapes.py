"""Which tile shape for the prefill GEMMs?  M ~ 4 900 rows (980^2 ViT image, LLM prefill of its tokens) leaves the 256x256 persistent
kernel with 100-340 tiles on 256 CUs; variants: 0 = 128x128, 2 = 256x128, 4 = persistent 256x256 ping-pong.  python tools/gemm_prefill_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, M, N, K, mode in (("vit qkv", 4900, 3456, 1152, "bias"), ("vit out", 4900, 1152, 1152, "bias"), ("vit fc1", 4900, 4304, 1152, "gelu"),
                            ("vit fc2", 4900, 1152, 4304, "bias"),
                            # round 6: the MLP width padded to the 64-deep k-tile (siglip_navit.py _pack): fc1 writes 4352 columns (48 exact zeros), fc2 contracts over them
                            ("vit fc1 pad", 4900, 4352, 1152, "gelu"), ("vit fc2 pad", 4900, 1152, 4352, "bias"),
                            # ... as the model calls them: bias AND residual in one epilogue (not instantiated in the persistent kernel: variant 4 falls back to 3)
                            ("vit out +res", 4900, 1152, 1152, "bias+residual"), ("vit fc2 pad +res", 4900, 1152, 4352, "bias+residual"),
                            # the packed tower's real shapes: heads padded 72 -> 128 (qkv writes 3 x 16 x 128 columns, the out projection contracts over 16 x 128)
                            ("vit qkv padded", 4900, 6144, 1152, "bias"), ("vit out padded +res", 4900, 1152, 2048, "bias+residual"),
                            ("llm qkv", 4902, 4608, 3584, "bias"), ("llm o", 4902, 3584, 3584, "residual"),
                            ("llm gate_up", 4902, 37888, 3584, "swiglu"), ("llm down", 4902, 3584, 18944, "residual"),
                            ("edit o", 12288, 3584, 3584, "residual"), ("edit down", 12288, 3584, 18944, "residual")):
    A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.zeros(N, dtype=torch.bfloat16, device=dev) if mode in ("bias", "gelu", "bias+residual") else None
    Nout = N // 2 if mode == "swiglu" else N
    C = torch.zeros((M, Nout), dtype=torch.bfloat16, device=dev)
    epi = {"swiglu": ops.EPI_SWIGLU16, "gelu": ops.EPI_GELU_TANH}.get(mode, ops.EPI_NONE)
    res = C if mode in ("residual", "bias+residual") else None
    out = []
    for v in (0, 3, 4, 5, None):        # None = what ops.gemm picks; 3 = one tile per workgroup; 4 = persistent; 5 = 4 with SGPR-base DMA (+ the ViT epilogues, round 6)
        try:
            t = timeit(lambda: ops.gemm(A, W, C, bias0=bias, residual=res, epilogue=epi, variant=v))
            out.append(f"{'auto' if v is None else 'v' + str(v)} {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF")
        except Exception as e:  # noqa: BLE001
            out.append(f"v{v} n/a ({str(e)[:30]})")
    print(f"{name:12s} M={M} N={N} K={K}: " + " | ".join(out), flush=True)
