"""Micro-benchmarks of the hot kernels at BAGEL-7B shapes (run on the GPU box through gpurun)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

BF16 = torch.bfloat16
DEV = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    res = {}
    B, n_img = 4, 4096
    M = B * (n_img + 2)
    H, I = 3584, 18944
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=DEV) * sc).to(BF16)  # noqa: E731
    x = rn(M, H)
    text_rows, vae_rows = [], []
    for b in range(B):
        base = b * (n_img + 2)
        text_rows += [base, base + n_img + 1]
        vae_rows += list(range(base + 1, base + n_img + 1))
    tr = torch.tensor(text_rows, dtype=torch.int32, device=DEV)
    vr = torch.tensor(vae_rows, dtype=torch.int32, device=DEV)
    shapes = [("qkv", 4608, H, 0), ("o_proj", H, H, 0), ("gate_up_swiglu", 2 * I, H, 3), ("down", H, I, 0)]
    for name, N, K, epi in shapes:
        W0, W1 = rn(N, K, sc=K ** -0.5), rn(N, K, sc=K ** -0.5)
        A = x if K == H else rn(M, K)
        C = torch.empty((M, N // 2 if epi == 3 else N), dtype=BF16, device=DEV)
        for variant in (1, 3):
            try:
                ms = timeit(lambda: ops.gemm(A, W0, C, a_rows0=tr, c_rows0=tr, M0=len(text_rows), W1=W1, a_rows1=vr, c_rows1=vr,
                                             M1=len(vae_rows), epilogue=epi, variant=variant))
                res[f"gemm_{name}_v{variant}"] = dict(ms=ms, tflops=2.0 * M * N * K / ms / 1e9)
                ms = timeit(lambda: ops.gemm(A, W0, C, epilogue=epi, variant=variant))
                res[f"gemm_{name}_dense_v{variant}"] = dict(ms=ms, tflops=2.0 * M * N * K / ms / 1e9)
            except Exception as e:
                res[f"gemm_{name}_v{variant}"] = dict(error=repr(e))
            print(name, variant, res.get(f"gemm_{name}_v{variant}"), res.get(f"gemm_{name}_dense_v{variant}"), flush=True)
        del W0, W1, C
    # attention at the denoise shape
    nq, nkv, D, C_ctx = 28, 4, 128, 32
    qkv = rn(M, (nq + 2 * nkv) * D)
    cu_q = torch.tensor([b * (n_img + 2) for b in range(B + 1)], dtype=torch.int32, device=DEV)
    col = torch.tensor([b * 4160 for b in range(B)], dtype=torch.int32, device=DEV)
    vt = torch.zeros((nkv * D, 4160 * B), dtype=BF16, device=DEV)
    kc, vc = rn(C_ctx * B, nkv * D), rn(C_ctx * B, nkv * D)
    cu_c = torch.tensor([b * C_ctx for b in range(B + 1)], dtype=torch.int32, device=DEV)
    ccol = torch.tensor([b * 64 for b in range(B)], dtype=torch.int32, device=DEV)
    vtc = torch.zeros((nkv * D, 256), dtype=BF16, device=DEV)
    ops.v_transpose(vc, vtc, cu_c, ccol, B, C_ctx, nkv, D)
    out = torch.empty((M, nq * D), dtype=BF16, device=DEV)
    qw, kw = nq * D, nkv * D
    ms = timeit(lambda: ops.v_transpose(qkv[:, qw + kw:], vt, cu_q, col, B, n_img + 2, nkv, D))
    res["v_transpose"] = dict(ms=ms, gbps=2 * M * kw * 2 / ms / 1e6)
    ms = timeit(lambda: ops.attn_varlen(qkv[:, :qw], qkv[:, qw:qw + kw], vt, out, cu_q, col, B, n_img + 2, nq, nkv, D, False, D ** -0.5,
                                        k_ctx=kc, vt_ctx=vtc, cu_ctx=cu_c, vt_ctx_col=ccol))
    fl = 4.0 * B * (n_img + 2) * (n_img + 2 + C_ctx) * nq * D
    res["attn_denoise"] = dict(ms=ms, tflops=fl / ms / 1e9)
    print("attn", res["attn_denoise"], flush=True)
    # norms
    w = rn(H)
    y = torch.empty_like(x)
    ex = torch.zeros(M, dtype=torch.int32, device=DEV)
    ex[vr.long()] = 1
    ms = timeit(lambda: ops.rmsnorm(x, w, y, 1e-6, w1=w, expert=ex))
    res["rmsnorm"] = dict(ms=ms, gbps=2 * M * H * 2 / ms / 1e6)
    pos = torch.zeros(M, dtype=torch.long, device=DEV) + 34
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2).float() / D))).to(DEV)
    cos, sin = ops.rope_table(pos, inv)
    qn = (1 + 0.1 * torch.randn(D, device=DEV)).to(BF16)
    ms = timeit(lambda: ops.qknorm_rope(qkv, cos, sin, qn, qn, qn, qn, ex, nq, nkv, D, D, 1e-6, 1, 1))
    res["qknorm_rope"] = dict(ms=ms, gbps=2 * M * (qw + kw) * 2 / ms / 1e6)
    print(json.dumps(res, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
