"""Attention kernel at the BAGEL-7B denoise shape (B=4, Lq=4098, C=32, 28/4 heads, D=128): time + max error vs an
fp32 torch restatement on one (sample, head) slice.  Run on the GPU box: python tools/attn_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import ops  # noqa: E402

BF16, DEV = torch.bfloat16, "cuda"


def timeit(fn, iters=20, warm=3):
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
    B, n_img, nq, nkv, D, C_ctx = 4, 4096, 28, 4, 128, 32
    Lq = n_img + 2
    M = B * Lq
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV).to(BF16)  # noqa: E731
    qkv = rn(M, (nq + 2 * nkv) * D)
    qw, kw = nq * D, nkv * D
    cu_q = torch.tensor([b * Lq for b in range(B + 1)], dtype=torch.int32, device=DEV)
    col = torch.tensor([b * 4160 for b in range(B)], dtype=torch.int32, device=DEV)
    vt = torch.zeros((nkv * D, 4160 * B), dtype=BF16, device=DEV)
    kc, vc = rn(C_ctx * B, kw), rn(C_ctx * B, kw)
    cu_c = torch.tensor([b * C_ctx for b in range(B + 1)], dtype=torch.int32, device=DEV)
    ccol = torch.tensor([b * 64 for b in range(B)], dtype=torch.int32, device=DEV)
    vtc = torch.zeros((nkv * D, 256), dtype=BF16, device=DEV)
    ops.v_transpose(vc, vtc, cu_c, ccol, B, C_ctx, nkv, D)
    ops.v_transpose(qkv[:, qw + kw:], vt, cu_q, col, B, Lq, nkv, D)
    out = torch.empty((M, nq * D), dtype=BF16, device=DEV)
    run = lambda causal=False: ops.attn_varlen(qkv[:, :qw], qkv[:, qw:qw + kw], vt, out, cu_q, col, B, Lq, nq, nkv, D, causal,  # noqa: E731
                                               D ** -0.5, k_ctx=kc, vt_ctx=vtc, cu_ctx=cu_c, vt_ctx_col=ccol)
    run()
    torch.cuda.synchronize()
    # reference on a few (sample, head) slices
    worst = 0.0
    for b, h in ((0, 0), (1, 9), (3, 27)):
        gk = h // (nq // nkv)
        q = qkv[b * Lq:(b + 1) * Lq, h * D:(h + 1) * D].float()
        k = torch.cat([kc[b * C_ctx:(b + 1) * C_ctx, gk * D:(gk + 1) * D], qkv[b * Lq:(b + 1) * Lq, qw + gk * D:qw + (gk + 1) * D]]).float()
        v = torch.cat([vc[b * C_ctx:(b + 1) * C_ctx, gk * D:(gk + 1) * D], qkv[b * Lq:(b + 1) * Lq, qw + kw + gk * D:qw + kw + (gk + 1) * D]]).float()
        ref = torch.softmax(q @ k.t() * D ** -0.5, -1) @ v
        got = out[b * Lq:(b + 1) * Lq, h * D:(h + 1) * D].float()
        worst = max(worst, ((got - ref).abs().max() / ref.abs().max()).item())
    dump = os.environ.get("BAGEL_ATTN_PROBE_DUMP")
    if dump:
        full = out.cpu().clone()
        run(True)
        torch.cuda.synchronize()
        torch.save({"full": full, "causal": out.cpu().clone()}, dump)
    ms = timeit(run)
    fl = 4.0 * B * Lq * (Lq + C_ctx) * nq * D
    tag = ""
    print(f"{tag}attn_denoise: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  max_rel_err {worst:.3e}", flush=True)
    ms_c = timeit(lambda: run(True))
    print(f"{tag}attn_causal : {ms_c:.3f} ms  {fl / 2 / ms_c / 1e9:.1f} TFLOP/s (half the work)", flush=True)


if __name__ == "__main__":
    main()
