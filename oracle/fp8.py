"""CPU restatement of the FP8 (OCP e4m3) option of the gen-expert GEMMs -- TEST INFRASTRUCTURE ONLY.

There is no reference implementation to pin against: the reference's quantised modes (app.py:114-131) are bitsandbytes NF4 / LLM.int8 on
CUDA, un-vendored; what the row in SURVEY.md 8f.4 asks for on MI355X is the chip's own low-precision MFMA path.  This file states the
scheme the kernels implement so the tests can check them bit for bit (quantiser) and to fp32-accumulation accuracy (GEMM):

    scale[r]  = max_k |x[r, k]| / 448                    (448 = largest finite e4m3fn value)
    q[r, k]   = e4m3fn(round-to-nearest-even(x[r, k] * (1 / scale[r])))
    C[m, n]   = epilogue( (sa[m] * sw[n]) * sum_k q_a[m, k] * q_w[n, k] )        fp32 accumulation, epilogues of the bf16 GEMM
"""
import torch


def quantize_rows_fp8(x):
    """bf16 / fp32 [rows, K] -> (uint8 view of float8_e4m3fn [rows, K], fp32 scale [rows]) exactly as bagel_quantize_rows_fp8:
    fp32 reciprocal, fp32 product, round-to-nearest-even conversion."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    inv = 1.0 / scale
    q = (xf * inv[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def dequant(q_u8, scale):
    return q_u8.view(torch.float8_e4m3fn).float() * scale[:, None]


def gemm_fp8(qa, sa, qw, sw, bias=None, residual=None, swiglu=False):
    """fp32 product of the de-quantised operands with the epilogue roundings of the bf16 GEMM (tests/test_ops_gpu.py::ref_gemm)."""
    acc = (qa.view(torch.float8_e4m3fn).float() @ qw.view(torch.float8_e4m3fn).float().t()) * (sa[:, None] * sw[None, :])
    if swiglu:
        N = qw.shape[0]
        a = acc.view(acc.shape[0], N // 32, 2, 16)
        g = a[:, :, 0].reshape(acc.shape[0], -1).to(torch.bfloat16)
        u = a[:, :, 1].reshape(acc.shape[0], -1).to(torch.bfloat16)
        return torch.nn.functional.silu(g) * u
    if bias is not None:
        acc = acc + bias.float()
    c = acc.to(torch.bfloat16)
    if residual is not None:
        c = residual + c
    return c
