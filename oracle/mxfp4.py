"""CPU restatement of the MXFP4 weight-only option of the text-decode projections -- TEST INFRASTRUCTURE ONLY.

The reference's 4-bit load mode is bitsandbytes NF4 (app.py:114-125), an un-vendored CUDA dependency; its MI355X-native counterpart is
the chip's own block-scaled 4-bit MFMA operand (OCP Microscaling Formats v1.0, "MXFP4": 32 elements of FP4 E2M1 sharing one E8M0
power-of-two scale), consumed by ``v_mfma_scale_f32_16x16x128_f8f6f4`` with FP8 (e4m3) activations.  Parity with bitsandbytes itself
is therefore **unpinned**; this file states the scheme the kernels implement so that the tests can check the quantiser bit for bit and
the product to fp32-accumulation accuracy:

    weights      per row and block of 32 along K:   e  = biased fp32 exponent of max |w|  (0 for an all-zero block)
                                                     sb = max(e - 2, 0)                     E8M0 byte: X = 2^(sb - 127)  (emax(E2M1) = 2)
                                                     c  = E2M1 code of round-to-nearest-even(|w| / X), saturating at 6, sign in bit 3
    activations  per row:  s = max |x| / 448,  q = e4m3fn(rne(x / s))                      (oracle/fp8.py)
    C[m, n]      = epilogue( s[m] * sum_k q[m, k] * (E2M1(c[n, k]) * 2^(sb[n, k/32] - 127)) )   fp32 accumulation, epilogues of the bf16 GEMM

Storage: codes packed two per byte (element 2i in the low nibble of byte i); the device keeps the scale bytes of a row permuted so that
a lane's four consecutive k-steps sit in one dword (``permute_scales``)."""
import torch

from . import fp8

E2M1 = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def quantize_mxfp4(w):
    """bf16 / fp32 [N, K] (K % 32 == 0) -> (codes uint8 [N, K/2], scale bytes uint8 [N, K/32]) in natural (un-permuted) order."""
    wf = w.float()
    N, K = wf.shape
    assert K % 32 == 0
    blk = wf.view(N, K // 32, 32)
    amax = blk.abs().amax(dim=2)
    e = (amax.view(torch.int32) >> 23) & 0xFF                       # biased exponent (bf16 / fp32 inputs: exact)
    sb = (e - 2).clamp(min=0)
    X = torch.pow(torch.tensor(2.0, dtype=torch.float64), (sb - 127).double()).float()      # exact powers of two (2^-127 is a subnormal fp32)
    a = (blk.abs().double() / X[:, :, None].double()).float()       # exact: division by a power of two
    # round-to-nearest-even onto {0, .5, 1, 1.5, 2, 3, 4, 6}: ties go to the code with an even mantissa bit
    code = ((a > 0.25).int() + (a >= 0.75).int() + (a > 1.25).int() + (a >= 1.75).int() + (a > 2.5).int() + (a >= 3.5).int() + (a > 5.0).int())
    code = code | ((blk < 0).int() << 3)
    code = code.view(N, K).to(torch.uint8)
    packed = code[:, 0::2] | (code[:, 1::2] << 4)
    return packed.contiguous(), sb.to(torch.uint8).contiguous()


def dequant_mxfp4(packed, sb):
    N, Kh = packed.shape
    lo, hi = (packed & 0xF).long(), (packed >> 4).long()
    code = torch.stack([lo, hi], dim=2).view(N, 2 * Kh)
    val = E2M1[code & 7] * torch.where((code & 8) != 0, -1.0, 1.0)
    X = torch.pow(torch.tensor(2.0, dtype=torch.float64), (sb.long() - 127).double()).float()
    return (val.view(N, -1, 32) * X[:, :, None]).view(N, 2 * Kh)


def permute_scales(sb):
    """natural [N, K/32] -> device order: groups of 4 k-steps (a k-step = 128 elements = 4 blocks); inside a group byte 4*q + j holds
    block q of k-step j, so lane group q of the MFMA reads the scales of its next four k-steps as ONE dword.  Groups are padded with
    127 (2^0) where K/128 is not a multiple of 4."""
    N, nb = sb.shape
    assert nb % 4 == 0
    nk = nb // 4
    ng = (nk + 3) // 4
    out = torch.full((N, ng, 4, 4), 127, dtype=torch.uint8)
    v = sb.view(N, nk, 4)                                            # [row, k-step, q]
    for j in range(4):
        sel = v[:, j::4]                                             # k-steps j, j+4, ... -> groups 0, 1, ...
        out[:, :sel.shape[1], :, j] = sel
    return out.view(N, ng * 16).contiguous()


def gemv_w4(x, packed, sb, bias=None, residual=None, swiglu=False, norm_w=None, eps=1e-6):
    """The product the device computes for activation rows x (bf16): optional Qwen2RMSNorm (und cast points), FP8 row quantisation,
    fp32 product with the de-quantised weights, epilogue roundings of the bf16 GEMM."""
    xb = x
    if norm_w is not None:
        xf = x.float()
        inv = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
        xb = norm_w * (xf * inv).to(torch.bfloat16)
    q, s = fp8.quantize_rows_fp8(xb)
    acc = (q.view(torch.float8_e4m3fn).float() @ dequant_mxfp4(packed, sb).t()) * s[:, None]
    if swiglu:
        N = packed.shape[0]
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
