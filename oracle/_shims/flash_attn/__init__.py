"""Pure-PyTorch stand-in for ``flash_attn`` so the unmodified reference imports on a CPU box.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  The reference's only by-name native import is
``flash_attn.flash_attn_varlen_func`` (qwen2_navit.py:24, siglip_navit.py:18; pinned
flash_attn==2.5.8, README.md:116 -- source NOT under /root/reference).  This module restates the
*published* semantics of that function: per-sample softmax(q k^T * D^-1/2 [+ bottom-right aligned
causal mask, flash-attn >= 2.1]) v with GQA head sharing, fp32 softmax, output in q's dtype.
It is the definition of "expected attention result" for every parity test in this repo.
"""
import torch

__version__ = "2.5.8+oracle"


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                           dropout_p=0.0, softmax_scale=None, causal=False, **kw):
    # The math below is fp32 by definition; an ambient torch.autocast('cpu', bf16) region (the reference
    # is run under one) must not downcast these matmuls.
    with torch.autocast("cpu", enabled=False):
        return _attn_fp32(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale, causal)


def _attn_fp32(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale, causal):
    Tq, Hq, D = q.shape
    group = Hq // k.shape[1]
    scale = softmax_scale if softmax_scale is not None else D ** -0.5
    out = torch.empty_like(q)
    cq = cu_seqlens_q.tolist()
    ck = cu_seqlens_k.tolist()
    for b in range(len(cq) - 1):
        qs, qe, ks, ke = cq[b], cq[b + 1], ck[b], ck[b + 1]
        qi = q[qs:qe].transpose(0, 1).float()
        ki = k[ks:ke].transpose(0, 1).float().repeat_interleave(group, dim=0)
        vi = v[ks:ke].transpose(0, 1).float().repeat_interleave(group, dim=0)
        s = torch.matmul(qi, ki.transpose(1, 2)) * scale
        if causal:
            Lq, Lk = qe - qs, ke - ks
            keep = torch.ones(Lq, Lk, dtype=torch.bool).tril(diagonal=Lk - Lq)
            s = s.masked_fill(~keep, float("-inf"))
        out[qs:qe] = torch.matmul(torch.softmax(s, dim=-1), vi).transpose(0, 1).to(q.dtype)
    return out
