"""flash_attn stand-in for an otherwise STOCK reference tree: binds ``bagel_attn_varlen_bf16`` / ``bagel_v_transpose_bf16``
(include/bagel_hip.h) over ctypes -- no torch extension, no import of the rest of bagel_amd.

Put the directory that contains this package on ``sys.path`` before the reference
(``sys.path.insert(0, ".../bagel_amd/integration")``); the reference's
``from flash_attn import flash_attn_varlen_func`` (qwen2_navit.py:24, siglip_navit.py:14) then resolves here and its three call sites
(qwen2_navit.py:361-370, 579-588; siglip_navit.py:232-241) run on the hand-written gfx950 kernel.

Semantics = flash-attn 2.5.8's ``flash_attn_varlen_func`` for what the reference uses: packed (T, H, D) bf16 q / k / v, int32
``cu_seqlens_*``, GQA by ``Hq // Hk``, fp32 softmax, ``causal`` bottom-right aligned, ``softmax_scale`` default D^-0.5, output in
q's dtype and shape.  The kernel takes head dims 64 and 128: any other head_dim (SigLIP so400m: 72) is zero-padded to the next
supported size -- zero q/k lanes leave every score unchanged, zero v lanes produce output columns that are sliced off -- with the
softmax scale still taken from the ORIGINAL head_dim.  Keys and queries of a sample may differ in number (cached LLM forwards pass
the merged [context | new] keys): the queries are then the LAST ``Lq`` positions of the sample's key range, which is exactly the
bottom-right alignment, so the call maps onto the kernel's two-segment form with context = the first ``Lk - Lq`` keys.
Training: when grad mode is on and q / k / v require grad, the SELF-attention form (same query and key ranges: SigLIP's call,
siglip_navit.py:232-241, the one flash-attn call the reference's training step makes) is an autograd node whose backward is the
hand-written reverse of csrc/attention_bwd.hip (row statistics from ``bagel_attn_varlen_ranges_lse_bf16``, one "full" or "causal"
split per sample); the cached two-segment form is inference-only and refuses inputs that require grad.
Fails loudly (RuntimeError with ``bagel_hip_last_error()``) -- there is no fallback."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("BAGEL_HIP_LIB", os.path.normpath(os.path.join(_HERE, "..", "..", "libbagel_hip.so")))
_L = ctypes.CDLL(_LIB_PATH)
_L.bagel_hip_last_error.restype = ctypes.c_char_p
_P, _I64, _I32, _F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
_L.bagel_v_transpose_bf16.argtypes = [_P, _I64, _P, _I64, _P, _P, _I32, _I32, _I32, _I32, _P]
_L.bagel_v_transpose_bf16.restype = ctypes.c_int
_L.bagel_attn_varlen_ranges_bf16.argtypes = [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P,
                                             _I32, _I32, _I32, _I32, _I32, _I32, _F, _P]
_L.bagel_attn_varlen_ranges_bf16.restype = ctypes.c_int
_L.bagel_attn_varlen_ranges_lse_bf16.argtypes = [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P,
                                                 _I32, _I32, _I32, _I32, _I32, _I32, _F, _P, _I64, _P]
_L.bagel_attn_varlen_ranges_lse_bf16.restype = ctypes.c_int
_L.bagel_transpose_bf16.argtypes = [_P, _I64, _P, _I32, _I32, _P, _I64, _I32, _P]
_L.bagel_transpose_bf16.restype = ctypes.c_int
_L.bagel_attn_bwd_blockmask_bf16.argtypes = [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _I64, _P, _I64,
                                             _P, _I32, _P, _I32, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _P]
_L.bagel_attn_bwd_blockmask_bf16.restype = ctypes.c_int


def _check(rc):
    if rc:
        raise RuntimeError("libbagel_hip: " + _L.bagel_hip_last_error().decode())


def _i32(t):
    return t.to(torch.int32).contiguous()


def _transpose(x):
    """[rows, C] bf16 -> [C, ceil64(rows)] (zero padded): the K-contiguous operand images of the attention reverse."""
    rows, C = x.shape
    npad = (max(rows, 1) + 63) // 64 * 64
    out = torch.empty((C, npad), dtype=x.dtype, device=x.device)
    _check(_L.bagel_transpose_bf16(x.data_ptr(), x.stride(0), None, rows, C, out.data_ptr(), npad, npad, torch.cuda.current_stream().cuda_stream))
    return out


class _SelfAttnVarlen(torch.autograd.Function):
    """flash_attn_varlen_func for cu_seqlens_q == cu_seqlens_k with a backward: forward = the tile kernel + row statistics, backward =
    bagel_attn_bwd_blockmask_bf16 over 128-row work items, one split per sample ("causal" or "full")."""

    @staticmethod
    def forward(ctx, q, k, v, cu, max_len, scale, causal):
        Tq, Hq, D0 = q.shape
        Hk = k.shape[1]
        D = 64 if D0 <= 64 else 128
        if D != D0:
            pad = lambda t: torch.nn.functional.pad(t, (0, D - D0))  # noqa: E731
            q, k, v = pad(q), pad(k), pad(v)
        q2, k2, v2 = q.reshape(Tq, Hq * D).contiguous(), k.reshape(Tq, Hk * D).contiguous(), v.reshape(Tq, Hk * D).contiguous()
        dev, s = q.device, torch.cuda.current_stream().cuda_stream
        B = cu.numel() - 1
        lens = cu[1:] - cu[:-1]
        blk = (lens + 63) // 64 * 64
        col = torch.zeros(B, dtype=torch.int32, device=dev)
        col[1:] = torch.cumsum(blk[:-1], 0)
        cols = int(blk.sum()) + 64
        vt = torch.zeros((Hk * D, cols), dtype=q.dtype, device=dev)
        _check(_L.bagel_v_transpose_bf16(v2.data_ptr(), Hk * D, vt.data_ptr(), cols, cu.data_ptr(), col.data_ptr(), B, int(max_len), Hk, D, s))
        out = torch.empty_like(q2)
        lse = torch.empty((Hq, Tq), dtype=torch.float32, device=dev)
        q_start, q_end = cu[:-1].contiguous(), cu[1:].contiguous()
        _check(_L.bagel_attn_varlen_ranges_lse_bf16(q2.data_ptr(), Hq * D, k2.data_ptr(), Hk * D, vt.data_ptr(), cols, None, Hk * D, None, 0,
                                                    out.data_ptr(), Hq * D, q_start.data_ptr(), q_end.data_ptr(), None, None, col.data_ptr(), None,
                                                    B, int(max_len), Hq, Hk, D, int(bool(causal)), scale, lse.data_ptr(), Tq, s))
        ctx.save_for_backward(q2, k2, v2, out, lse)
        ctx.meta = (Tq, Hq, Hk, D0, D, scale, bool(causal), [int(x) for x in cu.tolist()])
        o3 = out.view(Tq, Hq, D)
        return o3[..., :D0].contiguous() if D != D0 else o3.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        q2, k2, v2, out, lse = ctx.saved_tensors
        Tq, Hq, Hk, D0, D, scale, causal, cu = ctx.meta
        dev, s = q2.device, torch.cuda.current_stream().cuda_stream
        do = d_out.to(torch.bfloat16)
        if D != D0:
            do = torch.nn.functional.pad(do, (0, D - D0))
        do2 = do.reshape(Tq, Hq * D).contiguous()
        q_items, k_items = [], []
        for a0, a1 in zip(cu[:-1], cu[1:]):
            for r0 in range(a0, a1, 128):
                nr = min(128, a1 - r0)
                q_items.append((r0, nr, a0, a0, a1, int(causal), a0 // 64, -(-((r0 + nr) if causal else a1) // 64)))
                k_items.append((r0, nr, r0 if causal else a0, a1, a1, int(causal), 0, 0))
        qi = torch.tensor(q_items, dtype=torch.int32, device=dev).reshape(-1, 8)
        ki = torch.tensor(k_items, dtype=torch.int32, device=dev).reshape(-1, 8)
        noise = torch.zeros(((Tq + 63) // 64,), dtype=torch.int64, device=dev)
        qt, dot, kt = _transpose(q2), _transpose(do2), _transpose(k2)
        ws = torch.empty((2, Hq, Tq), dtype=torch.float32, device=dev)
        ws[0].copy_(lse)
        dq, dk, dv = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
        _check(_L.bagel_attn_bwd_blockmask_bf16(q2.data_ptr(), Hq * D, k2.data_ptr(), Hk * D, v2.data_ptr(), Hk * D, out.data_ptr(), Hq * D,
                                                do2.data_ptr(), Hq * D, qt.data_ptr(), dot.data_ptr(), kt.data_ptr(), qt.stride(0),
                                                dq.data_ptr(), Hq * D, dk.data_ptr(), Hk * D, dv.data_ptr(), Hk * D, qi.data_ptr(), qi.shape[0],
                                                ki.data_ptr(), ki.shape[0], noise.data_ptr(), ws.data_ptr(), 1, Tq, Hq, Hk, D, scale, s))
        cut = lambda t, H: t.view(Tq, H, D)[..., :D0].contiguous()  # noqa: E731
        return cut(dq, Hq), cut(dk, Hk), cut(dv, Hk), None, None, None, None


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0, softmax_scale=None,
                           causal=False, **unused):
    if dropout_p:
        raise NotImplementedError("flash_attn stand-in: dropout is not part of the inference path")
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16 or not q.is_cuda:
        raise RuntimeError("flash_attn stand-in: q / k / v must be bf16 tensors on the GPU")
    Tq, Hq, D0 = q.shape
    Tk, Hk, _ = k.shape
    B = cu_seqlens_q.numel() - 1
    scale = float(softmax_scale) if softmax_scale is not None else D0 ** -0.5
    D = 64 if D0 <= 64 else 128
    if D0 > 128:
        raise NotImplementedError(f"flash_attn stand-in: head_dim {D0} > 128")
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        if Tq != Tk or not torch.equal(cu_seqlens_q.to(torch.int64).cpu(), cu_seqlens_k.to(torch.int64).cpu()):
            raise NotImplementedError("flash_attn stand-in: the backward is built for self-attention (cu_seqlens_q == cu_seqlens_k); the cached "
                                      "two-segment form is inference-only")
        return _SelfAttnVarlen.apply(q, k, v, _i32(cu_seqlens_q), int(max_seqlen_q), scale, bool(causal))
    if D != D0:
        pad = lambda t: torch.nn.functional.pad(t, (0, D - D0))  # noqa: E731
        q, k, v = pad(q), pad(k), pad(v)
    q2, k2, v2 = q.reshape(Tq, Hq * D).contiguous(), k.reshape(Tk, Hk * D).contiguous(), v.reshape(Tk, Hk * D).contiguous()
    dev = q.device
    s = torch.cuda.current_stream().cuda_stream
    cq, ck = _i32(cu_seqlens_q), _i32(cu_seqlens_k)
    lq, lk = cq[1:] - cq[:-1], ck[1:] - ck[:-1]
    if bool((lk < lq).any()):
        raise RuntimeError("flash_attn stand-in: a sample has fewer keys than queries")
    nctx = lk - lq
    has_ctx = bool((nctx > 0).any())

    def transposed(rows, cu, lens, max_len):
        """V^T image [Hk * D, cols] of packed rows: sample b at column col[b] (64-aligned), zero padded."""
        blk = (lens + 63) // 64 * 64
        col = torch.zeros(B, dtype=torch.int32, device=dev)
        col[1:] = torch.cumsum(blk[:-1], 0)
        cols = int(blk.sum()) + 64
        vt = torch.zeros((Hk * D, cols), dtype=q.dtype, device=dev)
        _check(_L.bagel_v_transpose_bf16(rows.data_ptr(), Hk * D, vt.data_ptr(), cols, cu.data_ptr(), col.data_ptr(), B, int(max_len), Hk, D, s))
        return vt, cols, col
    # two-segment form: context = the first Lk - Lq keys of a sample, "new" = its last Lq keys (= the query positions).  The kernel
    # addresses the new segment's K rows by the QUERY row range, so with a context the query-aligned K / V rows are gathered once
    # (a copy of Tq rows, not of the context); without one (ViT, CFG without context) k / v are used as they are.
    if has_ctx:
        idx = torch.cat([torch.arange(int(ck[b] + nctx[b]), int(ck[b + 1]), device=dev) for b in range(B)])
        kn, vn = k2.index_select(0, idx), v2.index_select(0, idx)
        vt, cols, col = transposed(v2, ck, lk, max_seqlen_k)
    else:
        kn, vn = k2, v2
        vt, cols, col = None, 0, None
    vt_new, ld_new, vt_new_col = transposed(vn, cq, lq, max_seqlen_q)
    out = torch.empty_like(q2)
    ctx_start, ctx_end = ck[:-1].contiguous(), (ck[:-1] + nctx).to(torch.int32).contiguous()
    q_start, q_end = cq[:-1].contiguous(), cq[1:].contiguous()
    _check(_L.bagel_attn_varlen_ranges_bf16(
        q2.data_ptr(), Hq * D, kn.data_ptr(), Hk * D, vt_new.data_ptr(), ld_new,
        k2.data_ptr() if has_ctx else None, Hk * D, vt.data_ptr() if has_ctx else None, cols if has_ctx else 0,
        out.data_ptr(), Hq * D, q_start.data_ptr(), q_end.data_ptr(),
        ctx_start.data_ptr() if has_ctx else None, ctx_end.data_ptr() if has_ctx else None,
        vt_new_col.data_ptr(), col.data_ptr() if has_ctx else None,
        B, int(max_seqlen_q), Hq, Hk, D, int(bool(causal)), scale, s))
    out = out.view(Tq, Hq, D)
    return out[..., :D0].contiguous() if D != D0 else out
