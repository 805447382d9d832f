"""TEST INFRASTRUCTURE ONLY -- a torch-CPU stand-in for the operators of ``bagel_amd.ops`` that the prefill and the
image-generation path launch, so that the HOST logic above the C ABI (ForwardPlan, NaiveCache, MoT routing lists, the
denoise loop, stream-batched CFG, the marker-row side path) can be exercised by ``pytest -m "not gpu"`` in a container
without a GPU.

It is NOT a fallback of the product: nothing under ``bagel_amd/`` imports it, the product still raises
(``bagel_amd._lib`` / ``ops._ptr``) when it is handed a host tensor, and no parity or performance claim rests on it --
the kernels themselves are checked on the MI355X by the ``-m gpu`` tests.  A test installs it explicitly with
``mock_ops.install(monkeypatch)``.

Every function follows the data-layout contract of the entry point it stands in for (include/bagel_hip.h): the fused
[q | k | v] projection rows with padded head slots, V^T images with per-sample column offsets, row gather / scatter
lists, the SwiGLU16 gate/up row interleave.  Products are accumulated in fp64 and rounded once to fp32, so a row's
result does not depend on which rows it is batched with: two host-side schedules of the same arithmetic must agree
bit for bit, which is what the stream-batching tests assert."""
import math

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def _bf(x):
    return x.to(BF16)


def _rows(idx, n, M):
    return idx.long()[:n] if idx is not None else torch.arange(M)


def gemm(A, W0, C, *, bias0=None, a_rows0=None, c_rows0=None, M0=None, W1=None, bias1=None, a_rows1=None, c_rows1=None,
         M1=0, residual=None, epilogue=0, variant=None, splitk=True):
    N, K = W0.shape
    if M0 is None:
        M0 = a_rows0.numel() if a_rows0 is not None else A.shape[0]
    for W, b, ar, cr, M in ((W0, bias0, a_rows0, c_rows0, M0), (W1, bias1, a_rows1, c_rows1, M1)):
        if W is None or M == 0:
            continue
        ai, ci = _rows(ar, M, M), _rows(cr, M, M)
        acc = (A[ai, :K].double() @ W.double().t()).float()
        if b is not None:
            acc = acc + b.float()
        y = _bf(acc)
        if epilogue == 1:
            y = _bf(F.gelu(y.float(), approximate="tanh"))
        elif epilogue == 2:
            y = _bf(F.silu(y.float()))
        elif epilogue == 3:      # rows [16 gate | 16 up] ... -> N/2 columns
            y = y.view(M, N // 32, 2, 16)
            g, u = y[:, :, 0], y[:, :, 1]
            y = _bf(_bf(F.silu(g.float())).float() * u.float()).reshape(M, N // 2)
        if residual is not None:
            y = _bf(y.float() + residual[ci, :y.shape[1]].float())
        C[ci, :y.shape[1]] = y
    return C


def gemv(A, W, C, *, bias=None, residual=None, epilogue=0, M=None, norm_w=None, eps=0.0):
    if norm_w is not None:
        A = rmsnorm(A, norm_w, torch.empty_like(A), eps)
    return gemm(A, W, C, bias0=bias, residual=residual, epilogue=epilogue, M0=M if M is not None else A.shape[0])


def gemv_mb(A, W, C, *, bias=None, residual=None, epilogue=0, M=None, norm_w=None, eps=0.0, workspace=None):
    return gemv(A, W, C, bias=bias, residual=residual, epilogue=epilogue, M=M, norm_w=norm_w, eps=eps)


def gemm_skinny(A, W, C, *, bias=None, residual=None, epilogue=0, M=None):
    return gemm(A, W, C, bias0=bias, residual=residual, epilogue=epilogue, M0=M if M is not None else A.shape[0])


def rmsnorm(x, w0, out, eps, w1=None, expert=None):
    h = x.float()
    h = _bf(h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps))
    w = w0.float().expand(x.shape[0], -1)
    if expert is not None and w1 is not None:
        w = torch.where(expert.bool()[:, None], w1.float()[None], w0.float()[None])
    out.copy_(_bf(w * h.float()))
    return out


def layernorm(x, w, b, out, eps):
    out.copy_(_bf(F.layer_norm(x.float(), (x.shape[1],), w.float(), b.float(), eps)))
    return out


def rope_table(position_ids, inv_freq):
    ang = position_ids.float()[:, None] * inv_freq.float()[None, :]
    return _bf(torch.cos(ang)), _bf(torch.sin(ang))


def qknorm_rope(qkv, cos, sin, q_w0, k_w0, q_w1, k_w1, expert, nq, nkv, head_dim, head_dim_padded, eps, gen_mode, use_norm):
    M, hd, dp = qkv.shape[0], head_dim, head_dim_padded
    heads = qkv[:, :(nq + nkv) * dp].view(M, nq + nkv, dp)
    x = heads[:, :, :hd].float()
    ex = expert.bool() if expert is not None else torch.zeros(M, dtype=torch.bool)
    if use_norm:
        wq = torch.where(ex[:, None], q_w1.float()[None], q_w0.float()[None]) if q_w1 is not None else q_w0.float().expand(M, -1)
        wk = torch.where(ex[:, None], k_w1.float()[None], k_w0.float()[None]) if k_w1 is not None else k_w0.float().expand(M, -1)
        w = torch.cat([wq[:, None, :].expand(M, nq, hd), wk[:, None, :].expand(M, nkv, hd)], 1)
        inv = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
        x = w * (x * inv) if gen_mode else _bf(w * _bf(x * inv).float()).float()
    c = torch.cat([cos, cos], -1).float()[:, None, :]
    s = torch.cat([sin, sin], -1).float()[:, None, :]
    rot = torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1)
    out = x * c + rot * s if gen_mode else _bf(x * c).float() + _bf(rot * s).float()
    heads[:, :, :hd] = _bf(out)
    return qkv


def v_transpose(v, vt, cu_rows, col_start, batch, max_len, nkv, head_dim):
    cu, col = cu_rows.tolist(), col_start.tolist()
    for b in range(batch):
        L = cu[b + 1] - cu[b]
        if L <= 0:
            continue
        pad = (L + 63) // 64 * 64
        vt[:nkv * head_dim, col[b]:col[b] + pad] = 0
        vt[:nkv * head_dim, col[b]:col[b] + L] = v[cu[b]:cu[b + 1], :nkv * head_dim].t()
    return vt


def attn_varlen(q, k_new, vt_new, out, cu_q, vt_new_col, batch, max_lq, nq, nkv, head_dim, causal, softmax_scale,
                k_ctx=None, vt_ctx=None, cu_ctx=None, vt_ctx_col=None):
    D, grp = head_dim, nq // nkv
    cq, ncol = cu_q.tolist(), vt_new_col.tolist()
    cc = cu_ctx.tolist() if cu_ctx is not None else None
    ccol = vt_ctx_col.tolist() if vt_ctx_col is not None else None
    for b in range(batch):
        q0, Lq = cq[b], cq[b + 1] - cq[b]
        C = cc[b + 1] - cc[b] if cc is not None else 0
        if Lq <= 0:
            continue
        for h in range(nq):
            g = h // grp
            ks = [k_new[q0:q0 + Lq, g * D:(g + 1) * D]]
            vs = [vt_new[g * D:(g + 1) * D, ncol[b]:ncol[b] + Lq].t()]
            if C > 0:
                ks.insert(0, k_ctx[cc[b]:cc[b] + C, g * D:(g + 1) * D])
                vs.insert(0, vt_ctx[g * D:(g + 1) * D, ccol[b]:ccol[b] + C].t())
            k, v = torch.cat(ks).double(), torch.cat(vs).double()
            s = (q[q0:q0 + Lq, h * D:(h + 1) * D].double() @ k.t()).float() * softmax_scale
            if causal:      # bottom-right aligned
                Lk = C + Lq
                mask = torch.ones(Lq, Lk).tril(diagonal=Lk - Lq).bool()
                s = s.masked_fill(~mask, float("-inf"))
            p = torch.softmax(s, -1)
            out[q0:q0 + Lq, h * D:(h + 1) * D] = _bf((p.double() @ v).float())
    return out


def copy_rows(src, dst, n, cols, src_rows=None, dst_rows=None):
    si = src_rows.long()[:n] if src_rows is not None else torch.arange(n)
    di = dst_rows.long()[:n] if dst_rows is not None else torch.arange(n)
    dst[di, :cols] = src[si, :cols]
    return dst


def f32_to_bf16(src, dst=None, cols_padded=None):
    rows, cols = src.shape
    cp = cols if cols_padded is None else cols_padded
    if dst is None:
        dst = torch.empty((rows, cp), dtype=BF16)
    dst[:, :cols] = _bf(src)
    dst[:, cols:cp] = 0
    return dst


def timestep_sinusoid(t, freqs, out):
    half = freqs.numel()
    a = torch.tensor(float(t), dtype=torch.float32) * freqs.float()
    out.view(-1)[:half] = _bf(torch.cos(a))
    out.view(-1)[half:2 * half] = _bf(torch.sin(a))


def flow_add(seq, rows, temb, pos_table, pos_ids):
    r = rows.long()
    seq[r] = _bf(_bf(seq[r].float() + temb.float().view(1, -1)).float() + pos_table[pos_ids.long()].float())
    return seq


def add_table_rows(x, table, ids):
    x.copy_(_bf(x.float() + table[ids.long()].float()))
    return x


def _r(x):
    return x.to(BF16).float()


def _mix(base, x, s):
    return _r(base + _r(s * _r(x - base)))


def _scale(ss0, ss1, mn):
    n0, n1 = _r(torch.sqrt(ss0)), _r(torch.sqrt(ss1))
    return torch.clamp(_r(n0 / _r(n1 + 1e-8)), min=float(_r(torch.tensor(mn))), max=1.0)


def cfg_stage1(v, v_ct, v_ci, tmp, partials, text_scale, img_scale, renorm_min, mode):
    a = v.float()
    vt_ = _mix(v_ct.float(), a, text_scale)
    t = vt_ if (mode == 2 or v_ci is None) else _mix(v_ci.float(), vt_, img_scale)
    if mode == 0:
        tmp.copy_(_bf(t))
        partials[0], partials[1] = a.pow(2).sum(), t.pow(2).sum()
        return 1
    sc = _scale(a.pow(2).sum(-1, keepdim=True), t.pow(2).sum(-1, keepdim=True), renorm_min)
    o = _r(t * sc)
    if mode == 2 and v_ci is not None:
        o = _mix(v_ci.float(), o, img_scale)
    tmp.copy_(_bf(o))
    return 1


def cfg_stage2_euler(x_t, v_or_tmp, partials, nparts, renorm_min, dt, use_global_scale):
    vt = v_or_tmp.float()
    if use_global_scale:
        vt = _r(vt * _scale(partials[0], partials[1], renorm_min))
    x_t.sub_(_r(vt * dt).view(x_t.shape))
    return x_t


def argmax(logits):
    return torch.argmax(logits.float(), -1)


def argmax_into(logits, out):
    out.copy_(torch.argmax(logits.float(), -1))


def sample_gumbel_into(logits, out, temperature, seed, step_ctr=None):
    """bagel_sample_gumbel_bf16 through its restatement (oracle/sampling.py)."""
    from oracle import sampling
    step = 0 if step_ctr is None else int(step_ctr[0])
    out.copy_(torch.from_numpy(sampling.sample_gumbel(logits.float().numpy(), float(temperature), int(seed), step)))
    return out


def rope_table_into(position_ids, inv_freq, cos, sin):
    c, s_ = rope_table(position_ids, inv_freq)
    cos.copy_(c.view(cos.shape))
    sin.copy_(s_.view(sin.shape))
    return cos, sin


def _page_row(block_table, b, j, page=64):
    return int(block_table[b, j // page]) * page + j % page


def decode_qkv_post(qkv, cos, sin, q_w, k_w, kpool, vpool, block_table, kv_len, batch, nq, nkv, head_dim, head_dim_padded, eps,
                    use_norm):
    qknorm_rope(qkv[:batch], cos, sin, q_w, k_w, None, None, None, nq, nkv, head_dim, head_dim_padded, eps, False, use_norm)
    qw, kw = nq * head_dim_padded, nkv * head_dim_padded
    for b in range(batch):
        r = _page_row(block_table, b, int(kv_len[b]))
        kpool[r, :kw] = qkv[b, qw:qw + kw]
        vpool[r, :kw] = qkv[b, qw + kw:qw + 2 * kw]
    return qkv


def kv_append_paged(k_new, v_new, kpool, vpool, block_table, kv_len, batch, width):
    for b in range(batch):
        r = _page_row(block_table, b, int(kv_len[b]))
        kpool[r, :width] = k_new[b, :width]
        vpool[r, :width] = v_new[b, :width]


def attn_decode_paged(q, kpool, vpool, block_table, kv_len, len_add, max_len, part_o, part_ml, out, batch, nq, nkv, head_dim,
                      softmax_scale):
    D, grp = head_dim, nq // nkv
    for b in range(batch):
        n = int(kv_len[b]) + len_add
        rows = torch.tensor([_page_row(block_table, b, j) for j in range(n)], dtype=torch.long)
        for h in range(nq):
            g = h // grp
            k, v = kpool[rows, g * D:(g + 1) * D].double(), vpool[rows, g * D:(g + 1) * D].double()
            s = (q[b:b + 1, h * D:(h + 1) * D].double() @ k.t()).float() * softmax_scale
            out[b, h * D:(h + 1) * D] = _bf((torch.softmax(s, -1).double() @ v).float())[0]
    return out


def attn_decode_fused(qkv, cos, sin, q_w, k_w, kpool, vpool, block_table, kv_len, max_len, part_o, part_ml, out, batch, nq, nkv,
                      head_dim, head_dim_padded, eps, use_norm, softmax_scale):
    """decode_qkv_post + attn_decode_paged in one launch; the projection buffer stays untouched (the kernel works on copies in LDS)."""
    tmp = qkv.clone()
    decode_qkv_post(tmp, cos, sin, q_w, k_w, kpool, vpool, block_table, kv_len, batch, nq, nkv, head_dim, head_dim_padded, eps, use_norm)
    return attn_decode_paged(tmp, kpool, vpool, block_table, kv_len, 1, max_len, part_o, part_ml, out, batch, nq, nkv, head_dim_padded,
                             softmax_scale)


def quantize_rows_mxfp4(W):
    from oracle import mxfp4
    q, sb = mxfp4.quantize_mxfp4(W)
    return q, mxfp4.permute_scales(sb)


def gemv_w4(A, Wq, Ws, C, *, bias=None, residual=None, epilogue=0, M=None, norm_w=None, eps=0.0):
    from oracle import mxfp4
    N, ng = Ws.shape
    nk = Wq.shape[1] * 2 // 128
    v = Ws.view(N, ng // 16, 4, 4)                                   # [row, group, q, j] -> natural [row, k-step = 4 group + j, q]
    sb = v.permute(0, 1, 3, 2).reshape(N, -1, 4)[:, :nk].reshape(N, nk * 4)
    M = A.shape[0] if M is None else M
    C[:M] = mxfp4.gemv_w4(A[:M], Wq, sb, bias=bias, residual=None if residual is None else residual[:M], swiglu=epilogue == 3,
                          norm_w=norm_w, eps=eps)
    return C


def quantize_nf4(W):
    from oracle import nf4
    return nf4.quantize_nf4(W)


def dequantize_nf4(q, absmax, out=None):
    from oracle import nf4
    w = nf4.dequantize_nf4(q, absmax)
    if out is None:
        return w
    out[:w.shape[0], :w.shape[1]] = w
    return out[:w.shape[0], :w.shape[1]]


def quantize_rows_i8(W):
    s_ = W.float().abs().amax(1) / 127.0
    s_ = torch.where(s_ > 0, s_, torch.ones_like(s_))
    q = (torch.round(W.float() / s_[:, None]).clamp(-127, 127) + 128).to(torch.uint8)
    return q, s_


def gemv_w8(A, Wq, scale, C, *, bias=None, residual=None, epilogue=0, M=None, norm_w=None, eps=0.0):
    return gemv(A, dequantize_rows_i8(Wq, scale), C, bias=bias, residual=residual, epilogue=epilogue, M=M, norm_w=norm_w, eps=eps)


def dequantize_rows_i8(q, scale, out=None):
    w = _bf((q.float() - 128.0) * scale[:, None])
    if out is None:
        return w
    out[:w.shape[0], :w.shape[1]] = w
    return out[:w.shape[0], :w.shape[1]]


def gemv_nf4(A, Wq, absmax, C, *, bias=None, residual=None, epilogue=0, M=None, norm_w=None, eps=0.0):
    """The reference's Linear4bit with bf16 compute: the bf16 projection on the de-quantised bf16 weight (oracle/nf4.py)."""
    from oracle import nf4
    return gemv(A, nf4.dequantize_nf4(Wq, absmax), C, bias=bias, residual=residual, epilogue=epilogue, M=M, norm_w=norm_w, eps=eps)


def decode_advance(next_tok, cur_tok32, tokens_out, pos, kv_len, step, batch, max_steps):
    s_ = int(step[0])
    cur_tok32.copy_(next_tok.to(torch.int32))
    if s_ + 1 < max_steps:
        tokens_out[s_ + 1].copy_(next_tok)
    pos.add_(1)
    kv_len.add_(1)
    step.add_(1)


def rope2d(qkv, tables, pos_ids, nheads, head_dim, head_dim_padded):
    cos_h, sin_h, cos_w, sin_w = (t[pos_ids.long()].float()[:, None, :] for t in tables)      # [rows, 1, hd/2]
    hd, dp, h2, q4 = head_dim, head_dim_padded, head_dim // 2, head_dim // 4
    heads = qkv[:, :nheads * dp].view(qkv.shape[0], nheads, dp)
    for half, (c, s_) in enumerate(((cos_h, sin_h), (cos_w, sin_w))):
        x = heads[:, :, half * h2:(half + 1) * h2].float()
        rot = torch.cat([-x[..., q4:], x[..., :q4]], -1)
        heads[:, :, half * h2:(half + 1) * h2] = _bf(_bf(x * c).float() + _bf(rot * s_).float())
    return qkv


def taylor_update(feature, factors, n_diff, distance):
    cur = feature.clone()
    for i in range(n_diff):
        old = factors[i].clone()
        factors[i].copy_(cur)
        cur = _bf(_r(cur.float() - old.float()) / float(distance))
    factors[n_diff].copy_(cur)


def taylor_eval(factors, n, x, out):
    acc = factors[0].float()
    fact, xp = 1.0, 1.0
    for i in range(1, n):
        fact *= i
        xp *= x
        c = torch.tensor(1.0 / fact, dtype=torch.float32)
        acc = _r(acc + _r(_r(c * factors[i].float()) * torch.tensor(xp, dtype=torch.float32)))
    out.copy_(_bf(acc))
    return out


def attn_varlen_ranges(q, k_new, vt_new, out, q_start, q_end, vt_new_col, batch, max_lq, nq, nkv, head_dim, causal, softmax_scale,
                       k_ctx=None, vt_ctx=None, ctx_start=None, ctx_end=None, vt_ctx_col=None, lse=None):
    D, grp = head_dim, nq // nkv
    qs, qe, ncol = q_start.tolist(), q_end.tolist(), vt_new_col.tolist()
    cs = ctx_start.tolist() if ctx_start is not None else None
    ce = ctx_end.tolist() if ctx_end is not None else None
    ccol = vt_ctx_col.tolist() if vt_ctx_col is not None else None
    for b in range(batch):
        q0, Lq = qs[b], qe[b] - qs[b]
        C = ce[b] - cs[b] if cs is not None else 0
        if Lq <= 0:
            continue
        for h in range(nq):
            g = h // grp
            ks = [k_new[q0:q0 + Lq, g * D:(g + 1) * D]]
            vs = [vt_new[g * D:(g + 1) * D, ncol[b]:ncol[b] + Lq].t()]
            if C > 0:
                ks.insert(0, k_ctx[cs[b]:cs[b] + C, g * D:(g + 1) * D])
                vs.insert(0, vt_ctx[g * D:(g + 1) * D, ccol[b]:ccol[b] + C].t())
            k, v = torch.cat(ks).double(), torch.cat(vs).double()
            s = (q[q0:q0 + Lq, h * D:(h + 1) * D].double() @ k.t()).float() * softmax_scale
            if causal:
                Lk = C + Lq
                s = s.masked_fill(~torch.ones(Lq, Lk).tril(diagonal=Lk - Lq).bool(), float("-inf"))
            out[q0:q0 + Lq, h * D:(h + 1) * D] = _bf((torch.softmax(s, -1).double() @ v).float())
            if lse is not None:                              # log2 of the softmax denominator in the scaled base-2 domain
                lse[h, q0:q0 + Lq] = torch.logsumexp(s.double(), -1).float() * 1.4426950408889634
    return out


def attn_planned(q, k_new, vt_new, out, aplan, softmax_scale, k_ctx=None, vt_ctx=None):
    """Stand-in of ops.attn_planned: the attention the plan describes (its per-sample host arrays), computed densely.  (That the plan's
    ITEMS cover exactly this is pinned by tests/test_attn_plan_cpu.py.)"""
    t = lambda x: torch.tensor(x, dtype=torch.int32)  # noqa: E731
    B = len(aplan.q_len)
    q_end = [a + b for a, b in zip(aplan.q_start, aplan.q_len)]
    kw = {}
    if aplan.has_ctx:
        kw = dict(k_ctx=k_ctx, vt_ctx=vt_ctx, ctx_start=t(aplan.ctx_start), ctx_end=t([a + b for a, b in zip(aplan.ctx_start, aplan.ctx_len)]),
                  vt_ctx_col=t(aplan.vt_ctx_col))
    return attn_varlen_ranges(q, k_new, vt_new, out, t(aplan.q_start), t(q_end), t(aplan.vt_new_col), B, max(aplan.q_len), aplan.nq, aplan.nkv,
                              aplan.head_dim, aplan.causal, softmax_scale, **kw)


def flow_mix(clean, noise, t):
    tt = t.float()[:, None]
    return _bf((1.0 - tt) * clean.float() + tt * noise.float())


def flow_add_rows(seq, rows, temb, temb_ids, pos_table, pos_ids):
    r = rows.long()
    seq[r] = _bf(_bf(seq[r].float() + temb[temb_ids.long()].float()).float() + pos_table[pos_ids.long()].float())
    return seq


def mse_rows(pred, noise, clean, src_rows):
    i = src_rows.long()
    return (pred.float() - (noise[i].float() - clean[i].float())) ** 2


def cross_entropy(logits, labels):
    return F.cross_entropy(logits.float(), labels.long(), reduction="none")


# ---- training backward: stand-ins with the contracts of include/bagel_hip.h ("training backward" section) ----
def transpose(src, dst=None, rows=None, n=None):
    if n is None:
        n = rows.numel() if rows is not None else src.shape[0]
    C = src.shape[1]
    npad = -(-max(n, 1) // 64) * 64
    if dst is None:
        dst = torch.empty((C, npad), dtype=BF16)
    idx = rows.long()[:n] if rows is not None else torch.arange(n)
    dst[:C, :npad] = 0
    dst[:C, :n] = src[idx].t()
    return dst[:C, :npad]


def _by_expert(w0, w1, expert, M):
    if w1 is None or expert is None:
        return w0.float()[None].expand(M, -1), torch.zeros(M, dtype=torch.bool)
    ex = expert.bool()
    return torch.where(ex[:, None], w1.float()[None], w0.float()[None]), ex


def _rms_bwd(x, dy, w, eps):
    """-> (dx fp32, dy * bf16(xh) fp32) for y = w * bf16(x * rsqrt(mean(x^2) + eps))."""
    xf = x.double()
    r = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    xh = xf * r
    dxh = dy.double() * w.double()
    dx = r * (dxh - xh * (dxh * xh).mean(-1, keepdim=True))
    return dx.float(), (dy.double() * _bf(xh.float()).double()).float()


def rmsnorm_bwd(x, dy, w0, g, eps, w1=None, expert=None, accumulate=True):
    M = x.shape[0]
    w, ex = _by_expert(w0, w1, expert, M)
    dx, dwrow = _rms_bwd(x, dy, w, eps)
    base = g.float() if accumulate else torch.zeros_like(dx)
    g.copy_(_bf(base + _bf(dx).float()))
    dw0 = _bf(dwrow[~ex].double().sum(0).float())
    dw1 = _bf(dwrow[ex].double().sum(0).float()) if w1 is not None else None
    return dw0, dw1


def layernorm_bwd(x, dy, w, g, eps, accumulate=True):
    xf = x.double()
    mean = xf.mean(-1, keepdim=True)
    r = torch.rsqrt(((xf - mean) ** 2).mean(-1, keepdim=True) + eps)
    xh = (xf - mean) * r
    dxh = dy.double() * w.double()
    dx = (r * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))).float()
    base = g.float() if accumulate else torch.zeros_like(dx)
    g.copy_(_bf(base + _bf(dx).float()))
    return _bf((dy.double() * xh).sum(0).float()), _bf(dy.double().sum(0).float())


def qknorm_rope_bwd(dqkv, qkv_raw, cos, sin, q_w0, k_w0, q_w1, k_w1, expert, nq, nkv, head_dim, head_dim_padded, eps, use_norm):
    M, hd, dp = dqkv.shape[0], head_dim, head_dim_padded
    half = hd // 2
    heads = dqkv[:, :(nq + nkv) * dp].view(M, nq + nkv, dp)
    raw = qkv_raw[:, :(nq + nkv) * dp].view(M, nq + nkv, dp)[:, :, :hd]
    dy = heads[:, :, :hd].float()
    c, s_ = cos.float()[:, None, :], sin.float()[:, None, :]
    d1, d2 = dy[..., :half], dy[..., half:]
    dn = torch.cat([d1 * c + d2 * s_, d2 * c - d1 * s_], -1)            # gradient of the normalised heads
    outs = (None, None, None, None)
    if use_norm:
        wq, ex = _by_expert(q_w0, q_w1, expert, M)
        wk, _ = _by_expert(k_w0, k_w1, expert, M)
        w = torch.cat([wq[:, None, :].expand(M, nq, hd), wk[:, None, :].expand(M, nkv, hd)], 1)
        dx, dwrow = _rms_bwd(raw, _bf(dn), w, eps)
        dn = dx
        def red(rows, lo, hi):
            return _bf(dwrow[rows][:, lo:hi].double().sum((0, 1)).float())
        two = q_w1 is not None
        outs = (red(~ex, 0, nq), red(~ex, nq, nq + nkv), red(ex, 0, nq) if two else None, red(ex, nq, nq + nkv) if two else None)
    heads[:, :, :hd] = _bf(dn)
    return outs


def swiglu_bwd(gu, d_act):
    M, I = d_act.shape
    v = gu.view(M, I // 16, 2, 16)
    g, u = v[:, :, 0].float(), v[:, :, 1].float()
    d = d_act.view(M, I // 16, 16).float()
    sg = torch.sigmoid(g)
    du = d * _bf(g * sg).float()
    dg = d * u * (sg * (1 + g * (1 - sg)))
    v[:, :, 0] = _bf(dg)
    v[:, :, 1] = _bf(du)
    return gu


def swiglu_fwd(gu, act):
    M, I = act.shape
    v = gu.view(M, I // 16, 2, 16)
    act.copy_(_bf(_bf(F.silu(v[:, :, 0].float())).float() * v[:, :, 1].float()).reshape(M, I))
    return act


def act_bwd(pre, d_out, kind):
    x = pre.float().requires_grad_(True)
    with torch.enable_grad():
        y = F.gelu(x, approximate="tanh") if kind == 1 else F.silu(x)
        (gx,) = torch.autograd.grad(y, x, d_out.float())
    pre.copy_(_bf(gx))
    return pre


def cross_entropy_bwd(logits, labels, d_loss):
    p = torch.softmax(logits.float(), -1)
    lab = labels.long()
    ok = (lab >= 0) & (lab < logits.shape[1])
    p[torch.arange(p.shape[0])[ok], lab[ok]] -= 1.0
    p[~ok] = 0
    logits.copy_(_bf(p * d_loss.float()[:, None]))
    return logits


def mse_rows_bwd(pred, noise, clean, src_rows, d_loss):
    i = src_rows.long()
    return _bf(2.0 * (pred.float() - (noise[i].float() - clean[i].float())) * d_loss.float())


def rows_segment_sum(src, order, seg_off, dst_rows, dst):
    off, o = seg_off.tolist(), order.long()
    for s_, r in enumerate(dst_rows.tolist()):
        dst[r] = _bf(src[o[off[s_]:off[s_ + 1]]].double().sum(0).float())
    return dst


def colsum(src, rows=None, n=None):
    if n is None:
        n = rows.numel() if rows is not None else src.shape[0]
    idx = rows.long()[:n] if rows is not None else torch.arange(n)
    return _bf(src[idx].double().sum(0).float())


def attn_bwd_blockmask(q, k, v, o, d_o, dq, dk, dv, q_items, k_items, noise_bits, nq, nkv, head_dim, softmax_scale, lse=None):
    """An interpreter of the two kernels' work items (include/bagel_hip.h): what the items describe is what is computed, so the host
    code that builds them is checked against the oracle's autograd by the CPU suite."""
    M, D, grp = q.shape[0], head_dim, nq // nkv
    bits = noise_bits.tolist()
    noise = torch.tensor([(bits[c // 64] >> (c % 64)) & 1 for c in range(M)], dtype=torch.bool)
    lse_in, lse = lse, torch.zeros((nq, M), dtype=torch.float64)
    delta = torch.zeros((nq, M), dtype=torch.float64)
    seen_q, seen_k = torch.zeros(M, dtype=torch.int32), torch.zeros(M, dtype=torch.int32)
    for row0, nrows, kstart, sstart, send, causal, t0, t1 in q_items.tolist():
        rows = torch.arange(row0, row0 + nrows)
        seen_q[rows] += 1
        keys = torch.arange(64 * t0, min(64 * t1, M))
        c, r = keys[None, :], rows[:, None]
        allow = (c >= kstart) & (((c < sstart) & ~noise[keys][None, :]) | ((c >= sstart) & (c < send) & ((c <= r) | (causal == 0))))
        for h in range(nq):
            g = h // grp
            qh = q[rows, h * D:(h + 1) * D].double()
            kh, vh = k[keys, g * D:(g + 1) * D].double(), v[keys, g * D:(g + 1) * D].double()
            doh = d_o[rows, h * D:(h + 1) * D].double()
            s_ = (qh @ kh.t()) * softmax_scale
            s_ = s_.masked_fill(~allow, float("-inf"))
            L = torch.logsumexp(s_, -1) if lse_in is None else lse_in[h, rows].double() / 1.4426950408889634
            dl = (doh * o[rows, h * D:(h + 1) * D].double()).sum(-1)
            lse[h, rows], delta[h, rows] = L, dl
            p = torch.exp(s_ - L[:, None])
            ds = _bf((p * (doh @ vh.t() - dl[:, None]) * softmax_scale).float()).double()
            dq[rows, h * D:(h + 1) * D] = _bf((ds @ kh).float())
    for key0, nkeys, qbeg, qend, send, causal, _, _ in k_items.tolist():
        keys = torch.arange(key0, key0 + nkeys)
        seen_k[keys] += 1
        rows = torch.arange(qbeg, qend)
        c, r = keys[None, :], rows[:, None]
        allow = (r >= send) | (c <= r) | (causal == 0)
        for g in range(nkv):
            kh, vh = k[keys, g * D:(g + 1) * D].double(), v[keys, g * D:(g + 1) * D].double()
            dkh, dvh = torch.zeros_like(kh), torch.zeros_like(vh)
            for h in range(g * grp, (g + 1) * grp):
                qh, doh = q[rows, h * D:(h + 1) * D].double(), d_o[rows, h * D:(h + 1) * D].double()
                p = torch.exp((qh @ kh.t()) * softmax_scale - lse[h, rows][:, None]).masked_fill(~allow, 0.0)
                ds = _bf((p * (doh @ vh.t() - delta[h, rows][:, None]) * softmax_scale).float()).double()
                dvh += _bf(p.float()).double().t() @ doh
                dkh += ds.t() @ qh
            dk[keys, g * D:(g + 1) * D] = _bf(dkh.float())
            dv[keys, g * D:(g + 1) * D] = _bf(dvh.float())
    assert bool((seen_q == 1).all()) and bool((seen_k == 1).all()), "attention backward items must cover every row exactly once"
    return dq, dk, dv


def require_gpu_bf16(t, what=""):
    return None


def require_gpu_f32(t, what=""):
    return None


# ---- VAE (fp32 NHWC) ----------------------------------------------------------------------------------------------------
def conv_gemm_f32(x, ld_in, w, ld_w, bias, residual, out, ld_out, B, Hin, Win, Cin, Hout, Wout, Cout, mode):
    M = B * Hout * Wout
    if mode == 0:
        a = torch.as_strided(x, (M, Cin), (ld_in, 1))
        y = (a.double() @ w[:Cout, :Cin].double().t()).float()
    else:
        xi = x.reshape(B, Hin, Win, -1)[..., :Cin].permute(0, 3, 1, 2)
        w4 = w.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)             # tap-major (dy, dx, cin) -> [Cout, Cin, 3, 3]
        if mode == 1:
            y = F.conv2d(xi.double(), w4.double(), padding=1)
        elif mode == 2:
            y = F.conv2d(F.pad(xi.double(), (0, 1, 0, 1)), w4.double(), stride=2)
        else:
            y = F.conv2d(F.interpolate(xi.double(), scale_factor=2.0, mode="nearest"), w4.double(), padding=1)
        assert y.shape[2:] == (Hout, Wout), (y.shape, Hout, Wout)
        y = y.permute(0, 2, 3, 1).reshape(M, Cout).float()
    if bias is not None:
        y = y + bias.float()
    if residual is not None:
        y = y + residual.reshape(M, -1)[:, :Cout]
    o = torch.as_strided(out, (M, ld_out), (ld_out, 1))
    o[:, :Cout] = y
    o[:, Cout:] = 0
    return out


def groupnorm_f32(x, y, workspace, gamma, beta, B, HW, C, groups, eps, swish):
    v = x.reshape(B, HW, C).permute(0, 2, 1)
    o = F.group_norm(v.double(), groups, gamma.double(), beta.double(), eps).float()
    if swish:
        o = o * torch.sigmoid(o)
    y.copy_(o.permute(0, 2, 1).reshape(y.shape))
    return y


def softmax_rows_f32(x, ld, rows, cols, scale):
    v = torch.as_strided(x, (rows, cols), (ld, 1))
    v.copy_(torch.softmax(v * scale, -1))
    return x


def vae_reparam_f32(moments, noise, z, n_pix, z_channels, scale, shift):
    m = moments.reshape(n_pix, -1)
    mean, logvar = m[:, :z_channels], m[:, z_channels:2 * z_channels]
    z.copy_((scale * ((mean + torch.exp(0.5 * logvar) * noise.reshape(n_pix, z_channels)) - shift)).reshape(z.shape))
    return z


# ---- VAE under bf16 autocast (bf16 NHWC): the fp32 stand-ins on up-cast operands, rounded where the kernels round ----------------------
def conv_gemm_bf16(x, ld_in, w, ld_w, bias, residual, out, ld_out, B, Hin, Win, Cin, Hout, Wout, Cout, mode):
    M = B * Hout * Wout
    tmp = torch.zeros((M, ld_out), dtype=torch.float32)
    conv_gemm_f32(x.float(), ld_in, w.float(), ld_w, None if bias is None else bias.float(), None, tmp, ld_out, B, Hin, Win, Cin, Hout, Wout, Cout, mode)
    o = torch.as_strided(out, (M, ld_out), (ld_out, 1))
    if out.dtype == torch.float32:
        o.copy_(tmp)
        return out
    y = _bf(tmp[:, :Cout])
    if residual is not None:
        y = _bf(y.float() + residual.reshape(M, -1)[:, :Cout].float())
    o[:, :Cout] = y
    o[:, Cout:] = 0
    return out


def groupnorm_bf16_workspace_floats(B, C, groups):
    return B * groups * 2050 + B * C * 2


def groupnorm_bf16(x, y, workspace, gamma, beta, B, HW, C, groups, eps, swish):
    t = torch.empty(x.shape, dtype=torch.float32)
    groupnorm_f32(x.float(), t, workspace, gamma, beta, B, HW, C, groups, eps, swish)
    y.copy_(_bf(t))
    return y


def softmax_rows_bf16(x, y, rows, cols, scale):
    y[:rows, :cols] = _bf(torch.softmax(x[:rows, :cols] * scale, -1))
    return y


def vae_reparam_bf16(moments, noise, z, n_pix, z_channels, scale, shift):
    m = moments.reshape(n_pix, -1)
    mean, logvar = m[:, :z_channels], m[:, z_channels:2 * z_channels]            # bf16 tensors: every op below rounds like eager bf16
    zz = mean + torch.exp(0.5 * logvar) * noise.reshape(n_pix, z_channels) if noise is not None else mean
    z.copy_((scale * (zz - shift)).reshape(z.shape))
    return z


def chw_bf16_to_u8(src):
    return ((src * 0.5 + 0.5).clamp(0, 1).permute(1, 2, 0) * 255).to(torch.uint8).contiguous()


def vae_unscale_f32(z, out, n, scale, shift):
    out.copy_(z / scale + shift)
    return out


# ---- image pre/post-processing ---------------------------------------------------------------------------------------------
def resample_u8(src, dst, bounds, kk, vertical):
    a = src.to(torch.int64)
    if vertical:
        for o in range(dst.shape[0]):
            first, n = int(bounds[o, 0]), int(bounds[o, 1])
            ss = (a[first:first + n] * kk[o, :n].to(torch.int64)[:, None, None]).sum(0) + (1 << 21)
            dst[o] = (ss >> 22).clamp(0, 255).to(torch.uint8)
    else:
        for o in range(dst.shape[1]):
            first, n = int(bounds[o, 0]), int(bounds[o, 1])
            ss = (a[:, first:first + n] * kk[o, :n].to(torch.int64)[None, :, None]).sum(1) + (1 << 21)
            dst[:, o] = (ss >> 22).clamp(0, 255).to(torch.uint8)
    return dst


def u8_to_chw_f32(src, mean, std):
    v = src.permute(2, 0, 1).float() / 255.0
    m = torch.tensor([float(x) for x in mean], dtype=torch.float32)[:, None, None]
    s_ = torch.tensor([float(x) for x in std], dtype=torch.float32)[:, None, None]
    return (v - m) / s_


def chw_f32_to_u8(src):
    v = (src.float() * 0.5 + 0.5).clamp(0.0, 1.0) * 255.0
    return v.to(torch.uint8).permute(1, 2, 0).contiguous()


_NAMES = ["gemm", "gemv", "gemv_mb", "gemm_skinny", "rmsnorm", "layernorm", "rope_table", "qknorm_rope", "v_transpose", "attn_varlen",
          "copy_rows", "f32_to_bf16", "timestep_sinusoid", "flow_add", "add_table_rows", "cfg_stage1", "cfg_stage2_euler",
          "argmax", "require_gpu_bf16", "rope2d", "taylor_update", "taylor_eval", "attn_varlen_ranges", "flow_mix", "flow_add_rows",
          "mse_rows", "cross_entropy", "argmax_into", "sample_gumbel_into", "rope_table_into", "decode_qkv_post", "kv_append_paged", "attn_decode_paged", "attn_decode_fused", "quantize_rows_mxfp4", "gemv_w4", "quantize_nf4", "gemv_nf4", "dequantize_nf4", "quantize_rows_i8", "dequantize_rows_i8", "gemv_w8",
          "decode_advance", "require_gpu_f32", "attn_planned", "conv_gemm_f32", "groupnorm_f32", "softmax_rows_f32", "vae_reparam_f32", "conv_gemm_bf16", "groupnorm_bf16", "groupnorm_bf16_workspace_floats", "softmax_rows_bf16", "vae_reparam_bf16", "chw_bf16_to_u8",
          "vae_unscale_f32", "resample_u8", "u8_to_chw_f32", "chw_f32_to_u8", "transpose", "rmsnorm_bwd", "layernorm_bwd", "qknorm_rope_bwd", "swiglu_bwd",
          "act_bwd", "swiglu_fwd", "cross_entropy_bwd", "mse_rows_bwd", "rows_segment_sum", "colsum", "attn_bwd_blockmask"]


def install(monkeypatch):
    """Replace the launch wrappers of bagel_amd.ops by the CPU stand-ins for the duration of one test."""
    from bagel_amd import ops
    g = globals()
    for n in _NAMES:
        monkeypatch.setattr(ops, n, g[n])
    assert math.isfinite(1.0)


def install_permanently():
    """The same replacement for a whole PROCESS (bench.py --standins: the child ranks of tests/test_parallel_cpu.py)."""
    from bagel_amd import ops
    g = globals()
    for n in _NAMES:
        setattr(ops, n, g[n])
