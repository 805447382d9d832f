"""Autoregressive text decode engine: paged KV cache + a per-token step whose whole state lives on the device.

Mirrors the loop of ``Bagel.generate_text`` (bagel.py:930-1000) and the Lq = 1 path of
``Qwen2Model.forward_inference`` (qwen2_navit.py:1018-1092, und mode), re-planned for MI355X:

  * decode is HBM-bound (14.14 GB of und-expert weights per token at 7B), so every projection is the
    skinny weight-streaming kernel (``bagel_gemv_bf16``) with the RMSNorm fused into its activation staging --
    7 launches per layer instead of the reference's ~178 aten ops;
  * the KV cache is paged (64-token pages, block table per sample); a step appends one row per sample in place
    instead of re-allocating and re-scattering the whole cache per layer per token (qwen2_navit.py:563-575);
  * token, position, KV length and step counter are device buffers advanced by ``bagel_decode_advance``, so one step
    is captured ONCE into a hipGraph and replayed per token: the host issues one call per token and only reads a
    token back when the caller asked for an end-token check (bagel.py:996).
"""
import os

import numpy as np
import torch

from ... import ops

BF16 = torch.bfloat16
MB2_FUSED_NORM = os.environ.get("BAGEL_MB2_FUSED_NORM", "0") == "1"


def _ceil_to(x, m):
    return (x + m - 1) // m * m


class PagedKVCache:
    """Per-layer K/V page pools + per-sample block tables.  ``order`` (optional permutation of the page ids) exists
    so tests can prove that nothing assumes physically contiguous pages."""

    PAGE = ops.KV_PAGE

    def __init__(self, num_layers, batch, width, capacity_tokens, device, order=None):
        self.num_layers, self.batch, self.width = num_layers, batch, width
        self.pages_per_sample = max(1, _ceil_to(capacity_tokens, self.PAGE) // self.PAGE)
        self.num_pages = batch * self.pages_per_sample
        rows = self.num_pages * self.PAGE
        self.k = torch.empty((num_layers, rows, width), dtype=BF16, device=device)
        self.v = torch.empty((num_layers, rows, width), dtype=BF16, device=device)
        ids = list(range(self.num_pages)) if order is None else [int(x) for x in order]
        if sorted(ids) != list(range(self.num_pages)):
            raise ValueError("order must be a permutation of the page ids")
        self.block_table_host = [ids[b * self.pages_per_sample:(b + 1) * self.pages_per_sample] for b in range(batch)]
        self.block_table = torch.tensor(self.block_table_host, dtype=torch.int32, device=device)
        self.kv_len = torch.zeros((batch,), dtype=torch.int32, device=device)
        self.capacity = self.pages_per_sample * self.PAGE

    def physical_rows(self, b, start, stop):
        """Pool rows of tokens [start, stop) of sample b."""
        bt = np.asarray(self.block_table_host[b], dtype=np.int64)
        j = np.arange(start, stop, dtype=np.int64)
        return bt[j // self.PAGE] * self.PAGE + j % self.PAGE

    def rows_index(self, ranges):
        """Device int32 index of the pool rows of ``ranges`` = [(b, start, stop), ...], in that order (the same for every layer)."""
        src = [self.physical_rows(b, s, e) for b, s, e in ranges]
        src = np.concatenate(src) if src else np.zeros((0,), dtype=np.int64)
        return torch.from_numpy(src.astype(np.int32)).to(self.k.device)

    def adopt(self, cache, lens):
        """Copy a NaiveCache (merged layout [ctx_0 | ctx_1 | ...], per-sample ``lens``) into the pages."""
        if max(lens) > self.capacity:
            raise ValueError("context longer than the page capacity")
        dst_t = self.rows_index([(b, 0, int(n)) for b, n in enumerate(lens)])
        total = dst_t.numel()
        for li in range(self.num_layers):
            if total:
                ops.copy_rows(cache._k[li], self.k[li], total, self.width, dst_rows=dst_t)
                ops.copy_rows(cache._v[li], self.v[li], total, self.width, dst_rows=dst_t)
        self.kv_len.copy_(torch.tensor([int(n) for n in lens], dtype=torch.int32))

    def gather(self, layer, ranges, idx=None):
        """Rows of ``ranges`` = [(b, start, stop), ...] of one layer -> contiguous (K, V) tensors (``idx`` = rows_index(ranges), when the
        caller gathers the same ranges from every layer)."""
        if idx is None:
            idx = self.rows_index(ranges)
        n = idx.numel()
        k = torch.empty((n, self.width), dtype=BF16, device=self.k.device)
        v = torch.empty_like(k)
        if n:
            ops.copy_rows(self.k[layer], k, n, self.width, src_rows=idx)
            ops.copy_rows(self.v[layer], v, n, self.width, src_rows=idx)
        return k, v


class DecodeSession:
    """One ``generate_text`` call: device-resident loop state, the per-token launch sequence, and its hipGraph."""

    def __init__(self, engine, embed_table, lm_head_weight, cache, kv_lens, start_tokens, position_ids, max_length,
                 page_order=None, weight_quant=None, sampler=None):
        if sampler is not None and (len(sampler) != 3 or sampler[0] != "gumbel" or not float(sampler[1]) > 0.0):
            raise ValueError(f"DecodeSession: sampler={sampler!r}: expected None or ('gumbel', temperature > 0, seed)")
        self.eng = eng = engine
        dev = eng.device
        self.B = B = len(kv_lens)
        self.max_length = int(max_length)
        self.ctx_lens = [int(x) for x in kv_lens]
        self.table, self.head = embed_table, lm_head_weight
        nq, nkv, dp = eng.nq, eng.nkv, eng.dp
        self.width = nkv * dp
        L = len(eng.layers)
        self.max_len = max(self.ctx_lens) + self.max_length   # longest key range a step can see
        self.paged = PagedKVCache(L, B, self.width, self.max_len, dev, order=page_order)
        has_ctx = cache is not None and not cache.is_empty(0) and sum(self.ctx_lens) > 0
        if has_ctx:
            if list(cache.lens(0)) != self.ctx_lens:
                raise ValueError("key_values_lens does not match the KV cache contents")
            self.paged.adopt(cache, self.ctx_lens)
        elif sum(self.ctx_lens) != 0:
            raise ValueError("key_values_lens is non-zero but the KV cache is empty")
        e = lambda *s: torch.empty(s, dtype=BF16, device=dev)  # noqa: E731
        self.x = e(B, eng.H)
        self.h = e(B, eng.H)
        self.qkv = e(B, (nq + 2 * nkv) * dp)
        self.att = e(B, nq * dp)
        self.act = e(B, eng.I)
        self.logits = e(B, lm_head_weight.shape[0])
        self.cos = e(B, eng.hd // 2)
        self.sin = e(B, eng.hd // 2)
        self.part_o, self.part_ml = ops.attn_decode_workspace(B, nq, dp, self.max_len, dev)
        # fp32 K-slice slabs of bagel_gemv_mb_bf16 for the longest row a layer has (the down projection)
        self.mb_ws = torch.empty(max(ops.mb_workspace_floats(eng.H, eng.I), 4), dtype=torch.float32, device=dev) if 1 < B <= ops.MB_MAX_ROWS else None
        self.pos = position_ids.to(device=dev, dtype=torch.long).clone().contiguous()
        st = start_tokens.to(device=dev, dtype=torch.long).contiguous()
        self.cur32 = st.to(torch.int32)
        self.tokens = torch.zeros((self.max_length + 1, B), dtype=torch.long, device=dev)
        self.tokens[0].copy_(st)
        self.next_tok = torch.zeros((B,), dtype=torch.long, device=dev)
        self.step_ctr = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.inv_freq = eng.model.rotary_emb.inv_freq(dev)
        # q/k norm + RoPE + page append run in the prologue of the attention workgroups (16-lane groups own one head each, results
        # handed round through LDS): one launch fewer per layer, bit-identical to the two-kernel form, 3.31 -> 3.23 ms/token under
        # the profiler at 7B (profiles/r02_decode_mfma_attention.log).  BAGEL_DECODE_FUSED=0 keeps decode_qkv_post as its own launch;
        # head dims whose rotate half is not a multiple of 8 elements always take that path.
        # The fused prologue serves head dims 32 / 64 / 128 and needs one 16-byte-lane group per item (G query heads + the new key) out of
        # the 4 x 64 / (dp / 8) groups of a workgroup (G <= 15 at a padded head_dim of 128); everything else takes the two-kernel form.
        self.fused_attention = (os.environ.get("BAGEL_DECODE_FUSED", "1") == "1" and eng.hd in (32, 64, 128)
                                and nq // nkv + 1 <= 4 * (64 // (dp // 8)))
        # weight-only INT8 for the four projections of every layer (option; lm_head stays bf16 like the reference's quantised
        # modes keep it): the engine caches the quantised copies next to the bf16 ones
        # weight-only quantisation of the four projections of every layer (options; lm_head stays bf16 like the reference's quantised
        # modes keep it): "int8_rowwise" = row-wise absmax W8A16, de-quantised on the VALU (csrc/quant.hip); "mxfp4" = OCP-MX FP4 blocks of 32
        # with E8M0 scales x FP8 activations on the block-scaled MFMA (csrc/mxfp4.hip), the counterpart of the reference's NF4 mode.
        # The engine caches the quantised copies next to the bf16 ones.
        if weight_quant is None and getattr(eng, "weight_store", None) is not None:
            weight_quant = eng.weight_store            # a quantised engine decodes on its stored codes (the 4- / 8-bit gemv kernels), not on scratch copies
        if getattr(eng, "weight_store", None) is not None and weight_quant != eng.weight_store:
            raise NotImplementedError(f"the engine holds {eng.weight_store} weights: weight_quant={weight_quant!r} would re-quantise de-quantised copies")
        if weight_quant == "int8":
            from .qwen2_navit import LLM_INT8_NOT_BUILT
            raise NotImplementedError(LLM_INT8_NOT_BUILT)
        if weight_quant not in (None, "int8_rowwise", "mxfp4", "nf4"):
            raise NotImplementedError(f"weight_quant={weight_quant!r}: 'int8_rowwise' (row-wise absmax, W8A16), 'mxfp4' (OCP-MX FP4, W4A8) and 'nf4' "
                                      "(bitsandbytes NF4 blocks of 64, W4A16: the reference's own 4-bit mode) are built")
        self.weight_quant = weight_quant
        if weight_quant == "int8_rowwise" and (eng.H % 16 or eng.I % 16 or (nq * dp) % 16):
            raise NotImplementedError("int8 weights need row lengths that are multiples of 16")
        if weight_quant == "nf4" and (eng.H % 64 or eng.I % 64 or (nq * dp) % 64):
            raise NotImplementedError("nf4 weights need row lengths that are multiples of the 64-weight block")
        if weight_quant == "mxfp4":
            if eng.H % 128 or eng.I % 128 or (nq * dp) % 128:
                raise NotImplementedError("mxfp4 weights need row lengths that are multiples of 128")
            if B > 4:
                raise NotImplementedError("mxfp4 weights: the projection kernel quantises at most 4 activation rows per launch (batch <= 4)")
        self.w8 = self._quantised_weights() if weight_quant else None
        # OPT-IN (BAGEL_DECODE_ENGINE=1; the default is the launch form, which measured faster: csrc/engine.hip, profiles/r05_decode_engine.log).  One
        # request on bf16 weights: o_proj -> norm + gate/up -> down -> the next layer's norm + qkv (the last layer: final norm + lm_head) run as ONE
        # persistent launch per layer on the loader / consumer weight-streaming engine, bit-identical to the gemv launches it replaces; 3 launches per
        # layer instead of 6.  Shapes the engine refuses (ops.decode_engine_supported mirrors every refusal of eng_launch) take the launch form.
        self.engine_mode = bool(B == 1 and weight_quant is None and os.environ.get("BAGEL_DECODE_ENGINE", "0") == "1"
                                and self._engine_phases_supported())
        if self.engine_mode:
            self._eng_words = ops.decode_engine_sync_words(4)
            self.eng_sync = torch.zeros((L, self._eng_words), dtype=torch.int32, device=dev)
            self.eng_status = torch.zeros((4,), dtype=torch.int32, device=dev)
        # a quantised engine: per-layer views of the small tensors (no scratch copies), built once -- not per eager step
        self._layers = eng.layers.small_views() if getattr(eng, "weight_store", None) is not None else eng.layers
        # sampler = None: greedy (argmax) | ("gumbel", temperature, seed): the next token is SAMPLED inside the step (bagel_sample_gumbel_bf16: the
        # categorical distribution of bagel.py:980-983 by Gumbel-max over Philox numbers keyed by seed and the device-side step counter), so a sampled decode
        # replays from the hipGraph like a greedy one
        self.sampler = sampler
        self.steps_done = 0
        self.graph = None
        self.graph_error = None

    def _quantised_weights(self):
        eng = self.eng
        attr = {"int8_rowwise": "_w8_cache", "mxfp4": "_w4_cache", "nf4": "_nf4_cache"}[self.weight_quant]
        cache = getattr(eng, attr, None)
        if cache is None and getattr(eng, "weight_store", None) == self.weight_quant:
            # whole-model load mode: the und expert's stored codes ARE the decode weights
            cache = [{name: st[name][0] for name in ("wqkv", "wo", "wgu", "wd")} for st in eng.layers.stored]
            setattr(eng, attr, cache)
        if cache is None:
            qz = {"int8_rowwise": ops.quantize_rows_i8, "mxfp4": ops.quantize_rows_mxfp4, "nf4": ops.quantize_nf4}[self.weight_quant]
            cache = [dict(wqkv=qz(P.wqkv[0]), wo=qz(P.wo[0]), wgu=qz(P.wgu[0]), wd=qz(P.wd[0])) for P in eng.layers]
            setattr(eng, attr, cache)
        return cache

    # ---- persistent-engine form of the projections of one request -----------------------------------------------
    def _engine_phases(self, li):
        """The chain of layer ``li``: o_proj(+x) -> post-norm + gate/up (SwiGLU) -> down(+x) -> [next layer's input norm + qkv | final norm + lm_head]."""
        eng = self.eng
        layers = eng.layers
        P = layers[li]
        x = self.x
        ph = [dict(A=self.att, W=P.wo[0], C=x, residual=x),
              dict(A=x, W=P.wgu[0], C=self.act, norm_w=P.ln_post[0], epilogue=ops.EPI_SWIGLU16),
              dict(A=self.act, W=P.wd[0], C=x, residual=x)]
        if li + 1 < len(layers):
            Pn = layers[li + 1]
            ph.append(dict(A=x, W=Pn.wqkv[0], C=self.qkv, norm_w=Pn.ln_in[0], bias=Pn.bqkv[0]))
        else:
            ph.append(dict(A=x, W=self.head, C=self.logits, norm_w=eng.model.norm.weight.data))
        return ph

    def _engine_phases_supported(self):
        eng = self.eng
        if getattr(eng, "weight_store", None) is not None:
            return False
        try:
            n = len(eng.layers)
            return all(ops.decode_engine_supported(self._engine_phases(li)) for li in sorted({0, n - 1}))
        except Exception:
            return False

    def check_engine_status(self):
        """Raise if a workgroup of the persistent engine gave up a bounded wait (the outputs of that call are then undefined)."""
        if getattr(self, "engine_mode", False):
            code = int(self.eng_status[0])
            if code:
                raise ops.BagelHipError(f"decode engine: a bounded wait timed out (code 0x{code & 0xff:x}, workgroup {code >> 8}); "
                                        "set BAGEL_DECODE_ENGINE=0 for the launch form")

    # ---- the launch sequence of one token (no host-dependent values: safe to capture) ---------------------------
    def forward_launches(self):
        """embed -> L x [norm+qkv, qk-norm/rope + KV page append, attention (split + combine), o_proj(+res), norm+gate/up (SwiGLU),
        down(+res)]
        -> final norm + lm_head -> argmax."""
        eng, B = self.eng, self.B
        nq, nkv, dp, hd = eng.nq, eng.nkv, eng.dp, eng.hd
        x, qkv, att, act = self.x, self.qkv, self.att, self.act
        pg = self.paged
        scale = hd ** -0.5
        ops.rope_table_into(self.pos, self.inv_freq, self.cos, self.sin)
        ops.copy_rows(self.table, x, B, eng.H, src_rows=self.cur32)
        fused = B == 1      # one request: lane-FMA kernel with the RMSNorm fused in; 2..32: the register-resident MFMA stream (gemv_mb); more: RMSNorm kernel + skinny MFMA GEMM
        h = self.h

        def proj(inp, w, out, norm_w=None, **kw):
            if isinstance(w, tuple):        # int8: (u8 weights, fp32 row scales); mxfp4: (E2M1 codes, E8M0 block scales)
                if self.weight_quant == "mxfp4":
                    return ops.gemv_w4(inp, w[0], w[1], out, norm_w=norm_w, eps=eng.eps, M=B, **kw)
                if self.weight_quant == "nf4":
                    return ops.gemv_nf4(inp, w[0], w[1], out, norm_w=norm_w, eps=eng.eps, M=B, **kw)
                return ops.gemv_w8(inp, w[0], w[1], out, norm_w=norm_w, eps=eng.eps, M=B, **kw)
            if fused:
                return ops.gemv(inp, w, out, norm_w=norm_w, eps=eng.eps, **kw)
            if B <= ops.MB_MAX_ROWS and ops.gemv_mb_supported(inp, w, out, kw.get("bias"), kw.get("residual"), kw.get("epilogue", ops.EPI_NONE),
                                                              norm_w is not None, M=B):
                # 2..32 requests: the weight stream with the activations in registers and the RMSNorm fused (csrc/gemv_mb.hip); the K-slice
                # workspace of the long rows is the session's own, so the captured graph keeps a stable pointer
                if norm_w is not None and B > 16 and not MB2_FUSED_NORM:
                    # two request blocks: every one of the 256 workgroups normalises all 32 rows in its prologue (~8 us of VALU in front of the stream against
                    # ~4.5 at 16 rows); one RMSNorm launch in front of the un-fused stream is shorter (BAGEL_MB2_FUSED_NORM=1 keeps the fused form: same-box A/B)
                    ops.rmsnorm(inp, norm_w, h, eng.eps)
                    return ops.gemv_mb(h, w, out, M=B, workspace=self.mb_ws, **kw)
                return ops.gemv_mb(inp, w, out, norm_w=norm_w, eps=eng.eps, M=B, workspace=self.mb_ws, **kw)
            if norm_w is not None:
                ops.rmsnorm(inp, norm_w, h, eng.eps)
                inp = h
            return ops.gemm(inp, w, out, bias0=kw.get("bias"), residual=kw.get("residual"), epilogue=kw.get("epilogue", ops.EPI_NONE), M0=B)
        if self.engine_mode:
            # qkv of layer 0, then per layer: attention (split + combine) and ONE engine launch; the flag words of every layer are cleared once per token
            self.eng_sync.zero_()
            P0 = eng.layers[0]
            ops.gemv(x, P0.wqkv[0], qkv, norm_w=P0.ln_in[0], eps=eng.eps, bias=P0.bqkv[0])
            for li, P in enumerate(eng.layers):
                if self.fused_attention:
                    ops.attn_decode_fused(qkv, self.cos, self.sin, P.qn[0] if eng.use_norm else None, P.kn[0] if eng.use_norm else None,
                                          pg.k[li], pg.v[li], pg.block_table, pg.kv_len, self.max_len, self.part_o, self.part_ml, att, B,
                                          nq, nkv, hd, dp, eng.eps, eng.use_norm, scale)
                else:
                    ops.decode_qkv_post(qkv, self.cos, self.sin, P.qn[0] if eng.use_norm else None, P.kn[0] if eng.use_norm else None,
                                        pg.k[li], pg.v[li], pg.block_table, pg.kv_len, B, nq, nkv, hd, dp, eng.eps, eng.use_norm)
                    ops.attn_decode_paged(qkv, pg.k[li], pg.v[li], pg.block_table, pg.kv_len, 1, self.max_len, self.part_o,
                                          self.part_ml, att, B, nq, nkv, dp, scale)
                ops.decode_engine(self._engine_phases(li), eng.eps, self.eng_sync[li], self.eng_status)
            self._pick_next()
            return
        for li, P in enumerate(self._layers):
            Q = self.w8[li] if self.w8 is not None else None
            proj(x, Q["wqkv"] if Q else P.wqkv[0], qkv, norm_w=P.ln_in[0], bias=P.bqkv[0])
            if self.fused_attention:
                ops.attn_decode_fused(qkv, self.cos, self.sin, P.qn[0] if eng.use_norm else None, P.kn[0] if eng.use_norm else None,
                                      pg.k[li], pg.v[li], pg.block_table, pg.kv_len, self.max_len, self.part_o, self.part_ml, att, B,
                                      nq, nkv, hd, dp, eng.eps, eng.use_norm, scale)
            else:
                ops.decode_qkv_post(qkv, self.cos, self.sin, P.qn[0] if eng.use_norm else None, P.kn[0] if eng.use_norm else None,
                                    pg.k[li], pg.v[li], pg.block_table, pg.kv_len, B, nq, nkv, hd, dp, eng.eps, eng.use_norm)
                ops.attn_decode_paged(qkv, pg.k[li], pg.v[li], pg.block_table, pg.kv_len, 1, self.max_len, self.part_o,
                                      self.part_ml, att, B, nq, nkv, dp, scale)
            proj(att, Q["wo"] if Q else P.wo[0], x, residual=x)
            proj(x, Q["wgu"] if Q else P.wgu[0], act, norm_w=P.ln_post[0], epilogue=ops.EPI_SWIGLU16)
            proj(act, Q["wd"] if Q else P.wd[0], x, residual=x)
        proj(x, self.head, self.logits, norm_w=eng.model.norm.weight.data)
        self._pick_next()

    def _pick_next(self):
        if self.sampler is None:
            ops.argmax_into(self.logits, self.next_tok)
        else:
            ops.sample_gumbel_into(self.logits, self.next_tok, self.sampler[1], self.sampler[2], self.step_ctr)

    def advance_launch(self):
        ops.decode_advance(self.next_tok, self.cur32, self.tokens, self.pos, self.paged.kv_len, self.step_ctr, self.B,
                           self.max_length + 1)

    # ---- execution ---------------------------------------------------------------------------------------------
    def capture(self, include_advance):
        """Record one step into a hipGraph on a side stream.  Returns False (and keeps the eager path) if the runtime
        refuses; the reason is kept in ``graph_error``."""
        try:
            side = torch.cuda.Stream(device=self.eng.device)
            side.wait_stream(torch.cuda.current_stream())
            with ops.HipGraph.capture(side) as g:
                self.forward_launches()
                if include_advance:
                    self.advance_launch()
            self.graph, self._graph_has_advance = g, include_advance
            return True
        except ops.BagelHipError as err:
            self.graph, self.graph_error = None, str(err)
            return False

    def step(self, sampled=None):
        """Run one token.  ``sampled``: None -> greedy (argmax feeds the next step); else a callable
        logits -> int64 token tensor (host-framework sampling, outside any graph)."""
        if self.steps_done >= self.max_length:
            raise ValueError("DecodeSession: max_length steps already taken")
        if self.graph is not None:
            cur = torch.cuda.current_stream()
            self.graph.stream.wait_stream(cur)
            self.graph.launch()
            cur.wait_stream(self.graph.stream)
            if not self._graph_has_advance:
                if sampled is not None:
                    self.next_tok.copy_(sampled(self.logits))
                self.advance_launch()
        else:
            self.forward_launches()
            if sampled is not None:
                self.next_tok.copy_(sampled(self.logits))
            self.advance_launch()
        self.steps_done += 1

    def tokens_so_far(self):
        """(steps, B) int64: the INPUT token of every step taken (row 0 = the start tokens), bagel.py:942,999."""
        return self.tokens[: self.steps_done].clone()

    def last_token(self, b=0):
        return int(self.tokens[self.steps_done, b])

    def write_back(self, cache):
        """Append the K/V rows of the decoded tokens to the caller's NaiveCache (the reference mutates it in place)."""
        n = self.steps_done
        if n == 0 or cache is None:
            return
        eng = self.eng
        ranges = [(b, self.ctx_lens[b], self.ctx_lens[b] + n) for b in range(self.B)]
        new_dst = ctx_dst = None
        if self.B > 1:
            nd, cd, base = [], [], 0
            for b in range(self.B):
                c = self.ctx_lens[b]
                cd.append(np.arange(base, base + c, dtype=np.int32))
                nd.append(np.arange(base + c, base + c + n, dtype=np.int32))
                base += c + n
            dev = eng.device
            new_dst = torch.from_numpy(np.concatenate(nd)).to(dev)
            ctx_dst = torch.from_numpy(np.concatenate(cd)).to(dev) if base > self.B * n else None
        idx = self.paged.rows_index(ranges)
        for li in range(len(eng.layers)):
            k, v = self.paged.gather(li, ranges, idx)
            cache.store(li, k, v, [n] * self.B, self.ctx_lens, eng.nkv, eng.hd, eng.dp, new_dst, ctx_dst)
