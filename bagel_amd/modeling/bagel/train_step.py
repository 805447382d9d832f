"""The backward half of a training step: ``loss.backward()`` over ``Bagel.forward`` (bagel.py:101-229) without torch autograd inside.

The reference differentiates its eager graph (train/pretrain_unified_navit.py:683-735: bf16 autocast forward, weighted CE / MSE means,
``loss.backward()``).  Here the whole packed forward is ONE autograd node (``PackedTrainStep``): its forward runs the HIP kernels of the
training forward and keeps a tape of five activations per decoder layer; its backward chains hand-written reverse kernels
(csrc/backward.hip, csrc/attention_bwd.hip) and the forward's own GEMM kernel on transposed operand images, and hands the parameter
gradients back to torch -- so ``loss_dict = model(**data); loss.backward()`` of the reference's loop works unchanged and every
``param.grad`` (and whatever hooks FSDP / DDP hang on them) sees the same tensors it would see from autograd.

MI355X-first choices:
  * no activation checkpointing: 288 GB of HBM holds the tape of a 32k-token pack at 7B depth (x_in, raw qkv, attention output, x_mid,
    SwiGLU output per layer = 68 KB per token per layer, 62 GB at 32 768 tokens x 28 layers); only the un-activated gate/up projection
    (76 KB per token per layer on its own) is recomputed, by one more launch of the forward's GEMM;
  * dX = dY W and dW = dY^T X both run on the forward's NT MFMA kernel over transposed bf16 images (bagel_transpose_bf16, an HBM-bound
    pass that costs a few per cent of the product it feeds), including the MoT routing: each expert's dW contracts over that expert's
    rows only (the transpose gathers them), each dX launch routes rows to the two transposed weights exactly like the forward;
  * the block mask (causal / full / noise splits) is differentiated by two deterministic kernels over per-split work items -- no
    atomics, so the same batch gives the same gradients bit for bit on every run.

Scope: gradients of every parameter of the language model (both experts, embeddings, norms, lm_head), llm2vae / vae2llm / the time
embedder, the ViT connector and the SigLIP tower (learned positions or 2-D RoPE); the VAE is frozen as in the reference (``--freeze_vae True``,
pretrain_unified_navit.py:390-393) and the sin-cos position tables are frozen parameters there too -- a parameter outside that set that
requires grad raises instead of silently staying without a gradient."""
import os

import numpy as np
import torch

from ... import ops

BF16 = torch.bfloat16

# Two memory-for-time switches of the backward, both sized for 288 GB of HBM:
#   KEEP_GATE_UP  the tape keeps the un-activated gate/up projection (76 KB per token and layer) and the forward applies SwiGLU as a
#                 kernel of its own, instead of recomputing the largest GEMM of the layer in the backward (3.8 of 31 ms per layer at
#                 18 k tokens).  "auto": when the projections of all layers fit a third of the memory that is free when the tape starts.
#   CACHE_WT      the transposed weight images (the B operands of dX = dY W) stay with the packed layer: +2 bytes per parameter; when the
#                 parameters change (an optimizer step) MoTEngine.refresh() rewrites them IN PLACE together with the packed weights, so a
#                 training loop neither re-allocates them nor pays the transposes inside its backward.
KEEP_GATE_UP = {"0": False, "1": True}.get(os.environ.get("BAGEL_TRAIN_KEEP_GATE_UP", ""), "auto")
CACHE_WT = os.environ.get("BAGEL_TRAIN_CACHE_WT", "1") != "0"


def _ceil_to(x, m):
    return -(-x // m) * m


class AttnBackwardPlan:
    """Work items of bagel_attn_bwd_blockmask_bf16 for one packed batch (include/bagel_hip.h): 128-row query items and 128-key items
    per split, with the sample / split bounds that define the mask of data/data_utils.py:72-103, and the noise-key bitmap."""

    ROWS = 128

    def __init__(self, device, sample_lens, sample_splits):
        M = int(sum(sample_lens))
        q_items, k_items = [], []
        noise = np.zeros(_ceil_to(max(M, 1), 64), dtype=bool)
        row = 0
        for n, (lens, modes) in zip(sample_lens, sample_splits):
            a0, a1, s0 = row, row + n, row
            for L, mode in zip(lens, modes):
                s1 = s0 + L
                causal = int(mode == "causal")
                if mode == "noise":
                    noise[s0:s1] = True
                for r0 in range(s0, s1, self.ROWS):
                    nr = min(self.ROWS, s1 - r0)
                    k_end = r0 + nr if causal else s1                       # a causal chunk sees nothing beyond its last row
                    q_items.append((r0, nr, a0, s0, s1, causal, a0 // 64, -(-k_end // 64)))
                    # who sees these keys: the own split (from the chunk's first row when causal), and -- unless the split is
                    # noise -- every later row of the sample
                    k_items.append((r0, nr, r0 if causal else s0, s1 if mode == "noise" else a1, s1, causal, 0, 0))
                s0 = s1
            row = a1
        self.M = M
        self.q_items = torch.tensor(q_items, dtype=torch.int32, device=device).reshape(-1, 8)
        self.k_items = torch.tensor(k_items, dtype=torch.int32, device=device).reshape(-1, 8)
        words = np.packbits(noise.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1).astype(np.int64)
        self.noise_bits = torch.from_numpy(words.copy()).to(device)


class TrainTape:
    """What the forward leaves behind for the backward (all device tensors stay resident; nothing is recomputed but h = rmsnorm(x),
    the rotated q / k and the un-activated gate / up projection)."""

    def __init__(self):
        self.tp = None          # TrainPlan
        self.x = []             # L + 1 residual streams: x[l] = input of layer l, x[L] = input of the final norm
        self.x_mid = []         # residual stream after the attention block
        self.qkv_raw = []       # fused projection before QK-norm / RoPE
        self.att = []           # attention output (o_proj input)
        self.act = []           # SwiGLU output (down_proj input)
        self.lse = []           # log2 softmax denominators of the attention rows, fp32 [nq, M] (112 bytes per token and layer)
        self.gu = []            # un-activated gate/up projection, only with keep_gate_up (else recomputed in the backward)
        self.keep_gate_up = False
        self.front = {}         # embedding / ViT / latent front end and the loss heads (Bagel._forward_losses)
        self._store, self._pool, self._key = {}, None, None

    # The tape's buffers are RESIDENT and owned by the MODEL (the Qwen2Model module -- it outlives the packed engine, which is refreshed
    # after every optimizer step): a step takes a buffer SET from the model's pool and hands it back when its backward has run, so step
    # after step reuses the same 35-75 GB instead of sending 150+ multi-GB requests through the allocator.  A set is a dict of FLAT
    # buffers with capacity: a request of another shape (the reference's packs have a different sequence_length every step) gets a view
    # of the same storage, which only grows when a larger pack arrives -- the pool holds at most TAPE_POOL_SETS sets whatever shapes come
    # (a forward that starts while the pooled sets are all in use simply allocates one of its own and drops it afterwards).
    TAPE_POOL_SETS = 2

    def begin(self, eng, key):
        self._pool = eng.model.__dict__.setdefault("_tape_pool", [])
        self._key = key
        self._store = self._pool.pop() if self._pool else {}

    def buf(self, name, li, *shape, dtype=BF16, device=None):
        need = 1
        for d in shape:
            need *= int(d)
        k = (name, li, dtype)
        t = self._store.get(k)
        if t is None or t.numel() < need or (device is not None and t.device != torch.device(device)):
            self._store[k] = None                       # free the smaller buffer before the larger one is requested
            t = self._store[k] = torch.empty((max(need, 1),), dtype=dtype, device=device)
        return t[:need].view(*shape)

    def release(self):
        if self._pool is not None and self._store and len(self._pool) < self.TAPE_POOL_SETS:
            self._pool.append(self._store)
        self._store, self._pool = {}, None


def _decide_gate_up(self, eng, M):
    """KEEP_GATE_UP = "auto": decided once per (model, row count), on the first step, when neither the gradients nor the transposed
    weight images exist yet: the kept projections + the rest of the tape + three more copies of the decoder's weights (gradients, the
    weight images, slack for the backward's transients) have to fit 80 % of what is free."""
    if KEEP_GATE_UP != "auto":
        self.keep_gate_up = bool(KEEP_GATE_UP)
        if not self.keep_gate_up:
            for k in [k for k in self._store if k[0] == "gu"]:
                del self._store[k]
        return
    memo = eng.model.__dict__.setdefault("_keep_gate_up", {})        # on the model: survives the engine's refresh after an optimizer step
    if M not in memo:
        if torch.device(eng.device).type != "cuda":
            memo[M] = False
        else:
            L = len(eng.layers)
            free, _ = torch.cuda.mem_get_info(eng.device)
            free += torch.cuda.memory_reserved(eng.device) - torch.cuda.memory_allocated(eng.device)   # what torch's allocator can hand out again
            wbytes = sum(w.numel() * 2 for P in eng.layers for ws in (P.wqkv, P.wo, P.wgu, P.wd) for w in ws)
            base = L * M * 2 * (3 * eng.H + (eng.nq + 2 * eng.nkv) * eng.dp + eng.nq * eng.dp + eng.I)
            memo[M] = L * M * 2 * eng.I * 2 + base + 3 * wbytes <= 0.8 * free
    self.keep_gate_up = memo[M]
    if not self.keep_gate_up:                              # a set taken from the pool may still carry the 'gu' buffers of a step that kept them
        for k in [k for k in self._store if k[0] == "gu"]:
            del self._store[k]


TrainTape.decide_gate_up = _decide_gate_up


def _wt(W):
    """[N, K] -> its transposed image [K, N] (row stride ceil64(N)): the weight operand of dX = dY W on the NT kernel."""
    return ops.transpose(W)[:, :W.shape[0]]


def _wgrad(dY, X, rows=None, n=None):
    """dW [N, K] = sum over the listed rows of dY[r]^T X[r]  (all rows when rows is None)."""
    dYt, Xt = ops.transpose(dY, rows=rows, n=n), ops.transpose(X, rows=rows, n=n)
    dW = torch.empty((dY.shape[1], X.shape[1]), dtype=BF16, device=dY.device)
    return ops.gemm(dYt, Xt, dW)


def _unpad_rows(w, nheads, hd, dp):
    if dp == hd:
        return w
    return w.view(nheads, dp, *w.shape[1:])[:, :hd].reshape(nheads * hd, *w.shape[1:])


def _unpad_cols(w, nheads, hd, dp):
    if dp == hd:
        return w
    return w.view(w.shape[0], nheads, dp)[:, :, :hd].reshape(w.shape[0], nheads * hd)


def _deinterleave_gate_up(w):
    I2, H = w.shape
    v = w.view(I2 // 32, 2, 16, H)
    return v[:, 0].reshape(I2 // 2, H), v[:, 1].reshape(I2 // 2, H)


class _Grads:
    """parameter -> gradient, summed when a parameter collects more than one contribution (tied embeddings)."""

    def __init__(self):
        self.by_id = {}

    def add(self, param, g):
        if param is None or g is None or not param.requires_grad:
            return
        g = g.reshape(param.shape)
        if g.dtype != param.dtype:
            g = g.to(param.dtype)
        k = id(param)
        self.by_id[k] = g.contiguous() if k not in self.by_id else self.by_id[k] + g

    def get(self, param):
        return self.by_id.get(id(param))


def engine_backward_train(eng, tape, g, grads):
    """Reverse of MoTEngine.forward_train: ``g`` [M, H] bf16 = d loss / d (final-norm output), overwritten on the way; returns
    d loss / d (packed input sequence) [M, H] bf16 and adds every decoder-layer / final-norm parameter gradient to ``grads``."""
    tp = tape.tp
    dev = eng.device
    nq, nkv, dp, hd = eng.nq, eng.nkv, eng.dp, eng.hd
    qw, kw_ = nq * dp, nkv * dp
    M, H, I = tp.M, eng.H, eng.I
    two = tp.n_vae > 0
    # what is per modality: everything for Qwen2MoTDecoderLayer; the MLP and the model's final norm for Qwen2MoEDecoderLayer (shared attention and
    # layer norms); nothing for Qwen2DecoderLayer -- the same switches as MoTEngine.forward_train
    two_a, two_m = two and eng.mot, two and eng.moe_mlp
    expert_a, expert_m = (tp.expert if two_a else None), (tp.expert if two_m else None)
    scale = hd ** -0.5
    e = lambda *s: torch.empty(s, dtype=BF16, device=dev)  # noqa: E731
    sel2, sel1 = [(tp.text_idx, tp.n_text), (tp.vae_idx, tp.n_vae)], [(None, M)]
    sel_a, sel_m = (sel2 if two_a else sel1), (sel2 if two_m else sel1)
    sufs = ("", "_moe_gen")

    def groups(w, two_):
        if two_:
            return dict(W0=w[0], bias0=None, a_rows0=tp.text_idx, c_rows0=tp.text_idx, M0=tp.n_text,
                        W1=w[1], bias1=None, a_rows1=tp.vae_idx, c_rows1=tp.vae_idx, M1=tp.n_vae)
        return dict(W0=w[0], bias0=None, M0=M)

    def wgrads(dY, X, need, sel):
        """dW per expert; experts whose parameter is frozen are skipped (``_Grads.add`` ignores None)."""
        return [_wgrad(dY, X, rows, n) if nd else None for (rows, n), nd in zip(sel, need)]

    def wts(P, name, sel):
        ws = getattr(P, name)[:len(sel)]
        if not CACHE_WT:
            return [_wt(w) for w in ws]
        if name not in P.wt or len(P.wt[name]) < len(ws):
            P.wt[name] = [_wt(w) for w in ws]
        return P.wt[name]

    bplan = getattr(tp, "_bplan", None)
    if bplan is None:                    # built once per TrainPlan (host loops + a host-to-device copy), reused by every backward of that pack
        bplan = tp._bplan = AttnBackwardPlan(dev, tp.sample_lens, tp.sample_splits)
    m = eng.model
    # final norm (qwen2_navit.py:1011-1015)
    gx = e(M, H)
    dw0, dw1 = ops.rmsnorm_bwd(tape.x[-1], g, m.norm.weight.data, gx, eng.eps, w1=m.norm_moe_gen.weight.data if two_m else None, expert=expert_m,
                               accumulate=False)
    grads.add(m.norm.weight, dw0)
    if two_m:
        grads.add(m.norm_moe_gen.weight, dw1)
    g = gx
    h, d_h = e(M, H), e(M, H)
    for li in range(len(eng.layers) - 1, -1, -1):
        P, Lm = eng.layers[li], m.layers[li]
        x_in, x_mid, act, att, qkv_raw = tape.x[li], tape.x_mid[li], tape.act[li], tape.att[li], tape.qkv_raw[li]
        # which weight gradients this layer owes, per expert (frozen parameters: no dW launch, no column sum)
        a = Lm.self_attn
        rq = lambda *ps: any(p_ is not None and p_.requires_grad for p_ in ps)  # noqa: E731
        need_qkv = [rq(*(getattr(a, n_ + sufs[ei]).weight for n_ in ("q_proj", "k_proj", "v_proj"))) for ei in range(len(sel_a))]
        need_bqkv = [rq(*(getattr(a, n_ + sufs[ei]).bias for n_ in ("q_proj", "k_proj", "v_proj"))) for ei in range(len(sel_a))]
        need_o = [rq(getattr(a, "o_proj" + sufs[ei]).weight) for ei in range(len(sel_a))]
        need_gu = [rq(getattr(Lm, "mlp" + sufs[ei]).gate_proj.weight, getattr(Lm, "mlp" + sufs[ei]).up_proj.weight) for ei in range(len(sel_m))]
        need_d = [rq(getattr(Lm, "mlp" + sufs[ei]).down_proj.weight) for ei in range(len(sel_m))]
        # ---- MLP block: x_out = x_mid + down(swiglu(gate_up(rmsnorm(x_mid))))   (qwen2_navit.py:744-753; MoE kind :873-881; dense :641-644)
        d_act = e(M, I)
        ops.gemm(g, C=d_act, **groups(wts(P, "wd", sel_m), two_m))
        dWd = wgrads(g, act, need_d, sel_m)
        ops.rmsnorm(x_mid, P.ln_post[0], h, eng.eps, w1=P.ln_post[1] if two_a else None, expert=expert_a)
        if tape.gu:
            gu, tape.gu[li] = tape.gu[li], None                  # kept by the forward; consumed (overwritten with its gradient) here
        else:
            gu = e(M, 2 * I)
            ops.gemm(h, C=gu, **groups(P.wgu, two_m))            # the un-activated projection, recomputed
        ops.swiglu_bwd(gu, d_act)
        del d_act
        dWgu = wgrads(gu, h, need_gu, sel_m)
        ops.gemm(gu, C=d_h, **groups(wts(P, "wgu", sel_m), two_m))
        del gu
        dpost = ops.rmsnorm_bwd(x_mid, d_h, P.ln_post[0], g, eng.eps, w1=P.ln_post[1] if two_a else None, expert=expert_a)
        # ---- attention block: x_mid = x_in + o(attn(rope(qknorm(qkv(rmsnorm(x_in))))))   (qwen2_navit.py:406-497, 713-743; shared form :252-320)
        d_att = e(M, qw)
        ops.gemm(g, C=d_att, **groups(wts(P, "wo", sel_a), two_a))
        dWo = wgrads(g, att, need_o, sel_a)
        qkv = qkv_raw.clone()
        ops.qknorm_rope(qkv, tp.cos, tp.sin, P.qn[0] if eng.use_norm else None, P.kn[0] if eng.use_norm else None,
                        P.qn[1] if (eng.use_norm and two_a) else None, P.kn[1] if (eng.use_norm and two_a) else None,
                        expert_a, nq, nkv, hd, dp, eng.eps, gen_mode=False, use_norm=eng.use_norm)
        dqkv = e(M, qw + 2 * kw_)
        ops.attn_bwd_blockmask(qkv[:, :qw], qkv[:, qw:qw + kw_], qkv[:, qw + kw_:], att, d_att, dqkv[:, :qw], dqkv[:, qw:qw + kw_],
                               dqkv[:, qw + kw_:], bplan.q_items, bplan.k_items, bplan.noise_bits, nq, nkv, dp, scale, lse=tape.lse[li])
        del qkv, d_att
        dqn = ops.qknorm_rope_bwd(dqkv, qkv_raw, tp.cos, tp.sin, P.qn[0] if eng.use_norm else None, P.kn[0] if eng.use_norm else None,
                                  P.qn[1] if (eng.use_norm and two_a) else None, P.kn[1] if (eng.use_norm and two_a) else None,
                                  expert_a, nq, nkv, hd, dp, eng.eps, eng.use_norm)
        ops.rmsnorm(x_in, P.ln_in[0], h, eng.eps, w1=P.ln_in[1] if two_a else None, expert=expert_a)
        dWqkv = wgrads(dqkv, h, need_qkv, sel_a)
        dbqkv = [ops.colsum(dqkv, rows, n) if nd else None for (rows, n), nd in zip(sel_a, need_bqkv)]
        ops.gemm(dqkv, C=d_h, **groups(wts(P, "wqkv", sel_a), two_a))
        del dqkv
        din = ops.rmsnorm_bwd(x_in, d_h, P.ln_in[0], g, eng.eps, w1=P.ln_in[1] if two_a else None, expert=expert_a)
        # ---- unpack the MI355X layouts into the reference's parameter shapes
        for ei in range(len(sel_m)):
            mlp = getattr(Lm, "mlp" + sufs[ei])
            dg, du = (None, None) if dWgu[ei] is None else _deinterleave_gate_up(dWgu[ei])
            grads.add(mlp.gate_proj.weight, dg)
            grads.add(mlp.up_proj.weight, du)
            grads.add(mlp.down_proj.weight, dWd[ei])
        for ei in range(len(sel_a)):
            s = sufs[ei]
            cut = lambda t: (None, None, None) if t is None else (t[:qw], t[qw:qw + kw_], t[qw + kw_:])  # noqa: E731
            (wq, wk, wv), (bq, bk, bv) = cut(dWqkv[ei]), cut(dbqkv[ei])
            for name, w_, b_, nh in (("q_proj", wq, bq, nq), ("k_proj", wk, bk, nkv), ("v_proj", wv, bv, nkv)):
                lin = getattr(a, name + s)
                grads.add(lin.weight, None if w_ is None else _unpad_rows(w_, nh, hd, dp))
                grads.add(lin.bias, None if b_ is None else _unpad_rows(b_, nh, hd, dp))
            grads.add(getattr(a, "o_proj" + s).weight, None if dWo[ei] is None else _unpad_cols(dWo[ei], nq, hd, dp))
            if eng.use_norm:
                grads.add(getattr(a, "q_norm" + s).weight, dqn[2 * ei])
                grads.add(getattr(a, "k_norm" + s).weight, dqn[2 * ei + 1])
            grads.add(getattr(Lm, "input_layernorm" + s).weight, din[ei])
            grads.add(getattr(Lm, "post_attention_layernorm" + s).weight, dpost[ei])
    return g


def siglip_backward(vit, tape, g, grads):
    """Reverse of SiglipVisionModel.forward (siglip_navit.py:145-402: patch embedding + learned positions, pre-LN encoder layers with
    full attention inside every image, GELU-tanh MLP, post-LayerNorm): ``g`` [n, D] bf16 = d loss / d (tower output), overwritten;
    every tower parameter's gradient goes to ``grads``.  Same building blocks as the decoder's reverse (one "full" split per image for
    the attention items, dX / dW on the forward GEMM over transposed images), dense: no expert routing."""
    P = vit._packed
    cfg, vm = vit.config, vit.vision_model
    dev = g.device
    D, I, nh, dp, hd = cfg.hidden_size, cfg.intermediate_size, P["nh"], P["dp"], P["hd"]
    qw, n, eps = nh * dp, tape["n"], cfg.layer_norm_eps
    e = lambda *s: torch.empty(s, dtype=BF16, device=dev)  # noqa: E731
    bplan = AttnBackwardPlan(dev, tape["lens"], [([l], ["full"]) for l in tape["lens"]])
    rope_inv, pos_dev = None, None
    if P["rope"] is not None:            # 2-D RoPE variant (siglip_navit.py:102-142,224-230): the reverse of a rotation is the rotation by -angle
        ch, sh, cw, sw = P["rope"]
        rope_inv = (ch, (-sh.float()).to(BF16), cw, (-sw.float()).to(BF16))
        pos_dev = tape["pos"].to(device=dev, dtype=torch.long).contiguous()
    pl = vm.post_layernorm
    gx = e(n, D)
    dw, db = ops.layernorm_bwd(tape["x"][-1], g, pl.weight.data, gx, eps, accumulate=False)
    grads.add(pl.weight, dw); grads.add(pl.bias, db)
    g = gx
    h, d_h = e(n, D), e(n, D)
    for li in range(len(P["layers"]) - 1, -1, -1):
        L, Lm = P["layers"][li], vm.encoder.layers[li]
        x_in, x_mid, qkv, att, mid = tape["x"][li], tape["x_mid"][li], tape["qkv"][li], tape["att"][li], tape["mid"][li]
        # MLP: x_out = x_mid + fc2(gelu(fc1(ln2(x_mid))))
        grads.add(Lm.mlp.fc2.weight, _wgrad(g, mid)); grads.add(Lm.mlp.fc2.bias, ops.colsum(g))
        d_mid = e(n, I)
        ops.gemm(g, _wt(L["fc2"][0]), d_mid)
        ops.layernorm(x_mid, L["ln2"][0], L["ln2"][1], h, eps)
        pre = e(n, I)
        ops.gemm(h, L["fc1"][0], pre, bias0=L["fc1"][1])                  # the un-activated fc1 output, recomputed
        ops.act_bwd(pre, d_mid, ops.EPI_GELU_TANH)
        del d_mid
        grads.add(Lm.mlp.fc1.weight, _wgrad(pre, h)); grads.add(Lm.mlp.fc1.bias, ops.colsum(pre))
        ops.gemm(pre, _wt(L["fc1"][0]), d_h)
        del pre
        dw, db = ops.layernorm_bwd(x_mid, d_h, L["ln2"][0], g, eps)
        grads.add(Lm.layer_norm2.weight, dw); grads.add(Lm.layer_norm2.bias, db)
        # attention: x_mid = x_in + out_proj(attn(qkv(ln1(x_in))))
        a = Lm.self_attn
        grads.add(a.out_proj.weight, _unpad_cols(_wgrad(g, att), nh, hd, dp)); grads.add(a.out_proj.bias, ops.colsum(g))
        d_att = e(n, qw)
        ops.gemm(g, _wt(L["wo"]), d_att)
        dqkv = e(n, 3 * qw)
        ops.attn_bwd_blockmask(qkv[:, :qw], qkv[:, qw:2 * qw], qkv[:, 2 * qw:], att, d_att, dqkv[:, :qw], dqkv[:, qw:2 * qw], dqkv[:, 2 * qw:],
                               bplan.q_items, bplan.k_items, bplan.noise_bits, nh, nh, dp, hd ** -0.5, lse=tape["lse"][li])
        del d_att
        if rope_inv is not None:
            ops.rope2d(dqkv, rope_inv, pos_dev, 2 * nh, hd, dp)            # gradients of the rotated q / k heads -> of the projection's
        ops.layernorm(x_in, L["ln1"][0], L["ln1"][1], h, eps)
        dW, dB = _wgrad(dqkv, h), ops.colsum(dqkv)
        for j, name in enumerate(("q_proj", "k_proj", "v_proj")):
            lin = getattr(a, name)
            grads.add(lin.weight, _unpad_rows(dW[j * qw:(j + 1) * qw], nh, hd, dp))
            grads.add(lin.bias, _unpad_rows(dB[j * qw:(j + 1) * qw], nh, hd, dp))
        ops.gemm(dqkv, _wt(L["wqkv"]), d_h)
        del dqkv
        dw, db = ops.layernorm_bwd(x_in, d_h, L["ln1"][0], g, eps)
        grads.add(Lm.layer_norm1.weight, dw); grads.add(Lm.layer_norm1.bias, db)
    # front: x0 = patch_embedding(pixels) + position_embedding[pos]  (no position table in the RoPE variant)
    emb = vm.embeddings
    grads.add(emb.patch_embedding.weight, _wgrad(g, tape["a16"])[:, :P["kin"]])
    grads.add(emb.patch_embedding.bias, ops.colsum(g))
    pe = getattr(getattr(emb, "position_embedding", None), "weight", None)
    if pe is not None and pe.requires_grad and P["rope"] is None:
        pos = tape["pos"]
        order = torch.argsort(pos, stable=True)
        uniq, counts = torch.unique_consecutive(pos[order], return_counts=True)
        seg = torch.zeros(uniq.numel() + 1, dtype=torch.int32)
        seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
        i32 = lambda t: t.to(device=dev, dtype=torch.int32)  # noqa: E731
        dE = torch.zeros(tuple(pe.shape), dtype=BF16, device=dev)
        ops.rows_segment_sum(g, i32(order), i32(seg), i32(uniq), dE)
        grads.add(pe, dE)
    return g


class PackedTrainStep(torch.autograd.Function):
    """``Bagel.forward`` as one autograd node: forward = the HIP training forward with a tape, backward = the hand-chained reverse."""

    @staticmethod
    def forward(ctx, model, kwargs, *params):
        tape = TrainTape()
        out = model._forward_losses(tape=tape, **kwargs)
        ctx.model, ctx.tape, ctx.params = model, tape, params
        ctx.has = (out["mse"] is not None, out["ce"] is not None)
        dev = model.device
        empty = lambda: torch.zeros((0,), dtype=torch.float32, device=dev)  # noqa: E731
        return (out["mse"] if ctx.has[0] else empty()), (out["ce"] if ctx.has[1] else empty())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_mse, d_ce):
        if ctx.tape is None:
            raise RuntimeError("the tape of this training step was already consumed (backward runs once per forward)")
        grads = ctx.model._backward_losses(ctx.tape, d_mse if ctx.has[0] else None, d_ce if ctx.has[1] else None)
        ctx.tape.release()
        ctx.tape = None
        return (None, None) + tuple(grads.get(p) for p in ctx.params)
