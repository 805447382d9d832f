"""CPU restatement of BAGEL's unified forward path (the ORACLE).

TEST INFRASTRUCTURE ONLY -- never imported by ``bagel_amd`` (the product).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it, and only as the
checker / reported baseline.

Parity status: PINNED.  Every function below is checked bit-for-bit (torch CPU, bf16 weights,
explicit casts == ``torch.autocast('cpu', bf16)``) against the unmodified reference imported from
/root/reference by ``oracle/make_golden.py``; the resulting vectors are committed under
``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py`` on every box.
(The reference itself ships no tests or golden vectors -- SURVEY.md section 4.)

Style: functional, over a flat ``{state_dict_key: tensor}`` weight map ``W`` (reference key names,
SURVEY.md Appendix B) and plain config dicts (oracle/configs.py).  All cast points follow the
reference line by line; each function cites the lines it restates.
"""
import importlib.util
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def _explicit_casts(fn):
    """The oracle spells out every cast; make sure an ambient autocast region cannot add more."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with torch.set_grad_enabled(GRAD_ENABLED), torch.autocast("cpu", enabled=False):
            return fn(*a, **k)
    return wrapped


# Gradient oracle of the training step (oracle/make_golden_train_grads.py, tests/test_train_backward_*.py): with GRAD_ENABLED the
# restatement keeps torch's autograd graph, so ``loss.backward()`` on its losses yields the gradients the reference's own
# ``loss.backward()`` (train/pretrain_unified_navit.py:690-735) produces -- the explicit casts are differentiable ``.to()`` calls at
# the places autocast puts them.  Off everywhere else (the forward oracle never builds a graph).
GRAD_ENABLED = False

_spec = importlib.util.spec_from_file_location(
    "_oracle_flash_attn", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shims", "flash_attn", "__init__.py"))
_fa = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_fa)
attn_varlen = _fa.flash_attn_varlen_func   # definition of flash_attn_varlen_func (qwen2_navit.py:579-588)


# ----------------------------------------------------------------------------------------------
# elementary ops
# ----------------------------------------------------------------------------------------------
# Noise-floor probe (oracle/make_golden_wide.py): with LINEAR_FP32_ACCUM the SAME bf16-rounded operands are multiplied with an fp32
# matmul and the result is rounded to bf16 once -- the same mathematics and rounding points as the reference's bf16 F.linear, another
# summation order (what a different CPU backend, or a GPU, does).  How far the reference's outputs move under this switch is the
# size of its own accumulation-order noise; the parity tolerances at 7B width are derived from it.  Never on in a parity check.
LINEAR_FP32_ACCUM = False

# FP8 option of the product (model.gen_weight_quant = "fp8", oracle/fp8.py): data_ptr()s of the weight tensors whose linear runs on
# e4m3 operands with row-wise scales (the gen expert's q/k/v/o/gate/up/down projections).  Empty = the reference's arithmetic.
FP8_WEIGHT_PTRS = set()
# ... and, inside a denoise loop, the DELAYED row scales of the SwiGLU output (oracle/fp8.py DelayedScales; None = every input gets its exact row-wise scale)
FP8_DELAYED = None


def fp8_gen_weight_ptrs(W):
    """The weights the product quantises under gen_weight_quant='fp8': every *_moe_gen projection of the decoder layers."""
    names = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
    return {v.data_ptr() for k, v in W.items() if k.endswith(".weight") and "language_model.model.layers." in k
            and any(k.endswith(f"{n}_moe_gen.weight") or (f"mlp_moe_gen.{n}.weight" in k) for n in names)}


# MXFP4 option of the product's text decode (generate_text(weight_quant="mxfp4"), oracle/mxfp4.py): data_ptr()s of the weight tensors
# whose linear runs on OCP-MX FP4 weights x FP8 activations (the und expert's q/k/v/o/gate/up/down projections; lm_head stays bf16).
# Only meaningful around the decode loop: the product's prefill uses the bf16 weights.
MXFP4_WEIGHT_PTRS = set()


def mxfp4_decode_weight_ptrs(W):
    names = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")
    return {v.data_ptr() for k, v in W.items() if "language_model.model.layers." in k and any(k.endswith(f".{n}.weight") for n in names)}


# NF4 option of the product's text decode (generate_text(weight_quant="nf4"), oracle/nf4.py = the reference's own 4-bit load mode,
# app.py:114-125): data_ptr()s of the weight tensors whose linear runs on the de-quantised NF4 weight (same seven linears per layer).
NF4_WEIGHT_PTRS = set()


@_explicit_casts
def linear(x, w, b=None):
    """F.linear under bf16 autocast: inputs cast to the (bf16) weight dtype, bf16 result."""
    if NF4_WEIGHT_PTRS and w.data_ptr() in NF4_WEIGHT_PTRS:
        from oracle import nf4 as NF
        cache = linear.__dict__.setdefault("_nf4", {})
        key = (w.data_ptr(), w._version)
        if key not in cache:
            cache[key] = NF.dequantize_nf4(*NF.quantize_nf4(w), dtype=w.dtype)
        return F.linear(x.to(w.dtype), cache[key], None if b is None else b.to(w.dtype))
    if MXFP4_WEIGHT_PTRS and w.data_ptr() in MXFP4_WEIGHT_PTRS:
        from oracle import mxfp4 as MX
        x2 = x.to(w.dtype).reshape(-1, x.shape[-1])
        cache = linear.__dict__.setdefault("_mx", {})
        key = (w.data_ptr(), w._version)
        if key not in cache:
            cache[key] = MX.quantize_mxfp4(w)
        return MX.gemv_w4(x2, *cache[key], bias=None if b is None else b.to(w.dtype)).reshape(*x.shape[:-1], w.shape[0])
    if FP8_WEIGHT_PTRS and w.data_ptr() in FP8_WEIGHT_PTRS:
        from oracle import fp8 as F8
        x2 = x.to(w.dtype).reshape(-1, x.shape[-1])
        if FP8_DELAYED is not None and w.data_ptr() in FP8_DELAYED.ptrs:
            qa, sa = FP8_DELAYED.quantize(w.data_ptr(), x2)
        else:
            qa, sa = F8.quantize_rows_fp8(x2)
        qw, sw = F8.quantize_rows_fp8(w)
        return F8.gemm_fp8(qa, sa, qw, sw, bias=None if b is None else b.to(w.dtype)).reshape(*x.shape[:-1], w.shape[0])
    if LINEAR_FP32_ACCUM:
        y = x.to(w.dtype).float() @ w.float().t()
        if b is not None:
            y = y + b.to(w.dtype).float()
        return y.to(w.dtype)
    return F.linear(x.to(w.dtype), w, None if b is None else b.to(w.dtype))


@_explicit_casts
def rmsnorm(x, w, eps):
    """Qwen2RMSNorm.forward, modeling_qwen2.py:54-59 (weight multiplies AFTER the cast back)."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


@_explicit_casts
def rope_tables(position_ids, head_dim, theta, dtype):
    """Qwen2RotaryEmbedding.forward, modeling_qwen2.py:130-150 ('default' rope, scaling 1.0)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    pos = position_ids[None, :]
    inv = inv_freq[None, :, None].float().expand(pos.shape[0], -1, 1)
    freqs = (inv.float() @ pos[:, None, :].float()).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos() * 1.0, emb.sin() * 1.0
    return cos.to(dtype).squeeze(0), sin.to(dtype).squeeze(0)


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """apply_rotary_pos_emb, modeling_qwen2.py:162-186 with unsqueeze_dim=1."""
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


@_explicit_casts
def silu_mlp(x, W, p):
    """Qwen2MLP.forward, modeling_qwen2.py:200-201."""
    return linear(F.silu(linear(x, W[p + ".gate_proj.weight"])) * linear(x, W[p + ".up_proj.weight"]),
                  W[p + ".down_proj.weight"])


def sincos_2d_table(embed_dim, grid_size):
    """get_2d_sincos_pos_embed, modeling_utils.py:24-66 ('w goes first'); fp64 omega, fp32 table."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])

    def one(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float64)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float()


@_explicit_casts
def timestep_embed(t, W, freq_dim=256):
    """TimestepEmbedder.forward, modeling_utils.py:88-110 (fp32 sinusoid, autocast MLP)."""
    half = freq_dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    h = linear(emb, W["time_embedder.mlp.0.weight"], W["time_embedder.mlp.0.bias"])
    h = F.silu(h)
    return linear(h, W["time_embedder.mlp.2.weight"], W["time_embedder.mlp.2.bias"])


# ----------------------------------------------------------------------------------------------
# KV cache (NaiveCache, qwen2_navit.py:207-221)
# ----------------------------------------------------------------------------------------------
class OracleCache:
    def __init__(self, num_layers):
        self.key_cache = {i: None for i in range(num_layers)}
        self.value_cache = {i: None for i in range(num_layers)}

    def clone(self):
        c = OracleCache(len(self.key_cache))
        for i in self.key_cache:
            if self.key_cache[i] is not None:
                c.key_cache[i] = self.key_cache[i].clone()
                c.value_cache[i] = self.value_cache[i].clone()
        return c


# ----------------------------------------------------------------------------------------------
# MoT backbone
# ----------------------------------------------------------------------------------------------
@_explicit_casts
def mot_attention(W, pre, cfg, layer_idx, x, query_lens, cos_sin, q_idx, cache, kv_lens, kv_idx,
                  update, causal, mode, vae_idx, text_idx):
    """PackedAttentionMoT.forward_inference, qwen2_navit.py:499-600."""
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = cfg["hidden_size"] // nh
    eps = cfg["rms_norm_eps"]
    a = pre + ".self_attn."
    if mode == "und":
        q = linear(x, W[a + "q_proj.weight"], W[a + "q_proj.bias"]).view(-1, nh, hd)
        k = linear(x, W[a + "k_proj.weight"], W[a + "k_proj.bias"]).view(-1, nkv, hd)
        v = linear(x, W[a + "v_proj.weight"], W[a + "v_proj.bias"]).view(-1, nkv, hd)
        q = rmsnorm(q, W[a + "q_norm.weight"], eps)
        k = rmsnorm(k, W[a + "k_norm.weight"], eps)
    else:
        x = x.to(BF16)
        q = x.new_zeros((x.shape[0], nh * hd))
        k = x.new_zeros((x.shape[0], nkv * hd))
        v = x.new_zeros((x.shape[0], nkv * hd))
        xt, xv = x[text_idx], x[vae_idx]
        q[text_idx] = linear(xt, W[a + "q_proj.weight"], W[a + "q_proj.bias"])
        q[vae_idx] = linear(xv, W[a + "q_proj_moe_gen.weight"], W[a + "q_proj_moe_gen.bias"])
        k[text_idx] = linear(xt, W[a + "k_proj.weight"], W[a + "k_proj.bias"])
        k[vae_idx] = linear(xv, W[a + "k_proj_moe_gen.weight"], W[a + "k_proj_moe_gen.bias"])
        v[text_idx] = linear(xt, W[a + "v_proj.weight"], W[a + "v_proj.bias"])
        v[vae_idx] = linear(xv, W[a + "v_proj_moe_gen.weight"], W[a + "v_proj_moe_gen.bias"])
        q, k, v = q.view(-1, nh, hd), k.view(-1, nkv, hd), v.view(-1, nkv, hd)
        q = q.to(torch.float32)
        q[text_idx] = rmsnorm(q[text_idx], W[a + "q_norm.weight"], eps)
        q[vae_idx] = rmsnorm(q[vae_idx], W[a + "q_norm_moe_gen.weight"], eps)
        k = k.to(torch.float32)
        k[text_idx] = rmsnorm(k[text_idx], W[a + "k_norm.weight"], eps)
        k[vae_idx] = rmsnorm(k[vae_idx], W[a + "k_norm_moe_gen.weight"], eps)

    cos, sin = cos_sin
    q, k = apply_rope(q, k, cos, sin)
    q, k, v = q.to(BF16), k.to(BF16), v.to(BF16)

    if cache is not None and cache.key_cache[layer_idx] is not None:
        pk, pv = cache.key_cache[layer_idx], cache.value_cache[layer_idx]
        total = int(sum(query_lens)) + int(sum(kv_lens))
        mk = pk.new_zeros((total, nkv, hd))
        mv = pk.new_zeros((total, nkv, hd))
        mk[q_idx] = k
        mk[kv_idx] = pk
        mv[q_idx] = v
        mv[kv_idx] = pv
        klens = kv_lens + query_lens
    else:
        mk, mv, klens = k, v, query_lens

    cu_q = F.pad(torch.cumsum(query_lens, 0), (1, 0)).to(torch.int32)
    cu_k = F.pad(torch.cumsum(klens, 0), (1, 0)).to(torch.int32)
    o = attn_varlen(q, mk, mv, cu_q, cu_k, int(query_lens.max()), int(klens.max()), causal=causal)
    o = o.reshape(-1, nh * hd)
    if mode == "und":
        o = linear(o, W[a + "o_proj.weight"])
    else:
        o[text_idx] = linear(o[text_idx], W[a + "o_proj.weight"])
        o[vae_idx] = linear(o[vae_idx], W[a + "o_proj_moe_gen.weight"])
    if update:
        cache.key_cache[layer_idx] = mk
        cache.value_cache[layer_idx] = mv
    return o


@_explicit_casts
def mot_layer(W, cfg, layer_idx, x, query_lens, cos_sin, q_idx, cache, kv_lens, kv_idx, update, causal,
              mode, vae_idx, text_idx):
    """Qwen2MoTDecoderLayer.forward_inference, qwen2_navit.py:757-831 (TaylorSeer off); with ``cfg['layer_module']`` also
    Qwen2DecoderLayer (:648-684, no modality routing at all) and Qwen2MoEDecoderLayer (:885-933, shared attention with the
    und cast points, per-modality MLP only)."""
    pre = f"language_model.model.layers.{layer_idx}"
    eps = cfg["rms_norm_eps"]
    kind = cfg.get("layer_module", "Qwen2MoTDecoderLayer")
    if kind != "Qwen2MoTDecoderLayer":
        res = x
        h = rmsnorm(x, W[pre + ".input_layernorm.weight"], eps)
        h = mot_attention(W, pre, cfg, layer_idx, h, query_lens, cos_sin, q_idx, cache, kv_lens, kv_idx, update, causal, "und",
                          None, None)
        x = res + h
        res = x
        h = rmsnorm(x, W[pre + ".post_attention_layernorm.weight"], eps)
        if kind == "Qwen2MoEDecoderLayer" and mode == "gen":
            o = torch.zeros_like(h).to(BF16)
            o[text_idx] = silu_mlp(h[text_idx], W, pre + ".mlp")
            o[vae_idx] = silu_mlp(h[vae_idx], W, pre + ".mlp_moe_gen")
            h = o
        else:
            h = silu_mlp(h, W, pre + ".mlp")
        return res + h
    res = x
    if mode == "und":
        h = rmsnorm(x, W[pre + ".input_layernorm.weight"], eps)
    else:
        h = torch.zeros_like(x)
        h[text_idx] = rmsnorm(x[text_idx], W[pre + ".input_layernorm.weight"], eps)
        h[vae_idx] = rmsnorm(x[vae_idx], W[pre + ".input_layernorm_moe_gen.weight"], eps)
    h = mot_attention(W, pre, cfg, layer_idx, h, query_lens, cos_sin, q_idx, cache, kv_lens, kv_idx,
                      update, causal, mode, vae_idx, text_idx)
    x = res + h
    res = x
    if mode == "und":
        h = rmsnorm(x, W[pre + ".post_attention_layernorm.weight"], eps)
        h = silu_mlp(h, W, pre + ".mlp")
    else:
        ht = rmsnorm(x[text_idx], W[pre + ".post_attention_layernorm.weight"], eps).to(BF16)
        hv = rmsnorm(x[vae_idx], W[pre + ".post_attention_layernorm_moe_gen.weight"], eps).to(BF16)
        h = torch.zeros_like(x).to(BF16)
        h[text_idx] = silu_mlp(ht, W, pre + ".mlp")
        h[vae_idx] = silu_mlp(hv, W, pre + ".mlp_moe_gen")
    return res + h


# ----------------------------------------------------------------------------------------------
# TaylorSeer step skipping (modeling/cache_utils/taylorseer.py:11-153; hooks qwen2_navit.py:773-829,1034-1087)
# ----------------------------------------------------------------------------------------------
class TaylorState:
    """``cache_init`` (taylorseer.py:120-153): the (cache_dic, current) pair of ONE forward stream (cond / cfg-text /
    cfg-img each own one, bagel.py:680-684).  Constants are the reference's: fresh_threshold 3, max_order 6,
    first_enhance 5, taylor_cache True, fresh_ratio 0."""

    def __init__(self, num_steps):
        self.fresh_threshold, self.max_order, self.first_enhance = 3, 6, 5
        self.cache_counter = 0
        self.cal_threshold = None
        self.activated_steps = [0]
        self.step = 0
        self.num_steps = num_steps
        self.type = None
        self.factors = {}          # layer -> {order: bf16 tensor}


def taylor_cal_type(st):
    """``cal_type`` + ``force_scheduler`` (taylorseer.py:64-117) for taylor_cache=True, fresh_ratio=0."""
    first_step = st.step < st.first_enhance
    fresh_interval = st.fresh_threshold if first_step else st.cal_threshold
    if first_step or st.cache_counter == fresh_interval - 1:
        st.type = "full"
        st.cache_counter = 0
        st.activated_steps.append(st.step)
        st.cal_threshold = int(round(st.fresh_threshold / 1.0))     # linear_step_weight = 0 -> step_factor = 1
    else:
        st.cache_counter += 1
        st.type = "Taylor"
    return st.type


def taylor_derivative_approximation(st, layer, feature):
    """``derivative_approximation`` (taylorseer.py:11-30): finite differences in the feature dtype (bf16)."""
    dist = st.activated_steps[-1] - st.activated_steps[-2]
    old = st.factors.get(layer, {})
    new = {0: feature}
    for i in range(st.max_order):
        if old.get(i, None) is not None and st.step > st.first_enhance - 2:
            new[i + 1] = (new[i] - old[i]) / dist
        else:
            break
    st.factors[layer] = new


def taylor_formula(st, layer):
    """``taylor_formula`` (taylorseer.py:32-46): sum_i f_i x^i / i!, every product and sum rounded to bf16."""
    x = st.step - st.activated_steps[-1]
    out = 0
    f = st.factors[layer]
    for i in range(len(f)):
        out += (1 / math.factorial(i)) * f[i] * (x ** i)
    return out


@_explicit_casts
def llm_forward(W, cfg, x, query_lens, position_ids, q_idx, cache, kv_lens, kv_idx, update, causal,
                mode="und", vae_idx=None, text_idx=None, num_layers=None, taylor=None, taylor_last_layer_only=False):
    """Qwen2Model.forward_inference, qwen2_navit.py:1018-1092.  ``taylor``: a TaylorState -> the TaylorSeer hooks of
    :1034-1037,1057-1061,1086-1087 and of the layer (:773-829) are active.  On a 'Taylor' step every layer REPLACES the
    sequence by its own extrapolation, so only the last layer's cache reaches the output; ``taylor_last_layer_only``
    evaluates just that one (what the MI355X engine does) -- bit-identical, asserted in tests/test_oracle_golden.py."""
    hd = cfg["hidden_size"] // cfg["num_attention_heads"]
    cos_sin = rope_tables(position_ids, hd, cfg["rope_theta"], x.dtype)
    L = cfg["num_hidden_layers"] if num_layers is None else num_layers
    typ = taylor_cal_type(taylor) if taylor is not None else "full"
    for i in range(L):
        if typ == "full":
            if taylor is not None and taylor.step == 0:
                taylor.factors[i] = {}                      # taylor_cache_init, taylorseer.py:48-56
            x = mot_layer(W, cfg, i, x, query_lens, cos_sin, q_idx, cache, kv_lens, kv_idx, update, causal,
                          mode, vae_idx, text_idx)
            if taylor is not None and (not taylor_last_layer_only or i == L - 1):
                taylor_derivative_approximation(taylor, i, x)
        elif not taylor_last_layer_only or i == L - 1:
            x = taylor_formula(taylor, i)
    if taylor is not None:
        taylor.step += 1
    eps = cfg["rms_norm_eps"]
    if mode == "und" or "Mo" not in cfg.get("layer_module", "Qwen2MoTDecoderLayer"):     # use_moe, qwen2_navit.py:1074-1084
        x = rmsnorm(x, W["language_model.model.norm.weight"], eps)
    else:
        y = torch.zeros_like(x)
        y[text_idx] = rmsnorm(x[text_idx], W["language_model.model.norm.weight"], eps)
        y[vae_idx] = rmsnorm(x[vae_idx], W["language_model.model.norm_moe_gen.weight"], eps)
        x = y
    return x


def embed_tokens(W, ids):
    return F.embedding(ids, W["language_model.model.embed_tokens.weight"])


# ----------------------------------------------------------------------------------------------
# prefill entry points (bagel.py:267-297, 362-415, 491-550)
# ----------------------------------------------------------------------------------------------
@_explicit_casts
def forward_cache_update_text(W, cfg, cache, packed_text_ids, packed_text_position_ids, text_token_lens,
                              packed_text_indexes, packed_key_value_indexes, key_values_lens):
    x = embed_tokens(W, packed_text_ids)
    llm_forward(W, cfg["llm"], x, text_token_lens, packed_text_position_ids, packed_text_indexes, cache,
                key_values_lens, packed_key_value_indexes, True, True, "und")
    return cache


@_explicit_casts
def connector(W, x):
    """MLPconnector, modeling_utils.py:113-124 (gelu_pytorch_tanh)."""
    h = linear(x, W["connector.fc1.weight"], W["connector.fc1.bias"])
    h = F.gelu(h, approximate="tanh")
    return linear(h, W["connector.fc2.weight"], W["connector.fc2.bias"])


@_explicit_casts
def forward_cache_update_vit(W, cfg, cache, packed_text_ids, packed_text_indexes, packed_vit_tokens,
                             packed_vit_token_indexes, packed_vit_position_ids, vit_token_seqlens,
                             packed_position_ids, packed_seqlens, packed_indexes, packed_key_value_indexes,
                             key_values_lens):
    H = cfg["llm"]["hidden_size"]
    te = embed_tokens(W, packed_text_ids)
    seq = te.new_zeros((int(sum(packed_seqlens)), H))
    seq[packed_text_indexes] = te
    cu = F.pad(torch.cumsum(vit_token_seqlens, 0), (1, 0)).to(torch.int32)
    vt = siglip_forward(W, cfg["vit"], packed_vit_tokens, packed_vit_position_ids, cu,
                        int(vit_token_seqlens.max()))
    vt = connector(W, vt)
    vt = vt + W["vit_pos_embed.pos_embed"][packed_vit_position_ids]
    if vt.dtype != seq.dtype:
        vt = vt.to(seq.dtype)
    seq[packed_vit_token_indexes] = vt
    llm_forward(W, cfg["llm"], seq, packed_seqlens, packed_position_ids, packed_indexes, cache,
                key_values_lens, packed_key_value_indexes, True, False, "und")
    return cache


def patchify_latent(latent, h, w, p, C):
    """bagel.py:516-519."""
    lat = latent[:, : h * p, : w * p].reshape(C, h, p, w, p)
    return torch.einsum("chpwq->hwpqc", lat).reshape(-1, p * p * C)


@_explicit_casts
def forward_cache_update_vae(W, cfg, vae_W, cache, padded_images, patchified_vae_latent_shapes,
                             packed_vae_position_ids, packed_timesteps, packed_vae_token_indexes,
                             packed_text_ids, packed_text_indexes, packed_position_ids, packed_seqlens,
                             packed_indexes, key_values_lens, packed_key_value_indexes, sample_noise=None):
    H = cfg["llm"]["hidden_size"]
    p, C = cfg["bagel"]["latent_patch_size"], cfg["vae"]["z_channels"]
    te = embed_tokens(W, packed_text_ids)
    seq = te.new_zeros((int(sum(packed_seqlens)), H))
    seq[packed_text_indexes] = te
    lat = vae_encode(vae_W, cfg["vae"], padded_images, sample_noise)
    packed = torch.cat([patchify_latent(l, h, w, p, C) for l, (h, w) in zip(lat, patchified_vae_latent_shapes)], 0)
    pos = W["latent_pos_embed.pos_embed"][packed_vae_position_ids]
    temb = timestep_embed(packed_timesteps, W)
    packed = linear(packed, W["vae2llm.weight"], W["vae2llm.bias"]) + temb + pos
    if packed.dtype != seq.dtype:
        packed = packed.to(seq.dtype)
    seq[packed_vae_token_indexes] = packed
    llm_forward(W, cfg["llm"], seq, packed_seqlens, packed_position_ids, packed_indexes, cache,
                key_values_lens, packed_key_value_indexes, True, False, "gen",
                packed_vae_token_indexes, packed_text_indexes)
    return cache


# ----------------------------------------------------------------------------------------------
# rectified-flow sampler (bagel.py:644-907)
# ----------------------------------------------------------------------------------------------
def flow_schedule(num_timesteps, shift):
    """bagel.py:693-696: T points, T-1 Euler steps."""
    t = torch.linspace(1, 0, num_timesteps)
    t = shift * t / (1 + (shift - 1) * t)
    return t[:-1], t[:-1] - t[1:]


@_explicit_casts
def cfg_combine(v_t, v_ct, v_ci, cfg_text_scale, cfg_img_scale=1.0, cfg_renorm_min=0.0, cfg_renorm_type="global"):
    """The classifier-free-guidance combine + renorm of bagel.py:854-905 on the three velocities (v_ci None when cfg_img_scale <= 1): every operation is
    an eager op on the inputs' dtype, as in the reference."""
    if cfg_renorm_type == "text_channel":
        v_text_ = v_ct + cfg_text_scale * (v_t - v_ct)
        n0 = torch.norm(v_t, dim=-1, keepdim=True)
        n1 = torch.norm(v_text_, dim=-1, keepdim=True)
        scale = (n0 / (n1 + 1e-8)).clamp(min=cfg_renorm_min, max=1.0)
        v_text = v_text_ * scale
        return v_ci + cfg_img_scale * (v_text - v_ci) if cfg_img_scale > 1.0 else v_text
    v_text_ = v_ct + cfg_text_scale * (v_t - v_ct)
    v_ = v_ci + cfg_img_scale * (v_text_ - v_ci) if cfg_img_scale > 1.0 else v_text_
    if cfg_renorm_type == "global":
        n0, n1 = torch.norm(v_t), torch.norm(v_)
    elif cfg_renorm_type == "channel":
        n0, n1 = torch.norm(v_t, dim=-1, keepdim=True), torch.norm(v_, dim=-1, keepdim=True)
    else:
        raise NotImplementedError(cfg_renorm_type)
    scale = (n0 / (n1 + 1e-8)).clamp(min=cfg_renorm_min, max=1.0)
    return v_ * scale


def forward_flow(W, cfg, x_t, timestep, gi, cache, cfg_text=None, cfg_img=None, cfg_text_scale=1.0,
                 cfg_img_scale=1.0, cfg_renorm_min=0.0, cfg_renorm_type="global", taylor=None,
                 taylor_last_layer_only=False, parts=None):
    """Bagel._forward_flow, bagel.py:757-907.  ``cfg_text``/``cfg_img`` = dict(cache, position_ids,
    query_indexes, key_values_lens, key_value_indexes) or None.  ``taylor`` = (cond, cfg_text, cfg_img) TaylorStates
    (bagel.py:816-818,836-838,855-857) or None."""
    H = cfg["llm"]["hidden_size"]
    te = embed_tokens(W, gi["packed_text_ids"])
    seq = te.new_zeros((int(sum(gi["packed_seqlens"])), H))
    seq[gi["packed_text_indexes"]] = te
    assert timestep.unique().shape[0] == 1
    pos = W["latent_pos_embed.pos_embed"][gi["packed_vae_position_ids"]]
    temb = timestep_embed(timestep, W)
    h = linear(x_t, W["vae2llm.weight"], W["vae2llm.bias"]) + temb + pos
    if h.dtype != seq.dtype:
        h = h.to(seq.dtype)
    seq[gi["packed_vae_token_indexes"]] = h
    vae_idx, text_idx = gi["packed_vae_token_indexes"], gi["packed_text_indexes"]

    def run(cache_, pos_ids, q_idx, kv_lens, kv_idx, ts=None):
        mode = "gen" if "Mo" in cfg["llm"].get("layer_module", "Qwen2MoTDecoderLayer") else "und"     # use_moe, bagel.py:808-815
        out = llm_forward(W, cfg["llm"], seq, gi["packed_seqlens"], pos_ids, q_idx, cache_, kv_lens, kv_idx,
                          False, False, mode, vae_idx, text_idx, taylor=ts, taylor_last_layer_only=taylor_last_layer_only)
        v = linear(out, W["llm2vae.weight"], W["llm2vae.bias"])
        return v[vae_idx]

    ty = taylor if taylor is not None else (None, None, None)
    v_t = run(cache, gi["packed_position_ids"], gi["packed_indexes"], gi["key_values_lens"],
              gi["packed_key_value_indexes"], ty[0])
    if cfg_text_scale > 1.0:
        c = cfg_text
        v_ct = run(c["cache"], c["position_ids"], c["query_indexes"], c["key_values_lens"], c["key_value_indexes"], ty[1])
    if cfg_img_scale > 1.0:
        c = cfg_img
        v_ci = run(c["cache"], c["position_ids"], c["query_indexes"], c["key_values_lens"], c["key_value_indexes"], ty[2])

    if parts is not None:          # checker hook (bench.py full-depth parity): the single-forward velocities before the CFG combine
        parts["v_cond"] = v_t
        if cfg_text_scale > 1.0:
            parts["v_cfg_text"] = v_ct
        if cfg_img_scale > 1.0:
            parts["v_cfg_img"] = v_ci
    if cfg_text_scale > 1.0:
        v_t = cfg_combine(v_t, v_ct, v_ci if cfg_img_scale > 1.0 else None, cfg_text_scale, cfg_img_scale, cfg_renorm_min, cfg_renorm_type)
    return v_t


@_explicit_casts
def generate_image(W, cfg, gi, cache, cfg_text=None, cfg_img=None, num_timesteps=24, timestep_shift=1.0,
                   cfg_renorm_min=0.0, cfg_renorm_type="global", cfg_interval=(0, 1), cfg_text_scale=1.0,
                   cfg_img_scale=1.0, max_steps=None, enable_taylorseer=False, taylor_last_layer_only=False):
    """Bagel.generate_image, bagel.py:644-754 (``enable_taylorseer``: :680-689)."""
    taylor = tuple(TaylorState(num_timesteps) for _ in range(3)) if enable_taylorseer else None
    x_t = gi["packed_init_noises"]
    ts, dts = flow_schedule(num_timesteps, timestep_shift)
    for i, t in enumerate(ts):
        if max_steps is not None and i >= max_steps:
            break
        timestep = torch.tensor([t] * x_t.shape[0])
        if FP8_DELAYED is not None:
            FP8_DELAYED.begin_step()
        if t > cfg_interval[0] and t <= cfg_interval[1]:
            s_t, s_i = cfg_text_scale, cfg_img_scale
        else:
            s_t, s_i = 1.0, 1.0
        v_t = forward_flow(W, cfg, x_t, timestep, gi, cache, cfg_text, cfg_img, s_t, s_i, cfg_renorm_min,
                           cfg_renorm_type, taylor, taylor_last_layer_only)
        x_t = x_t - v_t.to(x_t.device) * dts[i]
    return x_t.split((gi["packed_seqlens"] - 2).tolist())


# ----------------------------------------------------------------------------------------------
# training forward (Bagel.forward bagel.py:101-229; forward_train chain qwen2_navit.py:406-497,713-755,970-1016)
# ----------------------------------------------------------------------------------------------
def attention_mask_per_sample(split_lens, attn_modes):
    """prepare_attention_mask_per_sample (data/data_utils.py:72-103): additive fp32 mask of ONE sample, 0 = attend,
    -inf = ignore.  causal split: lower triangle + everything before it; full/noise split: itself + everything before
    it; a noise split is then hidden from every other split."""
    n = sum(split_lens)
    allow = torch.zeros((n, n), dtype=torch.bool)
    c = 0
    for s, mode in zip(split_lens, attn_modes):
        assert mode in ("causal", "full", "noise")
        allow[c:c + s, c:c + s] = torch.ones((s, s)).tril().bool() if mode == "causal" else True
        allow[c:c + s, :c] = True
        c += s
    c = 0
    for s, mode in zip(split_lens, attn_modes):
        if mode == "noise":
            allow[:, c:c + s] = False
            allow[c:c + s, c:c + s] = True
        c += s
    return torch.zeros((n, n), dtype=torch.float).masked_fill_(~allow, float("-inf"))


@_explicit_casts
def mot_attention_train(W, pre, cfg, x, sample_lens, masks, cos_sin, und_idx, gen_idx):
    """PackedAttentionMoT.forward_train with nested masks, qwen2_navit.py:406-497 (bf16 cast points for BOTH experts --
    unlike forward_inference's gen mode there is no fp32 QK-norm here)."""
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = cfg["hidden_size"] // nh
    eps = cfg["rms_norm_eps"]
    a = pre + ".self_attn."
    n = x.shape[0]
    q, k, v = x.new_zeros((n, nh * hd)), x.new_zeros((n, nkv * hd)), x.new_zeros((n, nkv * hd))
    xu, xg = x[und_idx], x[gen_idx]
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        t[und_idx] = linear(xu, W[a + f"{name}_proj.weight"], W[a + f"{name}_proj.bias"])
        t[gen_idx] = linear(xg, W[a + f"{name}_proj_moe_gen.weight"], W[a + f"{name}_proj_moe_gen.bias"])
    q, k, v = q.view(-1, nh, hd), k.view(-1, nkv, hd), v.view(-1, nkv, hd)
    q_, k_ = q.new_zeros(q.shape), k.new_zeros(k.shape)
    q_[und_idx] = rmsnorm(q[und_idx], W[a + "q_norm.weight"], eps)
    q_[gen_idx] = rmsnorm(q[gen_idx], W[a + "q_norm_moe_gen.weight"], eps)
    k_[und_idx] = rmsnorm(k[und_idx], W[a + "k_norm.weight"], eps)
    k_[gen_idx] = rmsnorm(k[gen_idx], W[a + "k_norm_moe_gen.weight"], eps)
    cos, sin = cos_sin
    q_, k_ = apply_rope(q_, k_, cos, sin)
    g = nh // nkv
    k_ = k_[:, :, None, :].repeat(1, 1, g, 1).reshape(-1, nh, hd)
    vv = v[:, :, None, :].repeat(1, 1, g, 1).reshape(-1, nh, hd)
    outs = []
    for qs, ks, vs, m in zip(q_.transpose(0, 1).split(sample_lens, dim=1), k_.transpose(0, 1).split(sample_lens, dim=1),
                             vv.transpose(0, 1).split(sample_lens, dim=1), masks):
        o = F.scaled_dot_product_attention(qs.to(BF16).unsqueeze(0), ks.to(BF16).unsqueeze(0), vs.to(BF16).unsqueeze(0),
                                           m.to(BF16).unsqueeze(0))
        outs.append(o.squeeze(0))
    o = torch.cat(outs, dim=1).transpose(0, 1).reshape(-1, nh * hd)
    o_ = o.new_zeros(o.shape)
    o_[und_idx] = linear(o[und_idx], W[a + "o_proj.weight"])
    o_[gen_idx] = linear(o[gen_idx], W[a + "o_proj_moe_gen.weight"])
    return o_


@_explicit_casts
def shared_attention_train(W, pre, cfg, x, sample_lens, masks, cos_sin):
    """PackedAttention.forward_train, qwen2_navit.py:252-320: ONE set of projections / QK-norms over the whole packed sequence (the attention of
    Qwen2DecoderLayer and of Qwen2MoEDecoderLayer); everything after the projections is mot_attention_train's."""
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = cfg["hidden_size"] // nh
    eps = cfg["rms_norm_eps"]
    a = pre + ".self_attn."
    q = linear(x, W[a + "q_proj.weight"], W[a + "q_proj.bias"]).view(-1, nh, hd)
    k = linear(x, W[a + "k_proj.weight"], W[a + "k_proj.bias"]).view(-1, nkv, hd)
    v = linear(x, W[a + "v_proj.weight"], W[a + "v_proj.bias"]).view(-1, nkv, hd)
    q_, k_ = rmsnorm(q, W[a + "q_norm.weight"], eps), rmsnorm(k, W[a + "k_norm.weight"], eps)
    cos, sin = cos_sin
    q_, k_ = apply_rope(q_, k_, cos, sin)
    g = nh // nkv
    k_ = k_[:, :, None, :].repeat(1, 1, g, 1).reshape(-1, nh, hd)
    vv = v[:, :, None, :].repeat(1, 1, g, 1).reshape(-1, nh, hd)
    outs = []
    for qs, ks, vs, m in zip(q_.transpose(0, 1).split(sample_lens, dim=1), k_.transpose(0, 1).split(sample_lens, dim=1),
                             vv.transpose(0, 1).split(sample_lens, dim=1), masks):
        o = F.scaled_dot_product_attention(qs.to(BF16).unsqueeze(0), ks.to(BF16).unsqueeze(0), vs.to(BF16).unsqueeze(0),
                                           m.to(BF16).unsqueeze(0))
        outs.append(o.squeeze(0))
    o = torch.cat(outs, dim=1).transpose(0, 1).reshape(-1, nh * hd)
    return linear(o, W[a + "o_proj.weight"])


@_explicit_casts
def llm_forward_train(W, cfg, x, sample_lens, masks, position_ids, und_idx, gen_idx):
    """Qwen2Model.forward_train + Qwen2MoTDecoderLayer.forward_train, qwen2_navit.py:970-1016,713-755; with ``cfg['layer_module']`` also
    Qwen2DecoderLayer.forward_train (:620-646: no routing at all) and Qwen2MoEDecoderLayer.forward_train (:852-883: shared attention and
    layer norms, per-modality MLP; the model's final norm is per modality for both "Mo" kinds, :1003-1012)."""
    hd = cfg["hidden_size"] // cfg["num_attention_heads"]
    eps = cfg["rms_norm_eps"]
    cos_sin = rope_tables(position_ids, hd, cfg["rope_theta"], x.dtype)
    kind = cfg.get("layer_module", "Qwen2MoTDecoderLayer")
    if kind != "Qwen2MoTDecoderLayer":
        for i in range(cfg["num_hidden_layers"]):
            pre = f"language_model.model.layers.{i}"
            res = x
            x = res + shared_attention_train(W, pre, cfg, rmsnorm(x, W[pre + ".input_layernorm.weight"], eps), sample_lens, masks, cos_sin)
            res = x
            h = rmsnorm(x, W[pre + ".post_attention_layernorm.weight"], eps)
            if kind == "Qwen2MoEDecoderLayer":
                hn = h.new_zeros(h.shape)
                hn[und_idx] = silu_mlp(h[und_idx], W, pre + ".mlp")
                hn[gen_idx] = silu_mlp(h[gen_idx], W, pre + ".mlp_moe_gen")
                x = res + hn
            else:
                x = res + silu_mlp(h, W, pre + ".mlp")
        if kind == "Qwen2MoEDecoderLayer":
            y = torch.zeros_like(x)
            y[und_idx] = rmsnorm(x[und_idx], W["language_model.model.norm.weight"], eps)
            y[gen_idx] = rmsnorm(x[gen_idx], W["language_model.model.norm_moe_gen.weight"], eps)
            return y
        return rmsnorm(x, W["language_model.model.norm.weight"], eps)
    for i in range(cfg["num_hidden_layers"]):
        pre = f"language_model.model.layers.{i}"
        res = x
        h = x.new_zeros(x.shape)
        h[und_idx] = rmsnorm(x[und_idx], W[pre + ".input_layernorm.weight"], eps)
        h[gen_idx] = rmsnorm(x[gen_idx], W[pre + ".input_layernorm_moe_gen.weight"], eps)
        x = res + mot_attention_train(W, pre, cfg, h, sample_lens, masks, cos_sin, und_idx, gen_idx)
        res = x
        h = x.new_zeros(x.shape)
        h[und_idx] = silu_mlp(rmsnorm(x[und_idx], W[pre + ".post_attention_layernorm.weight"], eps), W, pre + ".mlp")
        h[gen_idx] = silu_mlp(rmsnorm(x[gen_idx], W[pre + ".post_attention_layernorm_moe_gen.weight"], eps), W, pre + ".mlp_moe_gen")
        x = res + h
    y = torch.zeros_like(x)
    y[und_idx] = rmsnorm(x[und_idx], W["language_model.model.norm.weight"], eps)
    y[gen_idx] = rmsnorm(x[gen_idx], W["language_model.model.norm_moe_gen.weight"], eps)
    return y


@_explicit_casts
def bagel_forward_train(W, cfg, batch, noise, timestep_shift=1.0):
    """Bagel.forward, bagel.py:101-229 (MoT, nested masks).  ``noise`` is the ``torch.randn_like(packed_latent_clean)``
    draw of :184, passed in so that both sides use the same numbers.  Returns dict(mse fp32 [n_mse, 64], ce fp32 [n_ce])."""
    H = cfg["llm"]["hidden_size"]
    b = batch
    te = embed_tokens(W, b["packed_text_ids"])
    seq = te.new_zeros((b["sequence_length"], H))
    seq[b["packed_text_indexes"]] = te
    und_idx = b["packed_text_indexes"]
    if b.get("packed_vit_tokens") is not None:
        cu = F.pad(torch.cumsum(b["vit_token_seqlens"], dim=0), (1, 0)).to(torch.int32)
        feats = siglip_forward(W, cfg["vit"], b["packed_vit_tokens"], b["packed_vit_position_ids"], cu,
                               int(b["vit_token_seqlens"].max()))
        emb = connector(W, feats) + W["vit_pos_embed.pos_embed"][b["packed_vit_position_ids"]]
        seq[b["packed_vit_token_indexes"]] = emb
        und_idx = torch.cat([b["packed_text_indexes"], b["packed_vit_token_indexes"]], dim=0)
    p, C = cfg["bagel"]["latent_patch_size"], cfg["vae"]["z_channels"]
    clean = torch.cat([patchify_latent(lat, h, w, p, C) for lat, (h, w) in zip(b["padded_latent"], b["patchified_vae_latent_shapes"])], 0)
    t = torch.sigmoid(b["packed_timesteps"])
    t = timestep_shift * t / (1 + (timestep_shift - 1) * t)
    x_t = (1 - t[:, None]) * clean + t[:, None] * noise
    lat = linear(x_t, W["vae2llm.weight"], W["vae2llm.bias"]) + timestep_embed(t, W) + W["latent_pos_embed.pos_embed"][b["packed_latent_position_ids"]]
    seq[b["packed_vae_token_indexes"]] = lat
    out = llm_forward_train(W, cfg["llm"], seq, b["sample_lens"], b["nested_attention_masks"], b["packed_position_ids"],
                            und_idx, b["packed_vae_token_indexes"])
    preds = linear(out[b["mse_loss_indexes"]], W["llm2vae.weight"], W["llm2vae.bias"])
    target = noise - clean
    mse = (preds - target[t > 0]) ** 2
    ce = None
    if b.get("ce_loss_indexes") is not None:
        logits = linear(out[b["ce_loss_indexes"]], W["language_model.lm_head.weight"])
        ce = F.cross_entropy(logits.float(), b["packed_label_ids"], reduction="none")
    return dict(mse=mse, ce=ce)


def training_step_loss(out, ce_loss_weights=None, ce_weight=1.0, mse_weight=1.0):
    """The scalar loss of one rank's micro-step, train/pretrain_unified_navit.py:705-727 (world size 1): the CE mean (weighted by
    ``ce_loss_weights`` under ``ce_loss_reweighting``) times ce_weight plus the per-token mean-over-channels MSE, averaged over the
    MSE tokens, times mse_weight."""
    loss = 0
    if out.get("ce") is not None:
        ce = out["ce"]
        ce = (ce * ce_loss_weights).sum() / ce_loss_weights.sum() if ce_loss_weights is not None else ce.sum() / ce.shape[0]
        loss = loss + ce * ce_weight
    if out.get("mse") is not None:
        loss = loss + out["mse"].mean(dim=-1).sum() / out["mse"].shape[0] * mse_weight
    return loss


def training_step_grads(W, cfg, batch, noise, ce_loss_weights=None, names=None, ce_weight=1.0, mse_weight=1.0):
    """``loss.backward()`` of the reference's training step (pretrain_unified_navit.py:683-735) through torch's autograd over the
    restated forward: -> (loss float, {state-dict key: gradient}, losses).  ``names`` = the keys to differentiate (default: every
    floating-point entry of ``W``).  Bit-exact against the unmodified reference's own backward on the host that runs both
    (oracle/make_golden_train_grads.py, tests/test_reference_crosscheck.py)."""
    global GRAD_ENABLED
    if names is None:
        names = {k for k, v in W.items() if v.is_floating_point() and "pos_embed" not in k and "inv_freq" not in k}
    Wg = {k: (v.detach().clone().requires_grad_(True) if k in names else v) for k, v in W.items()}
    GRAD_ENABLED = True
    try:
        out = bagel_forward_train(Wg, cfg, batch, noise, timestep_shift=cfg["bagel"]["timestep_shift"])
        loss = training_step_loss(out, ce_loss_weights, ce_weight, mse_weight)
        loss.backward()
    finally:
        GRAD_ENABLED = False
    return float(loss.detach()), {k: Wg[k].grad for k in names if Wg[k].grad is not None}, {k: (None if v is None else v.detach()) for k, v in out.items()}


# ----------------------------------------------------------------------------------------------
# autoregressive text decode (bagel.py:930-1000)
# ----------------------------------------------------------------------------------------------
@_explicit_casts
def generate_text(W, cfg, cache, packed_key_value_indexes, key_values_lens, packed_start_tokens,
                  packed_query_position_ids, max_length, do_sample=False, temperature=1.0, end_token_id=None,
                  return_logits=False):
    step, seq, logits_all = 0, [], []
    curr = packed_start_tokens
    kv_idx, kv_lens, pos = packed_key_value_indexes, key_values_lens, packed_query_position_ids
    while step < max_length:
        seq.append(curr)
        x = embed_tokens(W, curr)
        qlens = torch.ones_like(curr)
        q_idx = torch.cumsum(kv_lens, 0) + torch.arange(0, len(kv_lens), dtype=kv_lens.dtype)
        parts = list(kv_idx.split(kv_lens.tolist(), 0))
        for i in range(len(parts)):
            parts[i] = parts[i] + i
        kv_idx = torch.cat(parts, 0)
        out = llm_forward(W, cfg["llm"], x, qlens, pos, q_idx, cache, kv_lens, kv_idx, True, True, "und")
        logits = linear(out, W["language_model.lm_head.weight"])
        if return_logits:
            logits_all.append(logits)
        if do_sample:
            probs = F.softmax(logits / temperature, dim=-1)
            curr = torch.multinomial(probs, num_samples=1).squeeze(1)
        else:
            curr = torch.argmax(logits, dim=-1)
        parts = list(kv_idx.split(kv_lens.tolist(), 0))
        for i in range(len(parts)):
            parts[i] = torch.cat([parts[i], torch.tensor([parts[i][-1] + 1])], 0)
        kv_idx = torch.cat(parts, 0)
        kv_lens = kv_lens + 1
        pos = pos + 1
        step += 1
        if end_token_id is not None and curr[0] == end_token_id:
            break
    toks = torch.stack(seq, 0)
    return (toks, torch.stack(logits_all, 0)) if return_logits else toks


# ----------------------------------------------------------------------------------------------
# SigLIP NaViT encoder (siglip_navit.py:145-402), rope=False path + optional 2-D rope
# ----------------------------------------------------------------------------------------------
def rope2d_tables(dim, max_h, max_w, base=10000):
    """RotaryEmbedding2D, siglip_navit.py:102-133."""
    freq = torch.arange(0, dim, 2, dtype=torch.int64).float() / dim
    inv = 1.0 / (base ** freq)
    gh = torch.arange(0, max_h).to(inv.dtype)[:, None].repeat(1, max_w)
    gw = torch.arange(0, max_w).to(inv.dtype)[None, :].repeat(max_h, 1)

    def side(g):
        fr = g[..., None] * inv[None, None, :]
        e = torch.cat((fr, fr), -1).flatten(0, 1)
        return e.cos(), e.sin()

    return side(gh) + side(gw)   # cos_h, sin_h, cos_w, sin_w


@_explicit_casts
def siglip_forward(W, vcfg, pixels, pos_ids, cu_seqlens, max_seqlen):
    pre = "vit_model.vision_model."
    nh = vcfg["num_attention_heads"]
    D = vcfg["hidden_size"]
    hd = D // nh
    eps = vcfg.get("layer_norm_eps", 1e-6)
    x = linear(pixels, W[pre + "embeddings.patch_embedding.weight"], W[pre + "embeddings.patch_embedding.bias"])
    use_rope = vcfg.get("rope", False)
    if not use_rope:
        x = x + F.embedding(pos_ids, W[pre + "embeddings.position_embedding.weight"])
    else:
        # the RotaryEmbedding2D buffers are persistent state-dict entries: a bf16 model carries bf16-rounded tables
        ch, sh, cw, sw = [W[pre + "rope." + n][pos_ids] for n in ("cos_h", "sin_h", "cos_w", "sin_w")]
    for i in range(vcfg["num_hidden_layers"]):
        lp = f"{pre}encoder.layers.{i}."
        res = x
        h = F.layer_norm(x, (D,), W[lp + "layer_norm1.weight"], W[lp + "layer_norm1.bias"], eps)
        q = linear(h, W[lp + "self_attn.q_proj.weight"], W[lp + "self_attn.q_proj.bias"]).view(-1, nh, hd)
        k = linear(h, W[lp + "self_attn.k_proj.weight"], W[lp + "self_attn.k_proj.bias"]).view(-1, nh, hd)
        v = linear(h, W[lp + "self_attn.v_proj.weight"], W[lp + "self_attn.v_proj.bias"]).view(-1, nh, hd)
        if use_rope:
            def rot(a, b, c, s):
                c, s = c.unsqueeze(1), s.unsqueeze(1)
                return (a * c) + (rotate_half(a) * s), (b * c) + (rotate_half(b) * s)
            qh, kh = rot(q[:, :, : hd // 2], k[:, :, : hd // 2], ch, sh)
            qw, kw = rot(q[:, :, hd // 2:], k[:, :, hd // 2:], cw, sw)
            q, k = torch.cat([qh, qw], -1), torch.cat([kh, kw], -1)
        o = attn_varlen(q.to(BF16), k.to(BF16), v.to(BF16), cu_seqlens, cu_seqlens, max_seqlen, max_seqlen,
                        causal=False)
        o = linear(o.reshape(-1, D), W[lp + "self_attn.out_proj.weight"], W[lp + "self_attn.out_proj.bias"])
        x = res + o
        res = x
        h = F.layer_norm(x, (D,), W[lp + "layer_norm2.weight"], W[lp + "layer_norm2.bias"], eps)
        h = linear(h, W[lp + "mlp.fc1.weight"], W[lp + "mlp.fc1.bias"])
        h = F.gelu(h, approximate="tanh")
        h = linear(h, W[lp + "mlp.fc2.weight"], W[lp + "mlp.fc2.bias"])
        x = res + h
    return F.layer_norm(x, (D,), W[pre + "post_layernorm.weight"], W[pre + "post_layernorm.bias"], eps)


# ----------------------------------------------------------------------------------------------
# FLUX-style VAE (modeling/autoencoder.py), fp32 -- or under the inferencer's bf16 autocast (inferencer.py:233 -> :174-185)
# ----------------------------------------------------------------------------------------------
# VAE_AUTOCAST selects the cast points of the VAE functions below:
#   None     the VAE in fp32, outside any autocast region (app.py:48,138; eval/gen/gen_images_mp.py:93);
#   "cpu"    what ``torch.autocast("cpu", bf16)`` does to the unmodified reference on THIS container's CPU: conv2d and SDPA run in bf16
#            (autocast's lower-precision list), GroupNorm receives the bf16 tensor and -- being in NO CPU autocast list -- answers in bf16,
#            swish and the residual adds follow their inputs: pinned bit for bit against the reference by oracle/make_golden_vae_bf16.py;
#   "cuda"   what ``torch.autocast("cuda", bf16)`` does, i.e. what InterleaveInferencer runs the reference's VAE under: identical, except
#            that the CUDA autocast policy lists group_norm among the fp32 ops -- the bf16 input is cast up, statistics AND result are
#            fp32, swish runs on that fp32 tensor, and the next conv casts to bf16.  This is the semantics the MI355X bf16 VAE implements
#            (csrc/vae.hip bagel_conv_gemm_bf16 / bagel_groupnorm_bf16); it differs from "cpu" only by where the GroupNorm result is
#            rounded.
VAE_AUTOCAST = None


def _gn(x, W, p):
    if VAE_AUTOCAST == "cuda":
        return F.group_norm(x.float(), 32, W[p + ".weight"], W[p + ".bias"], 1e-6)
    return F.group_norm(x, 32, W[p + ".weight"], W[p + ".bias"], 1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(x, W, p, stride=1, padding=1):
    w, b = W[p + ".weight"], W[p + ".bias"]
    if VAE_AUTOCAST:
        x, w, b = x.to(torch.bfloat16), w.to(torch.bfloat16), b.to(torch.bfloat16)
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def _resblock(x, W, p):
    """ResnetBlock.forward, autoencoder.py:83-95."""
    h = _conv(_swish(_gn(x, W, p + ".norm1")), W, p + ".conv1")
    h = _conv(_swish(_gn(h, W, p + ".norm2")), W, p + ".conv2")
    if (p + ".nin_shortcut.weight") in W:
        x = _conv(x, W, p + ".nin_shortcut", padding=0)
    return x + h


def _attnblock(x, W, p):
    """AttnBlock.forward, autoencoder.py:49-65 (single head, head_dim = C)."""
    h = _gn(x, W, p + ".norm")
    q, k, v = (_conv(h, W, p + "." + n, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    from einops import rearrange   # same view/stride pattern as the reference => same conv kernel choice
    q, k, v = (rearrange(t, "b c h w -> b 1 (h w) c").contiguous() for t in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v)
    o = rearrange(o, "b 1 (h w) c -> b c h w", h=hh, w=ww, c=c, b=b)
    return x + _conv(o, W, p + ".proj_out", padding=0)


@_explicit_casts
def vae_encode(W, vcfg, x, sample_noise=None):
    """AutoEncoder.encode, autoencoder.py:315-318; Encoder.forward :172-193; DiagonalGaussian :280-287.
    ``sample_noise``: the randn_like draw (None -> draw from the global CPU generator like the reference)."""
    nres, nb = len(vcfg["ch_mult"]), vcfg["num_res_blocks"]
    h = _conv(x, W, "encoder.conv_in")
    for lvl in range(nres):
        for b in range(nb):
            h = _resblock(h, W, f"encoder.down.{lvl}.block.{b}")
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = _conv(h, W, f"encoder.down.{lvl}.downsample.conv", stride=2, padding=0)
    h = _resblock(h, W, "encoder.mid.block_1")
    h = _attnblock(h, W, "encoder.mid.attn_1")
    h = _resblock(h, W, "encoder.mid.block_2")
    h = _conv(_swish(_gn(h, W, "encoder.norm_out")), W, "encoder.conv_out")
    mean, logvar = torch.chunk(h, 2, dim=1)
    std = torch.exp(0.5 * logvar)
    noise = torch.randn_like(mean) if sample_noise is None else sample_noise.to(mean.dtype)     # randn_like(mean): the moments' dtype
    z = mean + std * noise
    return vcfg["scale_factor"] * (z - vcfg["shift_factor"])


@_explicit_casts
def vae_decode(W, vcfg, z):
    """AutoEncoder.decode, autoencoder.py:320-322; Decoder.forward :250-272."""
    nres, nb = len(vcfg["ch_mult"]), vcfg["num_res_blocks"]
    z = z / vcfg["scale_factor"] + vcfg["shift_factor"]
    h = _conv(z, W, "decoder.conv_in")
    h = _resblock(h, W, "decoder.mid.block_1")
    h = _attnblock(h, W, "decoder.mid.attn_1")
    h = _resblock(h, W, "decoder.mid.block_2")
    for lvl in reversed(range(nres)):
        for b in range(nb + 1):
            h = _resblock(h, W, f"decoder.up.{lvl}.block.{b}")
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, W, f"decoder.up.{lvl}.upsample.conv")
    return _conv(_swish(_gn(h, W, "decoder.norm_out")), W, "decoder.conv_out")


@_explicit_casts
def latent_to_image_uint8(vae_W, vcfg, latent, H, W_, downsample, p, C):
    """InterleaveInferencer.decode_image, inferencer.py:174-185 (truncating uint8 cast)."""
    h, w = H // downsample, W_ // downsample
    lat = latent.reshape(1, h, w, p, p, C)
    lat = torch.einsum("nhwpqc->nchpwq", lat).reshape(1, C, h * p, w * p)
    img = vae_decode(vae_W, vcfg, lat)
    img = (img * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255
    return img.to(torch.uint8)
