"""MoT Qwen2 backbone on packed (NaViT) sequences -- MI355X execution plan behind the reference's API.

Public surface mirrors modeling/bagel/qwen2_navit.py of the reference (Qwen2Config :46,152-204; NaiveCache :207-221;
BaseNavitOutputWithPast :224-227; Qwen2ForCausalLM.forward_inference :1157-1188; state-dict keys of
PackedAttentionMoT :381-398, Qwen2MoTDecoderLayer :687-705, Qwen2Model :943-959) so checkpoints and callers drop in.
What is different is HOW a forward runs (MoTEngine below):

  * q/k/v projections are ONE fused GEMM per layer, gate/up ONE GEMM with the SwiGLU product in its epilogue,
    o_proj / down_proj fold the residual add into their epilogue;
  * MoT routing (text rows -> und expert, latent rows -> gen expert) is a two-group GEMM launch with row index
    lists -- the reference's ~32 gather/scatter kernels per layer (qwen2_navit.py:526-548,593-594,784-787,812-820)
    do not exist here;
  * attention reads the context KV cache and this forward's K/V as two segments (no merge copy, :563-570);
  * every host-side scalar the reference syncs for (sum(query_lens), max(...).item(), :563,585-586) is computed
    once per ForwardPlan on the host; the 28-layer loop launches kernels only.
"""
import json

import torch
from torch import nn

from ... import ops
from ..packed import PackedWeights

BF16 = torch.bfloat16
# attention through the planned persistent kernel (csrc/attention2.hip; default) or the one-tile-per-workgroup kernel (BAGEL_ATTN_PLANNED=0:
# same-box A/B runs and a cross-check -- items that are not key-split are bit-identical between the two)
import os as _os
ATTN_PLANNED = _os.environ.get("BAGEL_ATTN_PLANNED", "1") != "0"


# ------------------------------------------------------------------------------------------------------------
# config / small types
# ------------------------------------------------------------------------------------------------------------
class Qwen2Config:
    """Attribute bag with the reference's field names (qwen2_navit.py:152-204)."""
    model_type = "qwen2"

    def __init__(self, vocab_size=151936, hidden_size=4096, intermediate_size=22016, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=32, hidden_act="silu", max_position_embeddings=32768,
                 initializer_range=0.02, rms_norm_eps=1e-6, use_cache=True, tie_word_embeddings=False,
                 rope_theta=10000.0, rope_scaling=None, use_sliding_window=False, sliding_window=4096,
                 max_window_layers=28, attention_dropout=0.0, is_causal=True, _attn_implementation="flash_attention_2",
                 qk_norm=True, layer_module="Qwen2DecoderLayer", freeze_und=False, pad_token_id=None, **kwargs):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.hidden_act = hidden_act
        self.max_position_embeddings = max_position_embeddings
        self.initializer_range = initializer_range
        self.rms_norm_eps = rms_norm_eps
        self.use_cache = use_cache
        self.tie_word_embeddings = tie_word_embeddings
        self.rope_theta = rope_theta
        self.rope_scaling = rope_scaling
        self.use_sliding_window = use_sliding_window
        self.sliding_window = sliding_window
        self.max_window_layers = max_window_layers
        self.attention_dropout = attention_dropout
        self.is_causal = is_causal
        self.qk_norm = qk_norm
        self.layer_module = layer_module
        self.freeze_und = freeze_und
        self.pad_token_id = pad_token_id
        for k, v in kwargs.items():
            setattr(self, k, v)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))

    def to_dict(self):
        return dict(self.__dict__)


class BaseNavitOutputWithPast:
    """(packed_query_sequence, past_key_values) -- qwen2_navit.py:224-227."""

    def __init__(self, packed_query_sequence=None, past_key_values=None):
        self.packed_query_sequence = packed_query_sequence
        self.past_key_values = past_key_values

    def __iter__(self):
        return iter((self.packed_query_sequence, self.past_key_values))


def _ceil_to(x, m):
    return (x + m - 1) // m * m


def padded_head_dim(hd):
    """Head dims the attention kernel runs natively; smaller heads are zero-padded in the packed weights."""
    if hd <= 64:
        return 64
    if hd <= 128:
        return 128
    raise NotImplementedError(f"head_dim {hd} > 128 is not supported by the attention kernel")


# ------------------------------------------------------------------------------------------------------------
# KV cache
# ------------------------------------------------------------------------------------------------------------
class _LayerView:
    """dict-like ``cache.key_cache`` / ``cache.value_cache`` of the reference: layer -> (L, nkv, hd) tensor | None."""

    def __init__(self, cache, which):
        self._c, self._w = cache, which

    def __len__(self):
        return self._c._num_layers

    def __iter__(self):
        return iter(range(self._c._num_layers))

    def keys(self):
        return range(self._c._num_layers)

    def __getitem__(self, layer):
        store = self._c._k if self._w == "k" else self._c._v
        t = store[layer]
        if t is None:
            return None
        c = self._c
        # READ-ONLY when the layer is shared copy-on-write (after copy.deepcopy / concat): this is a live view of the shared buffer, and torch has no
        # read-only tensors.  Writers go through store() / ctx_tensors(), which un-share first; a caller that wants to edit a cached tensor in place
        # through the reference protocol calls cache.unshare() first (the reference's deepcopy gave it private storage).
        return t[: c._total].view(c._total, c._nkv, c._dp)[..., : c._hd]

    def items(self):
        return [(i, self[i]) for i in range(len(self))]

    def values(self):
        return [self[i] for i in range(len(self))]


# the name "int8" is refused everywhere on purpose: it is the reference's load mode 3 and that algorithm is not what "int8_rowwise" computes
LLM_INT8_NOT_BUILT = ("'int8' names the reference's LLM.int8 load mode (app.py:126-131: vector-wise int8 activations x int8 weights with the outlier columns, "
                      "|x| >= 6.0, kept in fp16), which is NOT built; the 8-bit option here is 'int8_rowwise' (weight-only row-wise absmax INT8, bf16 "
                      "activations: a different algorithm with different results)")


class NaiveCache:
    """KV container with the reference's protocol (qwen2_navit.py:207-221): ``NaiveCache(num_layers)``,
    ``.key_cache[layer]`` / ``.value_cache[layer]`` -> packed (sum L, nkv, hd) bf16 or None, ``.num_layers``,
    ``.seq_lens``; survives ``copy.deepcopy`` (inferencer.py:189,230-231,244,253).

    Storage is MI355X-oriented: per layer one (capacity, nkv*Dp) K buffer and V buffer that a single-sample context
    appends to IN PLACE (the reference re-allocates and re-scatters the whole cache on every cached forward), plus a
    lazily built transposed copy V^T[(g*Dp+d), col] that the MFMA attention kernel consumes."""

    def __init__(self, num_layers):
        self._num_layers = num_layers
        self._k = {i: None for i in range(num_layers)}
        self._v = {i: None for i in range(num_layers)}
        self._vt = {i: None for i in range(num_layers)}
        self._vt_ok = {i: False for i in range(num_layers)}
        self._lens = {i: None for i in range(num_layers)}   # per-layer per-sample lens (layers fill in one by one)
        self._own = {i: [1] for i in range(num_layers)}      # [number of caches sharing this layer's K/V buffers] (copy-on-write, __deepcopy__)
        self._vt_own = {i: [1] for i in range(num_layers)}   # the same for the V^T image
        self._total = 0
        self._nkv = self._hd = self._dp = 0
        self._dev_meta = {}

    # -- reference protocol
    @property
    def key_cache(self):
        return _LayerView(self, "k")

    @property
    def value_cache(self):
        return _LayerView(self, "v")

    @property
    def num_layers(self):
        return self._num_layers

    @property
    def seq_lens(self):
        return self._total if self._k[0] is not None else 0

    def __deepcopy__(self, memo):
        """COPY-ON-WRITE: the copy shares every layer's K / V (and valid V^T) buffer with the original; whichever of the sharers writes a layer first
        (``store``: the in-place append of a one-sample context, or a V^T rebuild) clones that layer for itself -- up to the live rows, keeping the
        buffer's capacity -- and leaves the others untouched.  The reference deep-copies the whole context for every request stream
        (inferencer.py:189,230-231,244,253): an image-edit request would hold three copies of a 9 k-token context (0.5 GB each at 7B) of which two
        are only ever read."""
        c = NaiveCache(self._num_layers)
        c._total, c._nkv, c._hd, c._dp = self._total, self._nkv, self._hd, self._dp
        for i in range(self._num_layers):
            if self._k[i] is not None:
                c._k[i], c._v[i] = self._k[i], self._v[i]
                c._lens[i] = list(self._lens[i])
                c._own[i] = self._own[i]
                self._own[i][0] += 1
                if self._vt_ok[i]:
                    c._vt[i], c._vt_ok[i], c._vt_own[i] = self._vt[i], True, self._vt_own[i]
                    self._vt_own[i][0] += 1
        return c

    def __copy__(self):
        # a shallow copy would alias the owner counters themselves (its __del__ would then un-share buffers that are still shared): same as deepcopy
        return self.__deepcopy__({})

    def unshare(self):
        """Give this cache private K / V storage for every layer (what the reference's deepcopy does eagerly): call before editing
        ``key_cache[l]`` / ``value_cache[l]`` IN PLACE on a cache that came out of ``copy.deepcopy`` or ``concat``."""
        for i in range(self._num_layers):
            if self._k[i] is not None:
                self._writable(i)
                if self._vt_own[i][0] > 1:                 # an edited K / V invalidates a shared V^T image for THIS cache only
                    self._vt_own[i][0] -= 1
                    self._vt_own[i] = [1]
                self._vt[i], self._vt_ok[i] = None, False
        return self

    def _writable(self, layer):
        """Make this cache the only owner of the layer's K / V buffers before they are written."""
        own = self._own[layer]
        if own[0] > 1:
            own[0] -= 1
            n = int(sum(self._lens[layer]))
            k = torch.empty_like(self._k[layer])
            v = torch.empty_like(self._v[layer])
            if n:
                ops.copy_rows(self._k[layer], k, n, k.shape[1])
                ops.copy_rows(self._v[layer], v, n, v.shape[1])
            self._k[layer], self._v[layer] = k, v
            self._own[layer] = [1]

    def __del__(self):
        try:
            for own in list(self._own.values()) + list(self._vt_own.values()):
                own[0] -= 1
        except Exception:
            pass

    @staticmethod
    def concat(caches, batch_sizes):
        """The contexts of several forward streams as ONE cache whose samples are [stream 0's | stream 1's | ...]; a stream
        without context (``None`` or an empty cache, e.g. the CFG-text stream of text->image) contributes ``batch_sizes[s]``
        zero-length samples.  Rows are copied (the result is read-only context for a batched denoise forward)."""
        L = next(c for c in caches if c is not None).num_layers
        out = NaiveCache(L)
        live = [c for c in caches if c is not None and not c.is_empty(0)]
        if not live:
            return out
        out._nkv, out._hd, out._dp = live[0]._nkv, live[0]._hd, live[0]._dp
        for i in range(L):
            ks, vs, lens = [], [], []
            for c, B in zip(caches, batch_sizes):
                if c is None or c.is_empty(i):
                    lens += [0] * B
                    continue
                if len(c._lens[i]) != B or (c._nkv, c._dp) != (out._nkv, out._dp):
                    raise ValueError("NaiveCache.concat: stream caches disagree on the batch size or the head layout")
                n = int(sum(c._lens[i]))
                ks.append(c._k[i][:n])
                vs.append(c._v[i][:n])
                lens += [int(x) for x in c._lens[i]]
            if len(ks) == 1:
                # one stream carries all the context (text->image: the CFG stream has none): share its buffers copy-on-write instead of a second full copy
                src = next(c for c in caches if c is not None and not c.is_empty(i))
                out._k[i], out._v[i], out._own[i] = src._k[i], src._v[i], src._own[i]
                src._own[i][0] += 1
            else:
                out._k[i] = torch.cat(ks).contiguous()
                out._v[i] = torch.cat(vs).contiguous()
            out._lens[i] = lens
        out._total = int(sum(out._lens[0]))
        return out

    # -- engine side
    def is_empty(self, layer):
        return self._k[layer] is None

    def lens(self, layer):
        return self._lens[layer]

    def _meta(self, layer, device):
        """(cu_ctx int32[B+1], vt_col int32[B], vt_cols) for the current per-sample lens."""
        lens = tuple(self._lens[layer])
        m = self._dev_meta.get(lens)
        if m is None:
            cu = [0]
            col = []
            c = 0
            for l in lens:
                cu.append(cu[-1] + l)
                col.append(c)
                c += _ceil_to(max(l, 1), 64)
            m = (torch.tensor(cu, dtype=torch.int32, device=device), torch.tensor(col, dtype=torch.int32, device=device), c,
                 max(lens) if lens else 0)
            self._dev_meta = {lens: m}
        return m

    def ctx_tensors(self, layer):
        """(k, vt, cu_ctx, vt_col) for the attention kernel; builds V^T on first use after a change."""
        k, v = self._k[layer], self._v[layer]
        cu, col, cols, mx = self._meta(layer, k.device)
        if not self._vt_ok[layer]:
            vt = self._vt[layer]
            if vt is not None and self._vt_own[layer][0] > 1:         # shared with a copy that may still read it: rebuild into a buffer of our own
                self._vt_own[layer][0] -= 1
                self._vt_own[layer] = [1]
                vt = None
            if vt is None or vt.shape[1] < cols:
                vt = torch.zeros((self._nkv * self._dp, _ceil_to(cols, 256)), dtype=BF16, device=k.device)
                self._vt[layer] = vt
            ops.v_transpose(v, vt, cu, col, len(self._lens[layer]), mx, self._nkv, self._dp)
            self._vt_ok[layer] = True
        return k, self._vt[layer], cu, col

    def store(self, layer, k_rows, v_rows, q_lens, ctx_lens, nkv, hd, dp, new_dst=None, ctx_dst=None):
        """Merge this forward's K/V rows (views into the fused projection buffer) into the cache.
        Layout after the call is the reference's merged layout [ctx_0 | new_0 | ctx_1 | new_1 | ...]."""
        self._nkv, self._hd, self._dp = nkv, hd, dp
        width = nkv * dp
        B = len(q_lens)
        M = int(sum(q_lens))
        dev = k_rows.device
        if self._k[layer] is None:
            k = torch.empty((_ceil_to(M, 64) if B > 1 else max(_ceil_to(2 * M, 256), 256), width), dtype=BF16, device=dev)
            v = torch.empty_like(k)
            ops.copy_rows(k_rows, k, M, width)
            ops.copy_rows(v_rows, v, M, width)
            self._k[layer], self._v[layer] = k, v
            self._lens[layer] = [int(x) for x in q_lens]
        elif B == 1:
            old = int(self._lens[layer][0])
            need = old + M
            if need <= self._k[layer].shape[0]:
                self._writable(layer)                                  # (a grown buffer below is a fresh one anyway)
            k, v = self._k[layer], self._v[layer]
            if need > k.shape[0]:
                cap = _ceil_to(max(need * 2, 256), 256)
                k2 = torch.empty((cap, width), dtype=BF16, device=dev)
                v2 = torch.empty_like(k2)
                ops.copy_rows(k, k2, old, width)
                ops.copy_rows(v, v2, old, width)
                k, v = k2, v2
                self._k[layer], self._v[layer] = k, v
                self._own[layer][0] -= 1
                self._own[layer] = [1]
            ops.copy_rows(k_rows, k[old:], M, width)
            ops.copy_rows(v_rows, v[old:], M, width)
            self._lens[layer] = [need]
        else:
            old_total = int(sum(self._lens[layer]))
            total = old_total + M
            k2 = torch.empty((_ceil_to(total, 64), width), dtype=BF16, device=dev)
            v2 = torch.empty_like(k2)
            ops.copy_rows(self._k[layer], k2, old_total, width, dst_rows=ctx_dst)
            ops.copy_rows(self._v[layer], v2, old_total, width, dst_rows=ctx_dst)
            ops.copy_rows(k_rows, k2, M, width, dst_rows=new_dst)
            ops.copy_rows(v_rows, v2, M, width, dst_rows=new_dst)
            self._k[layer], self._v[layer] = k2, v2
            self._own[layer][0] -= 1
            self._own[layer] = [1]
            self._lens[layer] = [int(a) + int(b) for a, b in zip(ctx_lens, q_lens)]
        self._vt_ok[layer] = False
        self._total = int(sum(self._lens[layer]))


# ------------------------------------------------------------------------------------------------------------
# parameter holders (names/shapes == the reference's state dict)
# ------------------------------------------------------------------------------------------------------------
class _Linear(nn.Module):
    def __init__(self, in_features, out_features, bias):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features), requires_grad=False) if bias else None


class _NormWeight(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim), requires_grad=False)
        self.variance_epsilon = eps


class _Embedding(nn.Module):
    def __init__(self, n, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, dim), requires_grad=False)


class _MLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.gate_proj = _Linear(cfg.hidden_size, cfg.intermediate_size, False)
        self.up_proj = _Linear(cfg.hidden_size, cfg.intermediate_size, False)
        self.down_proj = _Linear(cfg.intermediate_size, cfg.hidden_size, False)


class _Attention(nn.Module):
    def __init__(self, cfg, layer_idx, mot):
        super().__init__()
        H, nh, nkv = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads
        hd = H // nh
        self.layer_idx = layer_idx
        sufs = ("", "_moe_gen") if mot else ("",)
        for s in sufs:
            setattr(self, "q_proj" + s, _Linear(H, nh * hd, True))
            setattr(self, "k_proj" + s, _Linear(H, nkv * hd, True))
            setattr(self, "v_proj" + s, _Linear(H, nkv * hd, True))
            setattr(self, "o_proj" + s, _Linear(nh * hd, H, False))
            if cfg.qk_norm:
                setattr(self, "q_norm" + s, _NormWeight(hd, cfg.rms_norm_eps))
                setattr(self, "k_norm" + s, _NormWeight(hd, cfg.rms_norm_eps))


class _DecoderLayer(nn.Module):
    """Qwen2DecoderLayer / Qwen2MoEDecoderLayer / Qwen2MoTDecoderLayer parameter sets (qwen2_navit.py:603-940)."""

    def __init__(self, cfg, layer_idx):
        super().__init__()
        kind = cfg.layer_module
        mot = kind == "Qwen2MoTDecoderLayer"
        moe = kind == "Qwen2MoEDecoderLayer"
        if kind not in ("Qwen2DecoderLayer", "Qwen2MoEDecoderLayer", "Qwen2MoTDecoderLayer"):
            raise ValueError(f"unknown layer_module {kind}")
        self.self_attn = _Attention(cfg, layer_idx, mot)
        self.mlp = _MLP(cfg)
        if mot or moe:
            self.mlp_moe_gen = _MLP(cfg)
        self.input_layernorm = _NormWeight(cfg.hidden_size, cfg.rms_norm_eps)
        self.post_attention_layernorm = _NormWeight(cfg.hidden_size, cfg.rms_norm_eps)
        if mot:
            self.input_layernorm_moe_gen = _NormWeight(cfg.hidden_size, cfg.rms_norm_eps)
            self.post_attention_layernorm_moe_gen = _NormWeight(cfg.hidden_size, cfg.rms_norm_eps)


class _Rotary(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        hd = cfg.hidden_size // cfg.num_attention_heads
        if getattr(cfg, "rope_scaling", None):
            raise NotImplementedError("only the 'default' rope type is on BAGEL's path (modeling_qwen2.py:105)")
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float().cpu() / hd))
        # fp32 on purpose, and immune to module.to(bf16): see DESIGN.md (non-persistent buffer, app.py:105-113)
        self._inv_freq_cpu = inv
        self._inv_freq_dev = {}

    def inv_freq(self, device):
        t = self._inv_freq_dev.get(device)
        if t is None:
            t = self._inv_freq_cpu.to(device)
            self._inv_freq_dev[device] = t
        return t


class Qwen2Model(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.use_moe = "Mo" in config.layer_module
        self.embed_tokens = _Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([_DecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = _NormWeight(config.hidden_size, config.rms_norm_eps)
        if self.use_moe:
            self.norm_moe_gen = _NormWeight(config.hidden_size, config.rms_norm_eps)
        self.rotary_emb = _Rotary(config)
        self.enable_taylorseer = False

    def release_train_buffers(self):
        """Free what a training step leaves RESIDENT on this module (train_step.TrainTape): up to TAPE_POOL_SETS flat tape-buffer sets -- 35-75 GB each at
        7B width on 18 k-32 k token packs -- and the keep-gate/up decisions that were taken against the memory that was free then.  Called when the packed
        engine is dropped and before ``Bagel.quantize_language_model``; call it by hand between a training phase and long-context inference."""
        self.__dict__.pop("_tape_pool", None)
        self.__dict__.pop("_keep_gate_up", None)


# ------------------------------------------------------------------------------------------------------------
# per-forward plan (everything the reference recomputes / syncs for inside the layer loop)
# ------------------------------------------------------------------------------------------------------------
def _tolist(t):
    if t is None:
        return []
    if torch.is_tensor(t):
        return [int(x) for x in t.detach().cpu().tolist()]
    return [int(x) for x in t]


class ForwardPlan:
    """Host-side digest of (query_lens, key_values_lens, packed_*_indexes, position ids, MoT index lists)."""

    def __init__(self, device, query_lens, position_ids, packed_query_indexes=None, key_values_lens=None,
                 packed_key_value_indexes=None, text_indexes=None, vae_indexes=None, inv_freq=None, cos_sin=None):
        q = _tolist(query_lens)
        B = len(q)
        c = _tolist(key_values_lens) if key_values_lens is not None else [0] * B
        if len(c) != B:
            raise ValueError("key_values_lens and query_lens disagree on the batch size")
        self.B, self.q_lens, self.ctx_lens = B, q, c
        self.M = sum(q)
        self.max_lq = max(q) if q else 0
        cu_q, cu_c, vcol = [0], [0], []
        col = 0
        new_dst, ctx_dst = [], []
        base = 0
        for b in range(B):
            cu_q.append(cu_q[-1] + q[b])
            cu_c.append(cu_c[-1] + c[b])
            vcol.append(col)
            col += _ceil_to(max(q[b], 1), 64)
            ctx_dst.extend(range(base, base + c[b]))
            new_dst.extend(range(base + c[b], base + c[b] + q[b]))
            base += c[b] + q[b]
        self.vt_cols = col
        self.has_ctx = any(x > 0 for x in c)
        # the merged layout [ctx_0 | q_0 | ctx_1 | q_1 ...] is the only one the reference's packers produce
        # (bagel.py:242-253, 309-338, 560-590); verify instead of silently assuming.
        if packed_query_indexes is not None and _tolist(packed_query_indexes) != new_dst:
            raise NotImplementedError("packed_query_indexes is not the [ctx_b | query_b] merged layout")
        if packed_key_value_indexes is not None and self.has_ctx and _tolist(packed_key_value_indexes) != ctx_dst:
            raise NotImplementedError("packed_key_value_indexes is not the [ctx_b | query_b] merged layout")

        def i32(x):
            return torch.tensor(x, dtype=torch.int32, device=device)

        self.cu_q = i32(cu_q)
        self.vt_new_col = i32(vcol)
        self.new_dst = i32(new_dst) if B > 1 else None
        self.ctx_dst = i32(ctx_dst) if (B > 1 and self.has_ctx) else None
        pos = position_ids if torch.is_tensor(position_ids) else torch.tensor(position_ids, dtype=torch.long)
        self.pos_ids = pos.to(device=device, dtype=torch.long).contiguous()
        if self.pos_ids.numel() != self.M:
            raise ValueError("position ids do not cover the packed query sequence")
        self.cos, self.sin = cos_sin if cos_sin is not None else ops.rope_table(self.pos_ids, inv_freq)
        self.und_side = False      # see MoTEngine.forward: marker rows through the dense side path
        self._attn_plans = {}
        # MoT routing
        self.text_idx = self.vae_idx = self.expert = None
        if text_indexes is not None and vae_indexes is not None:
            t, v = _tolist(text_indexes), _tolist(vae_indexes)
            if len(t) + len(v) != self.M or len(set(t) | set(v)) != self.M:
                raise NotImplementedError("gen mode expects text rows and latent rows to partition the sequence")
            self.text_idx, self.vae_idx = i32(t), i32(v)
            ex = torch.zeros(self.M, dtype=torch.int32)
            ex[torch.tensor(v, dtype=torch.long)] = 1
            self.expert = ex.to(device)
            self.n_text, self.n_vae = len(t), len(v)


def _attn_plan_of(plan, nq, nkv, dp, causal, with_ctx, device):
    """The persistent attention kernel's work list for this forward shape (ops.AttnPlan): built once, shared by every layer and every
    denoise step that reuses the ForwardPlan.  Context rows / V^T columns follow NaiveCache._meta's layout."""
    key = (nq, nkv, dp, bool(causal), bool(with_ctx))
    ap = plan._attn_plans.get(key)
    if ap is None:
        q_start, vcol, c = [], [], 0
        at = 0
        for n in plan.q_lens:
            q_start.append(at); at += n
            vcol.append(c); c += _ceil_to(max(n, 1), 64)
        kw = {}
        if with_ctx:
            cs, ccol, at, c = [], [], 0, 0
            for n in plan.ctx_lens:
                cs.append(at); at += n
                ccol.append(c); c += _ceil_to(max(n, 1), 64)
            kw = dict(ctx_start=cs, ctx_len=plan.ctx_lens, vt_ctx_col=ccol)
        ap = plan._attn_plans[key] = ops.AttnPlan(q_start, plan.q_lens, vcol, nq, nkv, dp, causal, device, **kw)
    return ap


def concat_plans(plans):
    """ONE plan for several forward streams over the same packed batch: the conditional and the CFG forwards of a denoise
    step (bagel.py:820-870: same query sequence, different position ids / context) become the samples [stream 0 | stream 1 | ...]
    of one forward.  Row r of stream s is row s*M + r; every per-sample quantity is simply concatenated (attention never crosses
    samples, every other op is row-wise), and the RoPE tables are the streams' own tables stacked."""
    p0 = plans[0]
    dev = p0.cu_q.device
    q, c, t, v, off = [], [], [], [], 0
    routed = all(x.text_idx is not None for x in plans)
    for x in plans:
        q += x.q_lens
        c += x.ctx_lens
        if routed:
            t += [off + i for i in _tolist(x.text_idx)]
            v += [off + i for i in _tolist(x.vae_idx)]
        off += x.M
    return ForwardPlan(dev, q, torch.cat([x.pos_ids for x in plans]), key_values_lens=c,
                       text_indexes=t if routed else None, vae_indexes=v if routed else None,
                       cos_sin=(torch.cat([x.cos for x in plans]).contiguous(), torch.cat([x.sin for x in plans]).contiguous()))


# ------------------------------------------------------------------------------------------------------------
# training-forward plan: the causal / full / noise block mask as sequences of the varlen attention kernel
# ------------------------------------------------------------------------------------------------------------
def splits_from_mask(mask):
    """Recover (split_lens, attn_modes) from one sample's additive mask built by prepare_attention_mask_per_sample
    (data/data_utils.py:72-103).  Adjacent causal splits merge (same mask); the result is verified by rebuilding."""
    allow = torch.isfinite(mask.detach().float().cpu()) if mask.dtype != torch.bool else mask.cpu()
    n = allow.shape[0]
    idx = torch.arange(n)
    last = torch.where(allow, idx[None, :], torch.full((1, 1), -1)).max(dim=1).values.tolist()
    lens, modes, r = [], [], 0
    while r < n:
        if last[r] > r:                                   # a full / noise block [r, last[r]]
            e = last[r] + 1
            hidden = e < n and not bool(allow[e, r])
            lens.append(e - r); modes.append("noise" if hidden else "full")
            r = e
        else:
            e = r
            while e < n and last[e] == e and not (e + 1 < n and not bool(allow[e + 1, e])):
                e += 1
            if e == r:                                    # a single token hidden from what follows: 1-token noise split
                lens.append(1); modes.append("noise"); r += 1
            else:
                lens.append(e - r); modes.append("causal"); r = e
    rebuilt = torch.zeros((n, n), dtype=torch.bool)
    c = 0
    for L, m in zip(lens, modes):
        rebuilt[c:c + L, :c] = True
        rebuilt[c:c + L, c:c + L] = torch.ones((L, L)).tril().bool() if m == "causal" else True
        c += L
    c = 0
    for L, m in zip(lens, modes):
        if m == "noise":
            rebuilt[:, c:c + L] = False
            rebuilt[c:c + L, c:c + L] = True
        c += L
    if not torch.equal(rebuilt, allow):
        raise NotImplementedError("attention mask is not a causal/full/noise split structure (data_utils.py:72-103)")
    return lens, modes


class TrainPlan:
    """Host digest of (sample_lens, per-sample split structure, und/gen row lists, position ids) for forward_train.

    Every split is one sequence of the attention kernel: its queries/new keys are its own rows, its context is the
    PREFIX of its sample's non-noise ("clean") key stream -- causal splits run with the bottom-right causal flag, full
    and noise splits without.  Two launches per layer cover any mask the reference's packer can produce."""

    def __init__(self, device, sample_lens, sample_splits, position_ids, und_indexes, gen_indexes, inv_freq):
        self.M = int(sum(sample_lens))
        self.sample_lens, self.sample_splits = list(sample_lens), [(list(l), list(m)) for l, m in sample_splits]   # the backward's items
        i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=device)  # noqa: E731
        q_start, q_end, new_col, clean_rows, cu_clean, clean_col = [], [], [], [], [0], []
        groups = {True: dict(qs=[], qe=[], cs=[], ce=[], ncol=[], ccol=[]), False: dict(qs=[], qe=[], cs=[], ce=[], ncol=[], ccol=[])}
        row, col, ccol = 0, 0, 0
        for n, (lens, modes) in zip(sample_lens, sample_splits):
            if sum(lens) != n:
                raise ValueError("split_lens do not add up to sample_lens")
            clean_start, prefix = cu_clean[-1], 0
            clean_col.append(ccol)
            for L, mode in zip(lens, modes):
                if mode not in ("causal", "full", "noise"):
                    raise ValueError(f"unknown attn mode {mode}")
                q_start.append(row); q_end.append(row + L); new_col.append(col)
                g = groups[mode == "causal"]
                g["qs"].append(row); g["qe"].append(row + L); g["cs"].append(clean_start); g["ce"].append(clean_start + prefix)
                g["ncol"].append(col); g["ccol"].append(ccol)
                if mode != "noise":
                    clean_rows.extend(range(row, row + L))
                    prefix += L
                row += L
                col += _ceil_to(max(L, 1), 64)
            cu_clean.append(clean_start + prefix)
            ccol += _ceil_to(max(prefix, 1), 64)
        self.n_splits = len(q_start)
        self.cu_splits = i32(q_start + [row])
        self.new_col = i32(new_col)
        self.vt_cols = col
        self.max_split = max(e - s for s, e in zip(q_start, q_end)) if q_start else 0
        self.n_clean = len(clean_rows)
        self.clean_rows = i32(clean_rows) if clean_rows else None
        self.n_samples = len(sample_lens)
        self.cu_clean = i32(cu_clean)
        self.clean_col = i32(clean_col)
        self.vt_clean_cols = ccol
        self.max_clean = max((b - a) for a, b in zip(cu_clean[:-1], cu_clean[1:])) if sample_lens else 0
        self.groups = []
        for causal in (True, False):
            g = groups[causal]
            if g["qs"]:
                self.groups.append(dict(causal=causal, n=len(g["qs"]), max_lq=max(e - s for s, e in zip(g["qs"], g["qe"])),
                                        qs=i32(g["qs"]), qe=i32(g["qe"]), cs=i32(g["cs"]), ce=i32(g["ce"]), ncol=i32(g["ncol"]),
                                        ccol=i32(g["ccol"])))
        pos = position_ids if torch.is_tensor(position_ids) else torch.tensor(position_ids, dtype=torch.long)
        self.pos_ids = pos.to(device=device, dtype=torch.long).contiguous()
        if self.pos_ids.numel() != self.M:
            raise ValueError("position ids do not cover the packed sequence")
        self.cos, self.sin = ops.rope_table(self.pos_ids, inv_freq)
        u, g_ = _tolist(und_indexes), _tolist(gen_indexes)
        if len(u) + len(g_) != self.M or len(set(u) | set(g_)) != self.M:
            raise NotImplementedError("und and gen token indexes must partition the packed sequence")
        self.text_idx, self.vae_idx = i32(u), (i32(g_) if g_ else None)
        self.n_text, self.n_vae = len(u), len(g_)
        ex = torch.zeros(self.M, dtype=torch.int32)
        if g_:
            ex[torch.tensor(g_, dtype=torch.long)] = 1
        self.expert = ex.to(device)


# ------------------------------------------------------------------------------------------------------------
# engine: packed weights + workspaces + the layer loop
# ------------------------------------------------------------------------------------------------------------
def _pad_heads_rows(w, nheads, hd, dp):
    """[nheads*hd, K] -> [nheads*dp, K] (zero rows for the padded head lanes)."""
    if dp == hd:
        return w
    out = w.new_zeros((nheads, dp) + tuple(w.shape[1:]))
    out[:, :hd] = w.view(nheads, hd, *w.shape[1:])
    return out.view(nheads * dp, *w.shape[1:])


def _pad_heads_cols(w, nheads, hd, dp):
    """[N, nheads*hd] -> [N, nheads*dp]."""
    if dp == hd:
        return w
    out = w.new_zeros((w.shape[0], nheads, dp))
    out[:, :, :hd] = w.view(w.shape[0], nheads, hd)
    return out.view(w.shape[0], nheads * dp)


def interleave_gate_up(gate, up):
    """[I,H],[I,H] -> [2I,H] in 16-row blocks g0..15,u0..15,g16..31,... (BAGEL_EPI_SWIGLU16 layout)."""
    I, H = gate.shape
    assert I % 16 == 0
    return torch.stack((gate.view(I // 16, 16, H), up.view(I // 16, 16, H)), dim=1).reshape(2 * I, H).contiguous()


class _PackedLayer:
    __slots__ = ("wqkv", "bqkv", "wo", "wgu", "wd", "qn", "kn", "ln_in", "ln_post", "wt")    # wt: transposed images for the training backward


class _StoredLayers:
    """``MoTEngine.layers`` of a quantised engine: a sequence whose items are materialised on access.  Layer i's packed bf16 matrices are
    de-quantised into scratch set i % 2 (two sets: a consumer may still hold the previous layer's view while it asks for the next), small
    tensors (biases, norm weights) are kept as they are.  Only the inference forwards iterate it; training and the fp8 option refuse."""
    MATS = ("wqkv", "wo", "wgu", "wd")

    def __init__(self, eng, modules, kind):
        self.eng, self.kind = eng, kind
        quant = ops.quantize_nf4 if kind == "nf4" else ops.quantize_rows_i8
        self.stored, self.small = [], []
        for L in modules:
            P = eng._pack_layer(L)
            for name in self.MATS:
                for w in getattr(P, name):
                    if kind == "nf4" and w.shape[1] % 64:
                        raise NotImplementedError("nf4 weight_store needs row lengths that are multiples of the 64-weight block")
            self.stored.append({name: [quant(w) for w in getattr(P, name)] for name in self.MATS})
            self.small.append({n: getattr(P, n) for n in ("bqkv", "qn", "kn", "ln_in", "ln_post")})
            del P                                              # the bf16 copies of this layer are gone before the next one is packed
        self.scratch = [None, None]

    def __len__(self):
        return len(self.stored)

    def __iter__(self):
        return (self[i] for i in range(len(self.stored)))

    def __getitem__(self, i):
        st = self.stored[i]
        slot = i % 2
        if self.scratch[slot] is None:
            self.scratch[slot] = {name: [torch.empty((q.shape[0], q.shape[1] * (2 if self.kind == "nf4" else 1)), dtype=BF16, device=q.device)
                                         for q, _ in st[name]] for name in self.MATS}
        deq = ops.dequantize_nf4 if self.kind == "nf4" else ops.dequantize_rows_i8
        P = _PackedLayer()
        for name in self.MATS:
            setattr(P, name, [deq(q, sc, out) for (q, sc), out in zip(st[name], self.scratch[slot][name])])
        for n, v in self.small[i].items():
            setattr(P, n, v)
        P.wt = {}
        return P

    def small_views(self):
        """The layers WITHOUT their matrices (biases and norm weights only): what the Lq = 1 decode needs beside the stored codes."""
        out = []
        for sm in self.small:
            P = _PackedLayer()
            for name in self.MATS:
                setattr(P, name, None)
            for n, v in sm.items():
                setattr(P, n, v)
            P.wt = {}
            out.append(P)
        return out

    def resident_bytes(self):
        return sum(q.numel() * q.element_size() + sc.numel() * sc.element_size() for st in self.stored for ms in st.values() for q, sc in ms)


class Fp8DelayedScales:
    """State of the DELAYED row scales of the FP8 gen expert's SwiGLU output across the forwards of ONE stream set of a denoise loop (the same latent rows at
    consecutive timesteps): per layer and physical row the max |value| the last forward saw (collected by the gate/up GEMM's epilogue with atomicMax) and the
    scale in use.  ``margin`` = headroom factor between what was seen and what fits the e4m3 range (2.0: a row may double between two steps before it
    saturates -- saturates, not overflows: the epilogue clamps).  ``oracle/fp8.py::DelayedScales`` restates the scheme."""

    def __init__(self, margin=2.0):
        self.margin = float(margin)
        self.amax = self.scale = None
        self.primed = False

    def fit(self, layers, rows, device):
        if self.amax is None or self.amax.shape != (layers, rows) or self.amax.device != torch.device(device):
            self.amax = torch.zeros((layers, rows), dtype=torch.float32, device=device)
            self.scale = torch.ones((layers, rows), dtype=torch.float32, device=device)
            self.primed = False


class MoTEngine:
    """Owns the MI355X-layout copies of a Qwen2Model's weights and runs forward_inference on them."""

    def __init__(self, model: "Qwen2Model", lm_head: "_Linear", weight_store=None):
        cfg = model.config
        self.cfg = cfg
        self.H, self.I = cfg.hidden_size, cfg.intermediate_size
        self.nq, self.nkv = cfg.num_attention_heads, cfg.num_key_value_heads
        self.hd = self.H // self.nq
        self.dp = padded_head_dim(self.hd)
        self.eps = cfg.rms_norm_eps
        self.kind = cfg.layer_module
        self.mot = self.kind == "Qwen2MoTDecoderLayer"
        self.moe_mlp = self.kind in ("Qwen2MoTDecoderLayer", "Qwen2MoEDecoderLayer")
        self.use_norm = bool(cfg.qk_norm)
        if self.H % 64 or self.I % 64 or (self.nq * self.dp) % 64:
            raise NotImplementedError("hidden/intermediate sizes must be multiples of 64 for the GEMM K loop")
        p0 = model.embed_tokens.weight
        ops.require_gpu_bf16(p0, "MoTEngine")
        self.device = p0.device
        self.model = model
        self.lm_head = lm_head
        # WHOLE-MODEL 4- / 8-bit load modes (app.py:114-131: bitsandbytes NF4 / INT8 over every nn.Linear of the language model).  weight_store =
        # "nf4" | "int8_rowwise": the four matrices of every decoder layer (both experts) stay resident as codes + scales; `layers[i]` materialises that
        # layer's bf16 matrices into one of two scratch sets right before they are used (_StoredLayers) -- exactly what bitsandbytes' matmul_4bit
        # does in front of F.linear for more than one activation row -- and the Lq = 1 decode runs its own 4- / 8-bit gemv kernels on the stored
        # codes.  Biases, norm weights, embeddings, lm_head stay bf16 (llm_int8_skip_modules / the library never touches non-Linear weights).
        if weight_store == "int8":
            raise NotImplementedError(LLM_INT8_NOT_BUILT)
        if weight_store not in (None, "nf4", "int8_rowwise"):
            raise NotImplementedError(f"weight_store={weight_store!r}: 'nf4' (bitsandbytes NF4: blocks of 64, fp32 absmax) and 'int8_rowwise' (row-wise absmax) are built")
        self.weight_store = weight_store
        if weight_store is None:
            self.layers = [self._pack_layer(l) for l in model.layers]
        else:
            self.layers = _StoredLayers(self, model.layers, weight_store)
        self._ws = {}
        self._ws_side = {}
        self._fp8 = None
        self._ws_fp8 = {}
        self._wt_extra = {}            # name -> (transposed image, parameter): see wt_of()

    def _pack_layer(self, L):
        nq, nkv, hd, dp = self.nq, self.nkv, self.hd, self.dp
        a = L.self_attn
        if a.q_proj.weight.numel() == 0:
            raise RuntimeError("the bf16 projection weights of this model were released by quantize_language_model(release_bf16=True) and the quantised "
                               "engine built from them has since been dropped (module.to() / load_state_dict / an in-place rewrite): reload the checkpoint")
        P = _PackedLayer()
        P.wqkv, P.bqkv, P.wo, P.wgu, P.wd, P.qn, P.kn, P.ln_in, P.ln_post = [], [], [], [], [], [], [], [], []
        P.wt = {}
        attn_sufs = ("", "_moe_gen") if self.mot else ("",)
        for s in attn_sufs:
            q, k, v, o = (getattr(a, n + s) for n in ("q_proj", "k_proj", "v_proj", "o_proj"))
            P.wqkv.append(torch.cat([_pad_heads_rows(q.weight.data, nq, hd, dp), _pad_heads_rows(k.weight.data, nkv, hd, dp),
                                     _pad_heads_rows(v.weight.data, nkv, hd, dp)], 0).contiguous())
            P.bqkv.append(torch.cat([_pad_heads_rows(q.bias.data, nq, hd, dp), _pad_heads_rows(k.bias.data, nkv, hd, dp),
                                     _pad_heads_rows(v.bias.data, nkv, hd, dp)], 0).contiguous())
            P.wo.append(_pad_heads_cols(o.weight.data, nq, hd, dp).contiguous())
            if self.use_norm:
                P.qn.append(getattr(a, "q_norm" + s).weight.data.contiguous())
                P.kn.append(getattr(a, "k_norm" + s).weight.data.contiguous())
            P.ln_in.append(getattr(L, "input_layernorm" + s).weight.data.contiguous())
            P.ln_post.append(getattr(L, "post_attention_layernorm" + s).weight.data.contiguous())
        for s in (("", "_moe_gen") if self.moe_mlp else ("",)):
            m = getattr(L, "mlp" + s)
            P.wgu.append(interleave_gate_up(m.gate_proj.weight.data, m.up_proj.weight.data))
            P.wd.append(m.down_proj.weight.data.contiguous())
        return P

    def refresh(self):
        """The parameters were rewritten IN PLACE (an optimizer step: same storage, new values): re-pack every layer INTO the packed
        tensors that exist, rewrite the cached transposed images of the training backward in place, forget the quantised copies.
        Workspaces, plans and the tape pool stay -- a training loop pays one re-pack pass per step (~2 bytes read + written per
        parameter and image), not a rebuild of the engine, re-allocation of 28 GB of packed weights and a fresh set of transposes."""
        from . import train_step as TS
        if self.weight_store is not None:
            raise NotImplementedError("a quantised engine is rebuilt, not refreshed (weight_store is an inference load mode)")
        for P, L in zip(self.layers, self.model.layers):
            fresh = self._pack_layer(L)
            for name in ("wqkv", "bqkv", "wo", "wgu", "wd", "qn", "kn", "ln_in", "ln_post"):
                for old, new in zip(getattr(P, name), getattr(fresh, name)):
                    if old.data_ptr() != new.data_ptr():        # (norm weights are views of the parameters themselves: already current)
                        old.copy_(new)
            for name, imgs in P.wt.items():
                for img, w in zip(imgs, getattr(P, name)):
                    img.copy_(TS._wt(w))
            del fresh
        for name, (img, src, _) in list(self._wt_extra.items()):
            img.copy_(TS._wt(src.data))
            self._wt_extra[name] = (img, src, src._version)
        self._fp8 = None
        self._ws_fp8 = {}
        for attr in ("_w8_cache", "_w4_cache", "_nf4_cache"):
            if hasattr(self, attr):
                delattr(self, attr)

    def wt_of(self, name, param):
        """Cached transposed image of a parameter outside the decoder layers (lm_head, llm2vae, connector, time embedder) for the
        training backward; refreshed in place by ``refresh()``."""
        from . import train_step as TS
        ent = self._wt_extra.get(name)
        if ent is None or ent[1] is not param or ent[0].shape[0] != param.shape[1]:
            ent = self._wt_extra[name] = (TS._wt(param.data), param, param._version)
        elif ent[2] != param._version:      # rewritten in place since (these parameters are outside the language model's own signature)
            ent[0].copy_(TS._wt(param.data))
            ent = self._wt_extra[name] = (ent[0], param, param._version)
        return ent[0]

    # -- workspaces (cached per row count; everything stays resident in HBM)
    def workspace(self, M, vt_cols):
        key = (M, vt_cols)
        ws = self._ws.get(key)
        if ws is None:
            dev = self.device
            e = lambda *s: torch.empty(s, dtype=BF16, device=dev)  # noqa: E731
            ws = dict(x=e(M, self.H), h=e(M, self.H), qkv=e(M, (self.nq + 2 * self.nkv) * self.dp),
                      attn=e(M, self.nq * self.dp), act=e(M, self.I),
                      vt=torch.zeros((self.nkv * self.dp, _ceil_to(vt_cols, 256)), dtype=BF16, device=dev))
            if len(self._ws) > 6:
                self._ws.pop(next(iter(self._ws)))
            self._ws[key] = ws
        return ws

    def release_workspaces(self, quantised=False):
        """Forget the cached activation workspaces (they come back, sized for the next request's row count, on its first forward) -- a serving process moving
        from batch-4 text->image to one edit request gives back 10+ GB of live bytes; ``quantised=True`` also drops the on-first-use quantised copies of the
        weights (FP8 gen expert, the decode's INT8 / MXFP4 / NF4 codes).  Nothing of the weights themselves is touched."""
        self._ws, self._ws_side, self._ws_fp8 = {}, {}, {}
        if quantised:
            self._fp8 = None
            for attr in ("_w8_cache", "_w4_cache", "_nf4_cache"):
                if hasattr(self, attr):
                    delattr(self, attr)

    def plan(self, query_lens, position_ids, **kw):
        return ForwardPlan(self.device, query_lens, position_ids, inv_freq=self.model.rotary_emb.inv_freq(self.device), **kw)

    def fp8_weights(self):
        """OCP e4m3 copies (row-wise absmax scales) of the gen expert's four projections, quantised on first use (option
        ``gen_weight_quant = "fp8"``: the MI355X counterpart of the reference's quantised load modes, app.py:114-131)."""
        if self._fp8 is None:
            if self.weight_store is not None:
                raise NotImplementedError("gen_weight_quant='fp8' and weight_store are two different load modes: pick one")
            if not self.mot:
                raise NotImplementedError("gen_weight_quant='fp8' is built for the MoT layer kind (a separate gen expert)")
            for n, k in (("H", self.H), ("I", self.I), ("nq*dp", self.nq * self.dp)):
                if k % 128:
                    raise NotImplementedError(f"gen_weight_quant='fp8' needs {n} = {k} to be a multiple of 128 (one fp8 k-tile)")
            self._fp8 = [dict(wqkv=ops.quantize_rows_fp8(P.wqkv[1]), wo=ops.quantize_rows_fp8(P.wo[1]), wgu=ops.quantize_rows_fp8(P.wgu[1]),
                              wd=ops.quantize_rows_fp8(P.wd[1])) for P in self.layers]
        return self._fp8

    def forward(self, seq, plan: ForwardPlan, mode="und", cache: NaiveCache = None, update=True, causal=True,
                num_layers=None, taylor=None, final_norm=True, gen_quant=None, fp8_state=None):
        """Qwen2Model.forward_inference (qwen2_navit.py:1018-1092).  ``seq`` is not modified.
        ``taylor``: a TaylorSeerState (cache_utils/taylorseer.py) -> the TaylorSeer hooks of :1034-1037,1057-1061,
        1086-1087 are active: a 'full' step runs the layers and refreshes the feature cache, a 'Taylor' step replaces
        the whole layer stack by the cached extrapolation."""
        if seq.shape != (plan.M, self.H):
            raise ValueError(f"packed sequence shape {tuple(seq.shape)} != ({plan.M}, {self.H})")
        gen = mode == "gen" and self.moe_mlp
        if gen and plan.expert is None:
            raise AssertionError("gen mode needs packed_vae_token_indexes and packed_text_indexes")  # qwen2_navit.py:1049-1050
        gen_attn = gen and self.mot
        ws = self.workspace(plan.M, plan.vt_cols)
        x, h, qkv, att, act, vt = ws["x"], ws["h"], ws["qkv"], ws["attn"], ws["act"], ws["vt"]
        # ``taylor`` may be a list of states, one per forward stream of a stream-batched plan (concat_plans; stream s = rows
        # [s*M/S, (s+1)*M/S)).  The streams advance separately (a step without CFG moves only the first one), so their step types can
        # differ: the layers run unless EVERY stream extrapolates; afterwards each stream either refreshes its cache from its slice or
        # replaces its slice by its extrapolation -- per stream exactly what the sequential forwards do.
        streams = list(taylor) if isinstance(taylor, (list, tuple)) else None
        if streams is not None:
            kinds = [st.next_type() for st in streams]
            skip_layers = all(k == "Taylor" for k in kinds)
            taylor = None
        else:
            skip_layers = taylor is not None and taylor.next_type() == "Taylor"
        if skip_layers and update:
            raise ValueError("a TaylorSeer-skipped forward cannot update the KV cache")
        if not skip_layers:
            x.copy_(seq)
        nq, nkv, dp, hd = self.nq, self.nkv, self.dp, self.hd
        qw, kw_ = nq * dp, nkv * dp
        q_v, k_v, v_v = qkv[:, :qw], qkv[:, qw:qw + kw_], qkv[:, qw + kw_:]
        expert = plan.expert if gen else None
        scale = hd ** -0.5

        def groups(w, b=None, on=gen):
            """kwargs selecting one (und) or two (und, gen) GEMM row groups."""
            if on:
                return dict(W0=w[0], bias0=None if b is None else b[0], a_rows0=plan.text_idx, c_rows0=plan.text_idx,
                            M0=plan.n_text, W1=w[1], bias1=None if b is None else b[1], a_rows1=plan.vae_idx,
                            c_rows1=plan.vae_idx, M1=plan.n_vae)
            return dict(W0=w[0], bias0=None if b is None else b[0], M0=plan.M)

        nl = len(self.layers) if num_layers is None else num_layers
        if skip_layers:
            nl = 0
            if streams is None:
                taylor.eval_into(x)
        # Marker-row side path (plan.und_side, MoT gen mode).  The und group of a denoise forward is 2 rows per sample; as a row
        # group of the tile GEMM it costs a whole 256-row tile row in every projection.  With the gen rows alone the tile count of
        # a stream-batched forward (concat_plans) is an exact multiple of the 256 persistent workgroups (2 x 4 x 4096 rows = 128
        # row tiles), so the marker rows travel beside it as a small dense matrix [n_text, H]: their projections are weight-streaming
        # skinny GEMMs, their q/k/v rows are scattered into the fused projection buffer before the attention and their attention
        # rows gathered after it.  Same operators, same rounding points; only the GEMM kernel that serves those rows differs.
        fp8 = delayed = None
        if gen_quant is not None and nl > 0 and gen:
            if gen_quant != "fp8":
                raise NotImplementedError(f"gen_weight_quant={gen_quant!r}: only 'fp8' (OCP e4m3, row-wise scales) is built")
            if not (gen_attn and 2 <= plan.n_text <= ops.SKINNY_MAX_ROWS):
                raise NotImplementedError("gen_weight_quant='fp8' needs the MoT layer kind and 2..64 marker rows (they take the bf16 side path)")
            fp8 = self.fp8_weights()
            fw = self._ws_fp8.get(plan.M)
            if fw is None:
                u8 = lambda *s: torch.empty(s, dtype=torch.uint8, device=self.device)  # noqa: E731
                f32 = lambda: torch.empty((plan.M,), dtype=torch.float32, device=self.device)  # noqa: E731
                if len(self._ws_fp8) > 3:
                    self._ws_fp8.pop(next(iter(self._ws_fp8)))
                fw = self._ws_fp8[plan.M] = dict(hq=u8(plan.M, self.H), sh=f32(), aq=u8(plan.M, nq * dp), sa=f32(), cq=u8(plan.M, self.I), sc=f32())
            # DELAYED scaling of the SwiGLU output (Fp8DelayedScales: a denoise loop hands the same state to every step of one forward stream set): from the
            # second forward on the gate/up GEMM writes e4m3 bytes itself, scaled by what the previous step's rows reached -- no bf16 round trip of the
            # [M, I] activation and no quantiser pass in front of the down projection.  The first forward (no history) and callers without a state take the
            # exact path and, with a state, leave their row maxima behind.
            if fp8_state is not None:
                fp8_state.fit(nl, plan.M, self.device)
                delayed = fp8_state
        side = bool(gen_attn and (plan.und_side or fp8 is not None) and nl > 0 and 2 <= plan.n_text <= ops.SKINNY_MAX_ROWS)
        if side:
            nt = plan.n_text
            sw = self._ws_side.get(nt)
            if sw is None:
                e = lambda *s: torch.empty(s, dtype=BF16, device=self.device)  # noqa: E731
                sw = self._ws_side[nt] = dict(x=e(nt, self.H), h=e(nt, self.H), qkv=e(nt, (nq + 2 * nkv) * dp), attn=e(nt, nq * dp),
                                              act=e(nt, self.I))
            xu, hu, qu, au, actu = sw["x"], sw["h"], sw["qkv"], sw["attn"], sw["act"]
            ops.copy_rows(x, xu, nt, self.H, src_rows=plan.text_idx)

            def gen_only(w, b=None):
                return dict(W0=w[1], bias0=None if b is None else b[1], a_rows0=plan.vae_idx, c_rows0=plan.vae_idx, M0=plan.n_vae)
        for li in range(nl):
            P = self.layers[li]
            if fp8 is None:
                ops.rmsnorm(x, P.ln_in[0], h, self.eps, w1=P.ln_in[1] if gen_attn else None, expert=expert if gen_attn else None)
            if side:
                ops.rmsnorm(xu, P.ln_in[0], hu, self.eps)
                ops.gemm(hu, P.wqkv[0], qu, bias0=P.bqkv[0])
                ops.copy_rows(qu, qkv, nt, qu.shape[1], dst_rows=plan.text_idx)
                if fp8 is not None:
                    # FP8 option: the latent rows' operands are e4m3 with row scales (the marker rows stay on the bf16 side path).  The
                    # norm writes the quantised operand directly; attention output and SwiGLU output get one quantiser pass each.
                    Q = fp8[li]
                    ops.rmsnorm_fp8(x, P.ln_in[1], fw["hq"], fw["sh"], self.eps)
                    ops.gemm_fp8(fw["hq"], fw["sh"], Q["wqkv"][0], Q["wqkv"][1], qkv, bias=P.bqkv[1], rows=plan.vae_idx, M=plan.n_vae)
                else:
                    ops.gemm(h, C=qkv, **gen_only(P.wqkv, P.bqkv))
            else:
                ops.gemm(h, C=qkv, **groups(P.wqkv, P.bqkv, gen_attn))
            ops.qknorm_rope(qkv, plan.cos, plan.sin, P.qn[0] if self.use_norm else None, P.kn[0] if self.use_norm else None,
                            P.qn[1] if (self.use_norm and gen_attn) else None, P.kn[1] if (self.use_norm and gen_attn) else None,
                            expert if gen_attn else None, nq, nkv, hd, dp, self.eps, gen_mode=(mode == "gen" and self.mot),
                            use_norm=self.use_norm)
            ops.v_transpose(v_v, vt, plan.cu_q, plan.vt_new_col, plan.B, plan.max_lq, nkv, dp)
            ctx = None
            if cache is not None and not cache.is_empty(li) and plan.has_ctx:
                if list(cache.lens(li)) != plan.ctx_lens:
                    raise ValueError("key_values_lens does not match the KV cache contents")
                ctx = cache.ctx_tensors(li)
            if ATTN_PLANNED:
                # persistent kernel on a host-built work list (csrc/attention2.hip): head-per-wave tail tiles, key-split leftovers
                ap = _attn_plan_of(plan, nq, nkv, dp, causal, ctx is not None, x.device)
                ops.attn_planned(q_v, k_v, vt, att, ap, scale, k_ctx=None if ctx is None else ctx[0], vt_ctx=None if ctx is None else ctx[1])
            else:
                ops.attn_varlen(q_v, k_v, vt, att, plan.cu_q, plan.vt_new_col, plan.B, plan.max_lq, nq, nkv, dp, causal, scale,
                                k_ctx=None if ctx is None else ctx[0], vt_ctx=None if ctx is None else ctx[1],
                                cu_ctx=None if ctx is None else ctx[2], vt_ctx_col=None if ctx is None else ctx[3])
            if update:
                if cache is None:
                    raise ValueError("update_past_key_values=True needs a NaiveCache")
                cache.store(li, k_v, v_v, plan.q_lens, plan.ctx_lens, nkv, hd, dp, plan.new_dst, plan.ctx_dst)
            if side:
                ops.copy_rows(att, au, nt, au.shape[1], src_rows=plan.text_idx)
                ops.gemm(au, P.wo[0], xu, residual=xu)
                if fp8 is not None:
                    ops.quantize_rows_fp8(att, fw["aq"], fw["sa"])
                    ops.gemm_fp8(fw["aq"], fw["sa"], Q["wo"][0], Q["wo"][1], x, rows=plan.vae_idx, M=plan.n_vae, residual=x)
                    ops.rmsnorm(xu, P.ln_post[0], hu, self.eps)
                    ops.rmsnorm_fp8(x, P.ln_post[1], fw["hq"], fw["sh"], self.eps)
                    ops.gemm(hu, P.wgu[0], actu, epilogue=ops.EPI_SWIGLU16)
                    if delayed is not None and delayed.primed:
                        ops.fp8_delayed_scales(delayed.amax[li], delayed.scale[li], rows=plan.vae_idx, n=plan.n_vae, margin=delayed.margin)
                        ops.gemm_fp8_swiglu_q8(fw["hq"], fw["sh"], Q["wgu"][0], Q["wgu"][1], fw["cq"], delayed.scale[li], delayed.amax[li], rows=plan.vae_idx, M=plan.n_vae)
                        ops.gemm(actu, P.wd[0], xu, residual=xu)
                        ops.gemm_fp8(fw["cq"], delayed.scale[li], Q["wd"][0], Q["wd"][1], x, rows=plan.vae_idx, M=plan.n_vae, residual=x)
                        continue
                    ops.gemm_fp8(fw["hq"], fw["sh"], Q["wgu"][0], Q["wgu"][1], act, rows=plan.vae_idx, M=plan.n_vae, epilogue=ops.EPI_SWIGLU16)
                    ops.gemm(actu, P.wd[0], xu, residual=xu)
                    ops.quantize_rows_fp8(act, fw["cq"], fw["sc"])
                    if delayed is not None:          # the exact scale is rowmax / 448: leave the row maxima for the next forward
                        torch.mul(fw["sc"], 448.0, out=delayed.amax[li])
                    ops.gemm_fp8(fw["cq"], fw["sc"], Q["wd"][0], Q["wd"][1], x, rows=plan.vae_idx, M=plan.n_vae, residual=x)
                    continue
                ops.gemm(att, C=x, residual=x, **gen_only(P.wo))
                ops.rmsnorm(xu, P.ln_post[0], hu, self.eps)
                ops.rmsnorm(x, P.ln_post[0], h, self.eps, w1=P.ln_post[1], expert=expert)
                ops.gemm(hu, P.wgu[0], actu, epilogue=ops.EPI_SWIGLU16)
                ops.gemm(h, C=act, epilogue=ops.EPI_SWIGLU16, **gen_only(P.wgu))
                ops.gemm(actu, P.wd[0], xu, residual=xu)
                ops.gemm(act, C=x, residual=x, **gen_only(P.wd))
                continue
            ops.gemm(att, C=x, residual=x, **groups(P.wo, None, gen_attn))
            ops.rmsnorm(x, P.ln_post[0], h, self.eps, w1=P.ln_post[1] if gen_attn else None, expert=expert if gen_attn else None)
            ops.gemm(h, C=act, epilogue=ops.EPI_SWIGLU16, **groups(P.wgu, None, gen))
            ops.gemm(act, C=x, residual=x, **groups(P.wd, None, gen))
        if side:
            ops.copy_rows(xu, x, nt, self.H, dst_rows=plan.text_idx)
        if delayed is not None:
            delayed.primed = True            # every layer has left its rows' maxima behind: the next forward of this stream set quantises with them
        if taylor is not None:
            if not skip_layers:
                taylor.update(x)
            taylor.advance()
        if streams is not None:
            rows = plan.M // len(streams)
            for si, (st, kind) in enumerate(zip(streams, kinds)):
                xs = x[si * rows:(si + 1) * rows]
                if kind == "Taylor":
                    st.eval_into(xs)
                else:
                    st.update(xs)
                st.advance()
        if not final_norm:
            return x.clone()          # the raw residual stream after the last executed layer (full-size parity probes)
        out = torch.empty_like(x)
        m = self.model
        if gen and self.mot or (gen and self.kind == "Qwen2MoEDecoderLayer"):
            ops.rmsnorm(x, m.norm.weight.data, out, self.eps, w1=m.norm_moe_gen.weight.data, expert=expert)
        else:
            ops.rmsnorm(x, m.norm.weight.data, out, self.eps)
        return out


def _engine_forward_train(self, seq, tp: "TrainPlan", tape=None):
    """Qwen2Model.forward_train (qwen2_navit.py:970-1016) with Qwen2MoTDecoderLayer.forward_train (:713-755) and
    PackedAttentionMoT.forward_train (:406-497): und rows (text + ViT) on the und expert, gen rows (VAE latents) on the gen
    expert, bf16 cast points for BOTH experts' QK-norm (no fp32 path here, unlike forward_inference's gen mode), and the
    block mask executed as per-split sequences (TrainPlan).  With a ``tape`` (train_step.TrainTape) every layer writes its residual
    streams, raw projection, attention output and SwiGLU output into buffers of its own, which the tape keeps for the backward."""
    if self.weight_store is not None:
        raise NotImplementedError("training runs on bf16 weights; weight_store is an inference load mode (app.py:114-131)")
    if seq.shape != (tp.M, self.H):
        raise ValueError(f"packed sequence shape {tuple(seq.shape)} != ({tp.M}, {self.H})")
    dev = self.device
    nq, nkv, dp, hd = self.nq, self.nkv, self.dp, self.hd
    qw, kw_ = nq * dp, nkv * dp
    M = tp.M
    e = lambda *s: torch.empty(s, dtype=BF16, device=dev)  # noqa: E731
    x, h, qkv, att, act = e(M, self.H), e(M, self.H), e(M, qw + 2 * kw_), e(M, qw), e(M, self.I)
    vt = torch.zeros((kw_, _ceil_to(max(tp.vt_cols, 1), 256)), dtype=BF16, device=dev)
    k_clean = torch.zeros((_ceil_to(tp.n_clean + 64, 64), kw_), dtype=BF16, device=dev)
    v_clean = torch.zeros_like(k_clean)
    vt_clean = torch.zeros((kw_, _ceil_to(max(tp.vt_clean_cols, 1), 256)), dtype=BF16, device=dev)
    if tape is not None:
        tape.tp = tp
        tape.decide_gate_up(self, M)
        tape.begin(self, (M, tape.keep_gate_up))
        tb = lambda name, li, *shape, dtype=BF16: tape.buf(name, li, *shape, dtype=dtype, device=dev)  # noqa: E731
        x = tb("x", 0, M, self.H)
        tape.x.append(x)
    x.copy_(seq)
    q_v, k_v, v_v = qkv[:, :qw], qkv[:, qw:qw + kw_], qkv[:, qw + kw_:]
    two = tp.n_vae > 0
    # which parts are per modality: everything for MoT; the MLP and the model's final norm for Qwen2MoEDecoderLayer (shared attention and layer
    # norms, qwen2_navit.py:852-883,1003-1012); nothing for Qwen2DecoderLayer (:620-646)
    two_attn, two_mlp = two and self.mot, two and self.moe_mlp
    expert = tp.expert if two_attn else None
    scale = hd ** -0.5

    def groups(w, b=None, two=None):
        two = two_attn if two is None else two
        if two:
            return dict(W0=w[0], bias0=None if b is None else b[0], a_rows0=tp.text_idx, c_rows0=tp.text_idx, M0=tp.n_text,
                        W1=w[1], bias1=None if b is None else b[1], a_rows1=tp.vae_idx, c_rows1=tp.vae_idx, M1=tp.n_vae)
        return dict(W0=w[0], bias0=None if b is None else b[0], M0=M)

    for P in self.layers:
        x_mid, x_out = x, x                                   # in place without a tape
        if tape is not None:
            li = len(tape.x_mid)
            raw, att, act, x_mid, x_out = (tb("raw", li, M, qw + 2 * kw_), tb("att", li, M, qw), tb("act", li, M, self.I), tb("x_mid", li, M, self.H),
                                           tb("x", li + 1, M, self.H))
            lse = tb("lse", li, nq, M, dtype=torch.float32)
            tape.qkv_raw.append(raw); tape.att.append(att); tape.act.append(act); tape.x_mid.append(x_mid); tape.x.append(x_out); tape.lse.append(lse)
        ops.rmsnorm(x, P.ln_in[0], h, self.eps, w1=P.ln_in[1] if two_attn else None, expert=expert)
        if tape is not None:
            ops.gemm(h, C=raw, **groups(P.wqkv, P.bqkv))
            qkv.copy_(raw)
        else:
            ops.gemm(h, C=qkv, **groups(P.wqkv, P.bqkv))
        ops.qknorm_rope(qkv, tp.cos, tp.sin, P.qn[0] if self.use_norm else None, P.kn[0] if self.use_norm else None,
                        P.qn[1] if (self.use_norm and two_attn) else None, P.kn[1] if (self.use_norm and two_attn) else None,
                        expert, nq, nkv, hd, dp, self.eps, gen_mode=False, use_norm=self.use_norm)
        ops.v_transpose(v_v, vt, tp.cu_splits, tp.new_col, tp.n_splits, tp.max_split, nkv, dp)
        if tp.n_clean:
            ops.copy_rows(k_v, k_clean, tp.n_clean, kw_, src_rows=tp.clean_rows)
            ops.copy_rows(v_v, v_clean, tp.n_clean, kw_, src_rows=tp.clean_rows)
            ops.v_transpose(v_clean, vt_clean, tp.cu_clean, tp.clean_col, tp.n_samples, tp.max_clean, nkv, dp)
        for g in tp.groups:
            ops.attn_varlen_ranges(q_v, k_v, vt, att, g["qs"], g["qe"], g["ncol"], g["n"], g["max_lq"], nq, nkv, dp, g["causal"], scale,
                                   k_ctx=k_clean, vt_ctx=vt_clean, ctx_start=g["cs"], ctx_end=g["ce"], vt_ctx_col=g["ccol"],
                                   lse=lse if tape is not None else None)
        ops.gemm(att, C=x_mid, residual=x, **groups(P.wo))
        ops.rmsnorm(x_mid, P.ln_post[0], h, self.eps, w1=P.ln_post[1] if two_attn else None, expert=expert)
        if tape is not None and tape.keep_gate_up:
            gu = tb("gu", len(tape.gu), M, 2 * self.I)        # the tape keeps the un-activated projection: no recompute in the backward
            tape.gu.append(gu)
            ops.gemm(h, C=gu, **groups(P.wgu, two=two_mlp))
            ops.swiglu_fwd(gu, act)                           # same bits as the fused epilogue
        else:
            ops.gemm(h, C=act, epilogue=ops.EPI_SWIGLU16, **groups(P.wgu, two=two_mlp))
        ops.gemm(act, C=x_out, residual=x_mid, **groups(P.wd, two=two_mlp))
        x = x_out
    out = torch.empty_like(x)
    m = self.model
    ops.rmsnorm(x, m.norm.weight.data, out, self.eps, w1=m.norm_moe_gen.weight.data if two_mlp else None, expert=tp.expert if two_mlp else None)
    return out


MoTEngine.forward_train = _engine_forward_train


class Qwen2ForCausalLM(PackedWeights):
    """Same constructor, attribute names and ``forward_inference`` signature as qwen2_navit.py:1095-1188.  The engine's packed
    weight copies follow the parameters (modeling/packed.py): ``.to()``, any ``load_state_dict`` incl. one on the parent Bagel,
    and in-place rewrites all drop them."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = Qwen2Model(config)
        self.vocab_size = config.vocab_size
        self.lm_head = _Linear(config.hidden_size, config.vocab_size, False)
        self._engine = None
        self._plans = {}

    # ---- weight lifecycle -------------------------------------------------------------------------------
    def init_moe(self):
        """Copy every *_moe_gen parameter from its und twin (qwen2_navit.py:1107-1111)."""
        sd = self.state_dict()
        for name, param in self.named_parameters():
            if "moe_gen" in name:
                param.data.copy_(sd[name.replace("_moe_gen", "")].data)
        self.invalidate_packed()

    def _drop_packed(self):
        self._engine = None
        self._plans = {}
        self.model.release_train_buffers()

    def _check_packed(self):
        """Parameters rewritten in place (every tensor still at its address, version counters moved: an optimizer step, ``param.copy_``)
        refresh the packed copies IN PLACE; anything else (re-seated storage, a different set of tensors) drops them as before."""
        if self._packed_sig is None or self._engine is None:
            return super()._check_packed()
        ptrs = tuple(t.data_ptr() for t in self.parameters()) + tuple(t.data_ptr() for t in self.buffers())
        if self._packed_sig != self._signature():
            if getattr(self, "_packed_ptrs", None) == ptrs and self._engine.weight_store is None:
                with torch.no_grad():
                    self._engine.refresh()
                self._packed_fresh()
            else:
                self.invalidate_packed()

    def _packed_fresh(self):
        super()._packed_fresh()
        self._packed_ptrs = tuple(t.data_ptr() for t in self.parameters()) + tuple(t.data_ptr() for t in self.buffers())

    def engine(self, check=False) -> MoTEngine:
        """``check=True`` (the public entry points: once per prefill / generate_image / generate_text call) compares the
        parameters' (data_ptr, version) signature with the one the packed copies were built from."""
        if check:
            self._check_packed()
        if self._engine is None:
            self._engine = MoTEngine(self.model, self.lm_head, weight_store=getattr(self, "weight_store", None))
            self._packed_fresh()
        return self._engine

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    # ---- inference --------------------------------------------------------------------------------------
    def make_plan(self, query_lens, packed_query_position_ids, packed_query_indexes=None, key_values_lens=None,
                  packed_key_value_indexes=None, packed_vae_token_indexes=None, packed_text_indexes=None):
        return self.engine().plan(query_lens, packed_query_position_ids, packed_query_indexes=packed_query_indexes,
                                  key_values_lens=key_values_lens, packed_key_value_indexes=packed_key_value_indexes,
                                  text_indexes=packed_text_indexes, vae_indexes=packed_vae_token_indexes)

    @torch.no_grad()
    def forward_inference(self, packed_query_sequence, query_lens, packed_query_position_ids, packed_query_indexes,
                          past_key_values=None, key_values_lens=None, packed_key_value_indexes=None,
                          update_past_key_values=True, is_causal=True, mode="und", packed_vae_token_indexes=None,
                          packed_text_indexes=None, plan=None):
        # TaylorSeer (qwen2_navit.py:1034-1037): the caller parks the stream's state on the model, like the reference's
        # model.cache_dic / model.current (bagel.py:816-818)
        taylor = getattr(self.model, "current", None) if getattr(self.model, "enable_taylorseer", False) else None
        eng = self.engine(check=True)
        if plan is None:
            gen = mode == "gen" and eng.moe_mlp
            plan = self.make_plan(query_lens, packed_query_position_ids, packed_query_indexes, key_values_lens,
                                  packed_key_value_indexes, packed_vae_token_indexes if gen else None,
                                  packed_text_indexes if gen else None)
        seq = packed_query_sequence
        if seq.device != eng.device or seq.dtype != BF16:
            seq = seq.to(device=eng.device, dtype=BF16)
        out = eng.forward(seq, plan, mode, past_key_values, update_past_key_values, is_causal, taylor=taylor)
        return BaseNavitOutputWithPast(packed_query_sequence=out, past_key_values=past_key_values)

    @torch.no_grad()
    def forward_train(self, packed_sequence, sample_lens, attention_mask, packed_position_ids, packed_und_token_indexes=None,
                      packed_gen_token_indexes=None, split_lens=None, attn_modes=None, tape=None):
        """qwen2_navit.py:1124-1143 (forward only: no autograd graph is built).  ``attention_mask``: the list of per-sample
        additive masks of the non-flex path, or None with flat ``split_lens`` / ``attn_modes`` (the flex path's inputs)."""
        eng = self.engine(check=True)
        sample_lens = [int(x) for x in sample_lens]
        if isinstance(attention_mask, (list, tuple)):
            splits = [splits_from_mask(m) for m in attention_mask]
        elif split_lens is not None and attn_modes is not None:
            splits, i = [], 0
            for n in sample_lens:
                lens, modes, tot = [], [], 0
                while tot < n:
                    lens.append(int(split_lens[i])); modes.append(attn_modes[i]); tot += lens[-1]; i += 1
                splits.append((lens, modes))
        else:
            raise NotImplementedError("pass nested_attention_masks, or split_lens + attn_modes (a flex BlockMask object cannot "
                                      "be decoded: it hides the split structure it was built from)")
        if packed_gen_token_indexes is None:
            packed_gen_token_indexes = []
        tp = TrainPlan(eng.device, sample_lens, splits, packed_position_ids, packed_und_token_indexes, packed_gen_token_indexes,
                       self.model.rotary_emb.inv_freq(eng.device))
        seq = packed_sequence
        if seq.device != eng.device or seq.dtype != BF16:
            seq = seq.to(device=eng.device, dtype=BF16)
        return eng.forward_train(seq, tp, tape)

    def forward(self, *args, **kwargs):
        if self.training:
            return self.forward_train(*args, **kwargs)
        return self.forward_inference(*args, **kwargs)
