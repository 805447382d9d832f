"""Bagel: packs text / ViT / VAE-latent tokens into NaViT sequences and drives the MoT backbone, the rectified-flow
sampler (CFG + renorm) and the autoregressive text decode -- the reference's ``modeling.bagel.Bagel`` surface
(bagel.py:27-1074: same method names, keyword names, returned dict keys, state-dict keys), executed on MI355X
through the kernels in ``bagel_amd/csrc`` (no eager torch arithmetic, no CPU path).

Host/device split: every ``prepare_*`` packer runs on the host and returns the reference's CPU index tensors
(bit-exact contract); ``generate_image`` digests them ONCE into ForwardPlans, uploads the noise, and then the whole
T-1 step Euler loop only launches kernels (the reference re-derives lengths and syncs the host several times per
layer per step).
"""
import os

import numpy as np
import torch
from torch import nn

from ... import ops
from ...data.data_utils import (get_flattened_position_ids_extrapolate, get_flattened_position_ids_interpolate, patchify)
from .modeling_utils import MLPconnector, PositionEmbedding, TimestepEmbedder
from ..cache_utils.taylorseer import TaylorSeerState
from .qwen2_navit import Fp8DelayedScales, NaiveCache, _Linear, concat_plans

BF16 = torch.bfloat16


class BagelConfig:
    """Field names of bagel.py:27-54."""

    def __init__(self, visual_gen=True, visual_und=True, llm_config=None, vit_config=None, vae_config=None,
                 latent_patch_size=2, max_latent_size=32, vit_max_num_patch_per_side=70,
                 connector_act="gelu_pytorch_tanh", interpolate_pos=False, timestep_shift=1.0, **kwargs):
        self.visual_gen = visual_gen
        self.visual_und = visual_und
        self.llm_config = llm_config
        self.vit_config = vit_config
        self.vae_config = vae_config
        self.latent_patch_size = latent_patch_size
        self.max_latent_size = max_latent_size
        self.vit_max_num_patch_per_side = vit_max_num_patch_per_side
        self.connector_act = connector_act
        self.interpolate_pos = interpolate_pos
        self.timestep_shift = timestep_shift
        for k, v in kwargs.items():
            setattr(self, k, v)


# ------------------------------------------------------------------------------------------------------------
# packed-layout arithmetic shared by all packers
# ------------------------------------------------------------------------------------------------------------
def _merged_layout(ctx_lens, new_lens):
    """Row numbers inside the merged KV layout [ctx_0 | new_0 | ctx_1 | new_1 | ...].
    Returns (rows of the cached tokens, rows of the new tokens, start row of each sample's new block)."""
    ctx = np.asarray(ctx_lens, dtype=np.int64)
    new = np.asarray(new_lens, dtype=np.int64)
    block_start = np.concatenate([[0], np.cumsum(ctx + new)[:-1]]) if len(ctx) else np.zeros(0, np.int64)
    kv_rows = np.concatenate([np.arange(s, s + c) for s, c in zip(block_start, ctx)]) if len(ctx) else np.zeros(0, np.int64)
    new_start = block_start + ctx
    new_rows = np.concatenate([np.arange(s, s + n) for s, n in zip(new_start, new)]) if len(ctx) else np.zeros(0, np.int64)
    return kv_rows.astype(np.int64), new_rows.astype(np.int64), new_start


def _framed_image_layout(n_tokens):
    """Query-sequence rows when every sample is <start> n_b image tokens <end>:
    (rows of the two marker tokens, rows of the image tokens)."""
    n = np.asarray(n_tokens, dtype=np.int64)
    first = np.concatenate([[0], np.cumsum(n + 2)[:-1]])
    marker_rows = np.stack([first, first + n + 1], axis=1).reshape(-1)
    token_rows = np.concatenate([np.arange(f + 1, f + 1 + k) for f, k in zip(first, n)]) if len(n) else np.zeros(0, np.int64)
    return marker_rows, token_rows


def _bf16_weights(fn):
    """Entry points of the hot path: fp32 master weights are cast to bf16 once, in place (Bagel._ensure_bf16)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        self._ensure_bf16()
        return fn(self, *a, **k)
    return wrapped


def _lt(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.int64))


def _it(x):
    return torch.tensor([int(v) for v in x], dtype=torch.int)


class Bagel(nn.Module):
    """Drop-in for ``modeling.bagel.Bagel`` (inference methods)."""

    def __init__(self, language_model, vit_model, config: BagelConfig):
        super().__init__()
        self.language_model = language_model
        llm = config.llm_config
        self.hidden_size = llm.hidden_size
        self.use_moe = "Mo" in llm.layer_module
        self.num_heads = llm.num_attention_heads
        if config.visual_gen:
            self.latent_patch_size = config.latent_patch_size
            self.timestep_shift = config.timestep_shift
            self.latent_downsample = config.vae_config.downsample * config.latent_patch_size
            self.max_latent_size = config.max_latent_size
            self.latent_channel = config.vae_config.z_channels
            self.patch_latent_dim = self.latent_patch_size ** 2 * self.latent_channel
            self.time_embedder = TimestepEmbedder(self.hidden_size)
            self.vae2llm = _Linear(self.patch_latent_dim, self.hidden_size, True)
            self.llm2vae = _Linear(self.hidden_size, self.patch_latent_dim, True)
            self.latent_pos_embed = PositionEmbedding(self.max_latent_size, self.hidden_size)
        if config.visual_und:
            self.vit_model = vit_model
            self.vit_patch_size = config.vit_config.patch_size
            self.vit_max_num_patch_per_side = config.vit_max_num_patch_per_side
            self.vit_hidden_size = config.vit_config.hidden_size
            self.connector = MLPconnector(self.vit_hidden_size, self.hidden_size, config.connector_act)
            self.vit_pos_embed = PositionEmbedding(self.vit_max_num_patch_per_side, self.hidden_size)
        self.get_flattened_position_ids = (get_flattened_position_ids_interpolate if config.interpolate_pos
                                           else get_flattened_position_ids_extrapolate)
        self.config = config
        if config.visual_gen:   # bagel.py:96-99
            nn.init.constant_(self.llm2vae.weight, 0)
            nn.init.constant_(self.llm2vae.bias, 0)
        self._k64 = {}
        # execution options of generate_image.  cfg_batched (default since round 2: 417.2 -> 399.4 ms per Euler step at config 3,
        # profiles/r02_stream_batch.log): the cond + CFG forwards of a step run as ONE forward (see _stream_batch); und_side_path
        # sends the 2 marker rows per sample through the skinny GEMM beside it so the latent rows fill whole 256-row tiles.
        # Without the side path the result is bit-identical to sequential forwards; with it the marker rows see another
        # accumulation order (bf16 noise).  BAGEL_CFG_BATCH=0 / BAGEL_UND_SIDE=0 switch them off.
        self.cfg_batched = os.environ.get("BAGEL_CFG_BATCH", "1") == "1"
        self.und_side_path = os.environ.get("BAGEL_UND_SIDE", "1") == "1"
        self.step_hook = None          # optional callable(steps_taken, x_t) inside generate_image (trajectory tests / tooling)
        self.fp8_delayed_scaling = os.environ.get("BAGEL_FP8_DELAYED", "1") != "0"   # gen_weight_quant = "fp8": delayed row scales for the SwiGLU output (qwen2_navit.Fp8DelayedScales)
        self.velocity_hook = None      # optional callable(batched: bool, [v_cond, v_cfg_text | None, v_cfg_img | None]) inside every Euler step, BEFORE the CFG
        #                                combine: the per-stream velocities of the forward(s) the step just ran (LIVE bf16 buffers) -- parity gates of the timed path
        self.global_renorm_allreduce = False      # see _renorm_sums_allreduce
        # option (changes results; off by default, reported beside the bf16 numbers): "fp8" = the gen expert's four projections of the
        # denoise forwards run on the OCP-e4m3 MFMA with row-wise scales -- the MI355X counterpart of the reference's quantised load
        # modes (app.py:114-131).  BAGEL_GEN_QUANT=fp8 sets it for a whole process.
        self.gen_weight_quant = os.environ.get("BAGEL_GEN_QUANT") or None

    # ------------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.language_model.model.embed_tokens.weight.device

    def _dev(self, t, dtype=None):
        if t is None:
            return None
        if not torch.is_tensor(t):
            t = torch.tensor(t)
        return t.to(device=self.device, dtype=dtype if dtype is not None else t.dtype).contiguous()

    def state_dict(self, *args, **kwargs):
        if getattr(self, "_released_bf16", None):
            raise RuntimeError(f"state_dict(): quantize_language_model({self._released_bf16!r}, release_bf16=True) freed the bf16 projection weights of the language "
                               "model; a state dict written now would hold 0-element tensors.  Build a new model and load the checkpoint to get one.")
        return super().state_dict(*args, **kwargs)

    def _ensure_bf16(self):
        """The engines run bf16 weights (the app.py:111 / inferencer.py:233 configuration).  A caller that keeps fp32 master weights
        and relies on autocast -- eval/gen/gen_images_mp.py:174 does -- gets them cast ONCE, in place, with a warning: under the
        reference's autocast every matmul already rounds the weights to bf16 on the fly; what changes is that the residual stream
        and the norms are bf16 here (fp32 there)."""
        if self.language_model.model.embed_tokens.weight.dtype == torch.float32:
            import warnings
            warnings.warn("bagel_amd: fp32 weights cast to bfloat16 in place (the MI355X engines run bf16 weights and activations)")
            self.to(torch.bfloat16)

    def _embed_into(self, seq, token_ids, rows):
        """seq[rows] = embed_tokens[token_ids]  (bagel.py:277/377/508/796-798)."""
        table = self.language_model.model.embed_tokens.weight.data
        ids = self._dev(token_ids, torch.int32)
        ops.copy_rows(table, seq, ids.numel(), self.hidden_size, src_rows=ids, dst_rows=rows)

    def _timestep_embedding(self, t, keep=None):
        """time_embedder(t) for one scalar timestep -> (1, H) bf16  (modeling_utils.py:106-110).  The reference
        recomputes this MLP for every latent token although ``timestep.unique()`` is asserted to be a single value
        (bagel.py:800-802); one row is the same bits."""
        te, dev = self.time_embedder, self.device
        fe = te.frequency_embedding_size
        sinus = torch.empty((1, fe), dtype=BF16, device=dev)
        ops.timestep_sinusoid(float(t), te.freqs(dev), sinus)
        h = torch.empty((1, self.hidden_size), dtype=BF16, device=dev)
        ops.gemm(sinus, te.mlp[0].weight.data, h, bias0=te.mlp[0].bias.data, M0=1, epilogue=ops.EPI_SILU)
        out = torch.empty_like(h)
        ops.gemm(h, te.mlp[2].weight.data, out, bias0=te.mlp[2].bias.data, M0=1)
        if keep is not None:
            keep.append((sinus, h))        # the training backward needs the sinusoid row and the SiLU output
        return out

    def _latent_tokens_into(self, seq, latent_f32, vae_rows, vae_pos_ids, t):
        """seq[vae_rows] = bf16(bf16(vae2llm(x) + t_emb) + latent_pos_embed[ids])  (bagel.py:521-526, 801-806)."""
        x16 = ops.f32_to_bf16(latent_f32)
        ops.gemm(x16, self.vae2llm.weight.data, seq, bias0=self.vae2llm.bias.data, c_rows0=vae_rows, M0=x16.shape[0])
        ops.flow_add(seq, vae_rows, self._timestep_embedding(t), self.latent_pos_embed.pos_embed.data, vae_pos_ids)

    # ------------------------------------------------------------------------------------------------
    # host-side packers (integer outputs are bit-exact with the reference)
    # ------------------------------------------------------------------------------------------------
    def prepare_prompts(self, curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids):
        bos, eos = new_token_ids["bos_token_id"], new_token_ids["eos_token_id"]
        toks = [[bos] + list(tokenizer.encode(p)) + [eos] for p in prompts]
        lens = [len(t) for t in toks]
        kv_rows, new_rows, _ = _merged_layout(curr_kvlens, lens)
        pos = np.concatenate([np.arange(r, r + n) for r, n in zip(curr_rope, lens)]) if lens else np.zeros(0, np.int64)
        generation_input = {
            "text_token_lens": _it(lens),
            "packed_text_ids": torch.tensor([i for t in toks for i in t], dtype=torch.long),
            "packed_text_position_ids": _lt(pos),
            "packed_text_indexes": _lt(new_rows),
            "packed_key_value_indexes": _lt(kv_rows),
            "key_values_lens": _it(curr_kvlens),
        }
        return generation_input, [c + n for c, n in zip(curr_kvlens, lens)], [r + n for r, n in zip(curr_rope, lens)]

    def _image_block_inputs(self, curr_kvlens, curr_rope, n_tokens, new_token_ids):
        """Index tensors common to the three '<start> tokens <end>' packers."""
        qlens = [n + 2 for n in n_tokens]
        kv_rows, new_rows, _ = _merged_layout(curr_kvlens, qlens)
        marker_rows, token_rows = _framed_image_layout(n_tokens)
        B = len(n_tokens)
        return dict(
            text_ids=torch.tensor([new_token_ids["start_of_image"], new_token_ids["end_of_image"]] * B, dtype=torch.long)
            if new_token_ids is not None else None,
            text_rows=_lt(marker_rows), token_rows=_lt(token_rows),
            position_ids=_lt(np.repeat(np.asarray(curr_rope, dtype=np.int64), qlens)),
            seqlens=_it(qlens), indexes=_lt(new_rows), kv_indexes=_lt(kv_rows), kv_lens=_it(curr_kvlens))

    def prepare_vit_images(self, curr_kvlens, curr_rope, images, transforms, new_token_ids):
        tensors = [transforms(im) for im in images]
        tokens = [patchify(t, self.vit_patch_size) for t in tensors]
        n = [t.shape[0] for t in tokens]
        pos = [self.get_flattened_position_ids(t.size(1), t.size(2), self.vit_patch_size,
                                               max_num_patches_per_side=self.vit_max_num_patch_per_side) for t in tensors]
        L = self._image_block_inputs(curr_kvlens, curr_rope, n, new_token_ids)
        generation_input = {
            "packed_text_ids": L["text_ids"],
            "packed_text_indexes": L["text_rows"],
            "vit_token_seqlens": _it(n),
            "packed_vit_tokens": torch.cat(tokens, dim=0),
            "packed_vit_position_ids": torch.cat(pos, dim=0),
            "packed_vit_token_indexes": L["token_rows"],
            "packed_position_ids": L["position_ids"],
            "packed_seqlens": L["seqlens"],
            "packed_indexes": L["indexes"],
            "packed_key_value_indexes": L["kv_indexes"],
            "key_values_lens": L["kv_lens"],
        }
        return generation_input, [c + k + 2 for c, k in zip(curr_kvlens, n)], [r + 1 for r in curr_rope]

    def prepare_vae_images(self, curr_kvlens, curr_rope, images, transforms, new_token_ids, timestep=0):
        tensors = [transforms(im) for im in images]
        ds = self.latent_downsample
        shapes = [(t.shape[1] // ds, t.shape[2] // ds) for t in tensors]
        n = [h * w for h, w in shapes]
        pos = [self.get_flattened_position_ids(t.size(1), t.size(2), ds, max_num_patches_per_side=self.max_latent_size)
               for t in tensors]
        L = self._image_block_inputs(curr_kvlens, curr_rope, n, new_token_ids)
        Hm = max(t.shape[1] for t in tensors)
        Wm = max(t.shape[2] for t in tensors)
        Cm = max(t.shape[0] for t in tensors)
        padded = torch.zeros(size=(len(tensors), Cm, Hm, Wm), device=tensors[0].device)   # stays on the GPU when the transform does
        for i, t in enumerate(tensors):
            padded[i, :, : t.shape[1], : t.shape[2]] = t
        generation_input = {
            "padded_images": padded,
            "patchified_vae_latent_shapes": shapes,
            "packed_vae_position_ids": torch.cat(pos, dim=0),
            "packed_timesteps": torch.tensor([timestep]),
            "packed_vae_token_indexes": L["token_rows"],
            "packed_text_ids": L["text_ids"],
            "packed_text_indexes": L["text_rows"],
            "packed_position_ids": L["position_ids"],
            "packed_seqlens": L["seqlens"],
            "packed_indexes": L["indexes"],
            "packed_key_value_indexes": L["kv_indexes"],
            "key_values_lens": L["kv_lens"],
        }
        return generation_input, [c + k + 2 for c, k in zip(curr_kvlens, n)], [r + 1 for r in curr_rope]

    def prepare_vae_latent(self, curr_kvlens, curr_rope, image_sizes, new_token_ids):
        ds = self.latent_downsample
        n = [(H // ds) * (W // ds) for H, W in image_sizes]
        pos = [self.get_flattened_position_ids(H, W, ds, max_num_patches_per_side=self.max_latent_size) for H, W in image_sizes]
        # initial noise: CPU generator, one draw per image in sample order (bagel.py:578-580) -- RNG parity
        noises = [torch.randn(k, self.latent_channel * self.latent_patch_size ** 2) for k in n]
        L = self._image_block_inputs(curr_kvlens, curr_rope, n, new_token_ids)
        return {
            "packed_text_ids": L["text_ids"],
            "packed_text_indexes": L["text_rows"],
            "packed_init_noises": torch.cat(noises, dim=0),
            "packed_vae_position_ids": torch.cat(pos, dim=0),
            "packed_vae_token_indexes": L["token_rows"],
            "packed_seqlens": L["seqlens"],
            "packed_position_ids": L["position_ids"],
            "key_values_lens": L["kv_lens"],
            "packed_indexes": L["indexes"],
            "packed_key_value_indexes": L["kv_indexes"],
        }

    def prepare_vae_latent_cfg(self, curr_kvlens, curr_rope, image_sizes):
        ds = self.latent_downsample
        n = [(H // ds) * (W // ds) for H, W in image_sizes]
        L = self._image_block_inputs(curr_kvlens, curr_rope, n, None)
        return {
            "cfg_packed_position_ids": L["position_ids"],
            "cfg_key_values_lens": L["kv_lens"],
            "cfg_packed_query_indexes": L["indexes"],
            "cfg_packed_key_value_indexes": L["kv_indexes"],
        }

    def prepare_start_tokens(self, curr_kvlens, curr_rope, new_token_ids):
        kv_rows = np.arange(int(sum(curr_kvlens)), dtype=np.int64)   # bagel.py:913-918: no gaps yet
        return {
            "packed_start_tokens": torch.tensor([new_token_ids["bos_token_id"]] * len(curr_kvlens), dtype=torch.long),
            "packed_query_position_ids": torch.tensor(list(curr_rope), dtype=torch.long),
            "key_values_lens": _it(curr_kvlens),
            "packed_key_value_indexes": _lt(kv_rows),
        }

    # ------------------------------------------------------------------------------------------------
    # prefill
    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    @_bf16_weights
    def forward_cache_update_text(self, past_key_values, packed_text_ids, packed_text_position_ids, text_token_lens,
                                  packed_text_indexes, packed_key_value_indexes, key_values_lens):
        lm = self.language_model
        n = packed_text_ids.numel()
        seq = torch.empty((n, self.hidden_size), dtype=BF16, device=self.device)
        self._embed_into(seq, packed_text_ids, None)
        out = lm.forward_inference(
            packed_query_sequence=seq, query_lens=text_token_lens, packed_query_position_ids=packed_text_position_ids,
            packed_query_indexes=packed_text_indexes, past_key_values=past_key_values,
            packed_key_value_indexes=packed_key_value_indexes, key_values_lens=key_values_lens,
            update_past_key_values=True, is_causal=True, mode="und")
        return out.past_key_values

    @torch.no_grad()
    @_bf16_weights
    def forward_cache_update_vit(self, past_key_values, packed_text_ids, packed_text_indexes, packed_vit_tokens,
                                 packed_vit_token_indexes, packed_vit_position_ids, vit_token_seqlens, packed_position_ids,
                                 packed_seqlens, packed_indexes, packed_key_value_indexes, key_values_lens):
        dev = self.device
        total = int(sum(int(x) for x in packed_seqlens.tolist()))
        seq = torch.empty((total, self.hidden_size), dtype=BF16, device=dev)
        self._embed_into(seq, packed_text_ids, self._dev(packed_text_indexes, torch.int32))
        lens = [int(x) for x in vit_token_seqlens.tolist()]
        cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
        feats = self.vit_model(packed_pixel_values=packed_vit_tokens, packed_flattened_position_ids=packed_vit_position_ids,
                               cu_seqlens=cu, max_seqlen=max(lens))
        # connector (fc1 - gelu_tanh - fc2) + vit_pos_embed, scattered into the packed sequence (bagel.py:390-395)
        c = self.connector
        n = feats.shape[0]
        hmid = torch.empty((n, self.hidden_size), dtype=BF16, device=dev)
        ops.gemm(feats, c.fc1.weight.data, hmid, bias0=c.fc1.bias.data, epilogue=ops.EPI_GELU_TANH)
        emb = torch.empty_like(hmid)
        ops.gemm(hmid, c.fc2.weight.data, emb, bias0=c.fc2.bias.data)
        ops.add_table_rows(emb, self.vit_pos_embed.pos_embed.data, self._dev(packed_vit_position_ids, torch.long))
        ops.copy_rows(emb, seq, n, self.hidden_size, dst_rows=self._dev(packed_vit_token_indexes, torch.int32))
        out = self.language_model.forward_inference(
            packed_query_sequence=seq, query_lens=packed_seqlens, packed_query_position_ids=packed_position_ids,
            packed_query_indexes=packed_indexes, past_key_values=past_key_values,
            packed_key_value_indexes=packed_key_value_indexes, key_values_lens=key_values_lens,
            update_past_key_values=True, is_causal=False, mode="und")
        return out.past_key_values

    @torch.no_grad()
    @_bf16_weights
    def forward_cache_update_vae(self, vae_model, past_key_values, padded_images, patchified_vae_latent_shapes,
                                 packed_vae_position_ids, packed_timesteps, packed_vae_token_indexes, packed_text_ids,
                                 packed_text_indexes, packed_position_ids, packed_seqlens, packed_indexes, key_values_lens,
                                 packed_key_value_indexes):
        dev = self.device
        total = int(sum(int(x) for x in packed_seqlens.tolist()))
        seq = torch.empty((total, self.hidden_size), dtype=BF16, device=dev)
        self._embed_into(seq, packed_text_ids, self._dev(packed_text_indexes, torch.int32))
        padded_latent = vae_model.encode(padded_images)
        p, C = self.latent_patch_size, self.latent_channel
        pieces = []
        for latent, (h, w) in zip(padded_latent, patchified_vae_latent_shapes):
            lat = latent[:, : h * p, : w * p].reshape(C, h, p, w, p)
            pieces.append(lat.permute(1, 3, 2, 4, 0).reshape(h * w, p * p * C))   # "chpwq->hwpqc"
        packed_latent = torch.cat(pieces, dim=0).to(device=dev, dtype=torch.float32)
        ts = packed_timesteps.reshape(-1)
        if ts.numel() != 1:
            raise NotImplementedError("one shared timestep per call (bagel.py:477 always passes tensor([t]))")
        self._latent_tokens_into(seq, packed_latent, self._dev(packed_vae_token_indexes, torch.int32),
                                 self._dev(packed_vae_position_ids, torch.long), float(ts[0]))
        out = self.language_model.forward_inference(
            packed_query_sequence=seq, query_lens=packed_seqlens, packed_query_position_ids=packed_position_ids,
            packed_query_indexes=packed_indexes, past_key_values=past_key_values, key_values_lens=key_values_lens,
            packed_key_value_indexes=packed_key_value_indexes, update_past_key_values=True, is_causal=False,
            mode="gen", packed_vae_token_indexes=packed_vae_token_indexes, packed_text_indexes=packed_text_indexes)
        return out.past_key_values

    # ------------------------------------------------------------------------------------------------
    # rectified-flow sampler
    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def flow_schedule(num_timesteps, timestep_shift):
        """(timesteps[:-1], dts) exactly as bagel.py:693-696, evaluated on the host in fp32."""
        t = torch.linspace(1, 0, num_timesteps)
        t = timestep_shift * t / (1 + (timestep_shift - 1) * t)
        return t[:-1], t[:-1] - t[1:]

    @torch.no_grad()
    @_bf16_weights
    def generate_image(self, packed_text_ids, packed_text_indexes, packed_init_noises, packed_vae_position_ids,
                       packed_vae_token_indexes, packed_seqlens, packed_position_ids, packed_indexes, past_key_values,
                       key_values_lens, packed_key_value_indexes, num_timesteps=24, timestep_shift=1.0,
                       cfg_renorm_min=0.0, cfg_renorm_type="global", cfg_interval=[0, 1],
                       cfg_text_scale=1.0, cfg_text_packed_query_indexes=None, cfg_text_packed_position_ids=None,
                       cfg_text_past_key_values=None, cfg_text_key_values_lens=None, cfg_text_packed_key_value_indexes=None,
                       cfg_img_scale=1.0, cfg_img_packed_query_indexes=None, cfg_img_packed_position_ids=None,
                       cfg_img_past_key_values=None, cfg_img_key_values_lens=None, cfg_img_packed_key_value_indexes=None,
                       cfg_type="parallel", enable_taylorseer=False):
        # bagel.py:680-689: one TaylorSeer state per forward stream (cond, cfg-text, cfg-img)
        self.language_model.model.enable_taylorseer = False     # the engine gets the state explicitly (see _velocity)
        self.language_model.engine(check=True)                  # re-pack if the parameters changed since the last call
        taylor = [TaylorSeerState(num_timesteps) for _ in range(3)] if enable_taylorseer else [None, None, None]
        self._last_taylor_states = taylor
        # FP8 gen expert: one delayed-scale state per forward stream (sequential steps) + one for the stream-batched forward (its rows are all the streams')
        f8 = [Fp8DelayedScales() for _ in range(4)] if (self.gen_weight_quant == "fp8" and self.fp8_delayed_scaling) else [None] * 4
        if cfg_renorm_type not in ops.RENORM_MODES:
            raise NotImplementedError(f"{cfg_renorm_type} is not suppoprted")
        st = self._flow_state(packed_text_ids, packed_text_indexes, packed_vae_position_ids, packed_vae_token_indexes,
                              packed_seqlens)
        lm = self.language_model
        mk = lambda pos, qidx, kvlens, kvidx: lm.make_plan(  # noqa: E731
            packed_seqlens, pos, qidx, kvlens, kvidx, packed_vae_token_indexes, packed_text_indexes)
        plan = mk(packed_position_ids, packed_indexes, key_values_lens, packed_key_value_indexes)
        plan_t = plan_i = None
        if cfg_text_scale > 1.0:
            plan_t = mk(cfg_text_packed_position_ids, cfg_text_packed_query_indexes, cfg_text_key_values_lens,
                        cfg_text_packed_key_value_indexes)
            if cfg_img_scale > 1.0:
                # the reference also runs this forward when cfg_text_scale <= 1 but discards it (bagel.py:854-905)
                plan_i = mk(cfg_img_packed_position_ids, cfg_img_packed_query_indexes, cfg_img_key_values_lens,
                            cfg_img_packed_key_value_indexes)
        x_t = packed_init_noises.to(device=self.device, dtype=torch.float32).contiguous().clone()
        timesteps, dts = self.flow_schedule(num_timesteps, timestep_shift)
        mode = ops.RENORM_MODES[cfg_renorm_type]
        multi = None
        if self.cfg_batched and plan_t is not None:
            # the batched forward reads ONE concatenated context: a second resident copy of every stream's K / V rows for the duration of
            # this call (the image-edit request: 3 streams x ~9 k tokens x 28 layers x 2 KB = 1.6 GB per sample).  If that does not fit,
            # the step falls back to the sequential forwards (same results, no copy) instead of failing the request.
            try:
                multi = self._stream_batch(st, [plan, plan_t] + ([plan_i] if plan_i is not None else []),
                                           [past_key_values, cfg_text_past_key_values] + ([cfg_img_past_key_values] if plan_i is not None else []))
            except torch.cuda.OutOfMemoryError:
                import warnings
                warnings.warn("bagel_amd: no memory for the stream-batched CFG context copy; running the conditional and CFG forwards one after the other")
                torch.cuda.empty_cache()
                multi = None
        for i, t in enumerate(timesteps):
            use_cfg = bool(t > cfg_interval[0] and t <= cfg_interval[1])   # fp32 tensor vs python float, as bagel.py:701
            s_t = cfg_text_scale if use_cfg else 1.0
            s_i = cfg_img_scale if use_cfg else 1.0
            self._flow_step(st, x_t, float(t), float(dts[i]), plan, past_key_values,
                            plan_t if s_t > 1.0 else None, cfg_text_past_key_values,
                            plan_i if (s_t > 1.0 and s_i > 1.0) else None, cfg_img_past_key_values,
                            s_t, s_i, cfg_renorm_min, mode, taylor, multi, f8)
            if self.step_hook is not None:     # test / tooling aid (trajectory parity): called with (Euler steps taken, x_t) -- x_t is the LIVE buffer
                self.step_hook(i + 1, x_t)
        return x_t.split([int(n) - 2 for n in packed_seqlens.tolist()])

    def _renorm_sums_allreduce(self, partials, nparts, mode):
        """Optional batch-global 'global' renorm across data-parallel ranks: the reference's norm spans the tokens
        of the LOCAL pack only (bagel.py:892-895), which is also the default here; with ``model.global_renorm_allreduce = True``
        the per-block partial sums of stage 1 are summed over the ranks (one small all-reduce per step) so that an N-rank batch
        renormalises like the same batch in one process.  The collective always carries the WHOLE fixed-size partials buffer
        (2 x 256 fp32, the unused tail zeroed): ``nparts`` depends on the local row count, and ranks with different pack sizes
        must not enter an all-reduce with different tensor sizes.  Returns the number of partial pairs stage 2 has to sum.
        Every rank of the group has to run the same number of denoise steps (a rank with an empty shard must not skip
        generate_image while the others wait in the collective)."""
        if mode != 0 or not self.global_renorm_allreduce:
            return nparts
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from ...parallel import allreduce_renorm_sums
            full = partials.numel() // 2
            if nparts < full:
                partials[2 * nparts:].zero_()
            allreduce_renorm_sums(partials)
            return full
        return nparts

    def _stream_batch(self, st, plans, caches):
        """``model.cfg_batched`` (default on; BAGEL_CFG_BATCH=0 turns it off): the conditional and the CFG forwards of a denoise step share the
        query sequence and differ only in position ids and context (bagel.py:820-870), so they run as ONE forward over
        [stream 0's samples | stream 1's | ...] -- row-wise operators and per-sample attention make every row's arithmetic the
        same as in separate forwards.  What it buys is tile quantisation: 4 x 4096 latent rows per stream are 64 row tiles of
        256, i.e. 3.5 / 4.5 / 37 rounds of the 256 persistent GEMM workgroups per projection (paid as 4 / 5 / 38); two streams
        are exactly 7 / 9 / 74 once the 2-per-sample marker rows take the dense side path (ForwardPlan.und_side)."""
        S, M = len(plans), plans[0].M
        pm = concat_plans(plans)
        pm.und_side = self.und_side_path
        cm = NaiveCache.concat(caches, [p.B for p in plans])
        return dict(plan=pm, cache=cm, S=S, M=M,
                    seq=torch.empty((S * M, self.hidden_size), dtype=BF16, device=self.device),
                    vae_rows=[(st["vae_rows"] + s * M).contiguous() for s in range(S)])

    def _flow_state(self, packed_text_ids, packed_text_indexes, packed_vae_position_ids, packed_vae_token_indexes,
                    packed_seqlens):
        dev = self.device
        total = int(sum(int(x) for x in packed_seqlens.tolist()))
        nv = packed_vae_token_indexes.numel()
        return dict(
            seq=torch.empty((total, self.hidden_size), dtype=BF16, device=dev),
            text_ids=packed_text_ids, text_rows=self._dev(packed_text_indexes, torch.int32),
            vae_rows=self._dev(packed_vae_token_indexes, torch.int32),
            vae_pos=self._dev(packed_vae_position_ids, torch.long),
            v=[torch.empty((nv, self.patch_latent_dim), dtype=BF16, device=dev) for _ in range(3)],
            tmp=torch.empty((nv, self.patch_latent_dim), dtype=BF16, device=dev),
            partials=torch.empty((2 * 256,), dtype=torch.float32, device=dev), embedded=False)

    def _velocity(self, st, plan, cache, out, taylor=None, fp8_state=None):
        """llm2vae(backbone(seq))[latent rows] -> out (bagel.py:820-833)."""
        h = self.language_model.engine().forward(st["seq"], plan, "gen" if self.use_moe else "und", cache, update=False,
                                                 causal=False, taylor=taylor, gen_quant=self.gen_weight_quant, fp8_state=fp8_state)
        ops.gemm(h, self.llm2vae.weight.data, out, bias0=self.llm2vae.bias.data, a_rows0=st["vae_rows"], M0=out.shape[0])
        return out

    def _flow_step(self, st, x_t, t, dt, plan, cache, plan_t, cache_t, plan_i, cache_i, s_t, s_i, renorm_min, mode,
                   taylor=(None, None, None), multi=None, f8=(None, None, None, None)):
        """One Euler step of bagel.py:698-746 (= _forward_flow + the update), all on the current stream."""
        seq = st["seq"]
        if not st["embedded"]:      # marker-token rows never change across steps
            self._embed_into(seq, st["text_ids"], st["text_rows"])
            st["embedded"] = True
        self._latent_tokens_into(seq, x_t, st["vae_rows"], st["vae_pos"], t)
        if multi is not None and plan_t is not None and multi["S"] == 2 + (plan_i is not None):
            # every stream of this step in one forward (see _stream_batch)
            S, M = multi["S"], multi["M"]
            for s in range(S):
                multi["seq"][s * M:(s + 1) * M].copy_(seq)
            h = self.language_model.engine().forward(multi["seq"], multi["plan"], "gen" if self.use_moe else "und", multi["cache"],
                                                     update=False, causal=False, taylor=list(taylor[:S]) if taylor[0] is not None else None,
                                                     gen_quant=self.gen_weight_quant, fp8_state=f8[3])
            for s in range(S):
                ops.gemm(h, self.llm2vae.weight.data, st["v"][s], bias0=self.llm2vae.bias.data, a_rows0=multi["vae_rows"][s],
                         M0=st["v"][s].shape[0])
            v, v_ct, v_ci = st["v"][0], st["v"][1], (st["v"][2] if plan_i is not None else None)
            if self.velocity_hook is not None:
                self.velocity_hook(True, [v, v_ct, v_ci])
            nparts = ops.cfg_stage1(v, v_ct, v_ci, st["tmp"], st["partials"], s_t, s_i, renorm_min, mode)
            nparts = self._renorm_sums_allreduce(st["partials"], nparts, mode)
            ops.cfg_stage2_euler(x_t, st["tmp"], st["partials"], nparts, renorm_min, dt, use_global_scale=(mode == 0))
            return
        v = self._velocity(st, plan, cache, st["v"][0], taylor[0], f8[0])
        if plan_t is not None:
            v_ct = self._velocity(st, plan_t, cache_t, st["v"][1], taylor[1], f8[1])
            v_ci = self._velocity(st, plan_i, cache_i, st["v"][2], taylor[2], f8[2]) if plan_i is not None else None
            if self.velocity_hook is not None:
                self.velocity_hook(False, [v, v_ct, v_ci])
            nparts = ops.cfg_stage1(v, v_ct, v_ci, st["tmp"], st["partials"], s_t, s_i, renorm_min, mode)
            nparts = self._renorm_sums_allreduce(st["partials"], nparts, mode)
            ops.cfg_stage2_euler(x_t, st["tmp"], st["partials"], nparts, renorm_min, dt, use_global_scale=(mode == 0))
        else:
            ops.cfg_stage2_euler(x_t, v, None, 0, renorm_min, dt, use_global_scale=False)

    @torch.no_grad()
    @_bf16_weights
    def _forward_flow(self, x_t, timestep, packed_vae_token_indexes, packed_vae_position_ids, packed_text_ids,
                      packed_text_indexes, packed_indexes, packed_position_ids, packed_seqlens, key_values_lens,
                      past_key_values, packed_key_value_indexes, cfg_renorm_min=0.0, cfg_renorm_type="global",
                      cfg_text_scale=1.0, cfg_text_packed_position_ids=None, cfg_text_packed_query_indexes=None,
                      cfg_text_key_values_lens=None, cfg_text_past_key_values=None, cfg_text_packed_key_value_indexes=None,
                      cfg_img_scale=1.0, cfg_img_packed_position_ids=None, cfg_img_packed_query_indexes=None,
                      cfg_img_key_values_lens=None, cfg_img_past_key_values=None, cfg_img_packed_key_value_indexes=None,
                      cfg_type="parallel", **taylorseer_kwargs):
        """Velocity for one timestep (bagel.py:757-907), kept for callers that drive the sampler themselves.
        Returns v_t (bf16, on the GPU); does not touch x_t."""
        tvals = timestep.reshape(-1)
        if tvals.unique().shape[0] != 1:
            raise AssertionError("all latent tokens must share one timestep")   # bagel.py:800
        st = self._flow_state(packed_text_ids, packed_text_indexes, packed_vae_position_ids, packed_vae_token_indexes,
                              packed_seqlens)
        lm = self.language_model
        mk = lambda pos, qidx, kvlens, kvidx: lm.make_plan(  # noqa: E731
            packed_seqlens, pos, qidx, kvlens, kvidx, packed_vae_token_indexes, packed_text_indexes)
        x_dev = x_t.to(device=self.device, dtype=torch.float32).contiguous()
        self._embed_into(st["seq"], st["text_ids"], st["text_rows"])
        self._latent_tokens_into(st["seq"], x_dev, st["vae_rows"], st["vae_pos"], float(tvals[0]))
        v = self._velocity(st, mk(packed_position_ids, packed_indexes, key_values_lens, packed_key_value_indexes),
                           past_key_values, st["v"][0])
        if cfg_text_scale > 1.0:
            v_ct = self._velocity(st, mk(cfg_text_packed_position_ids, cfg_text_packed_query_indexes, cfg_text_key_values_lens,
                                         cfg_text_packed_key_value_indexes), cfg_text_past_key_values, st["v"][1])
            v_ci = None
            if cfg_img_scale > 1.0:
                v_ci = self._velocity(st, mk(cfg_img_packed_position_ids, cfg_img_packed_query_indexes, cfg_img_key_values_lens,
                                             cfg_img_packed_key_value_indexes), cfg_img_past_key_values, st["v"][2])
            mode = ops.RENORM_MODES[cfg_renorm_type]
            nparts = ops.cfg_stage1(v, v_ct, v_ci, st["tmp"], st["partials"], cfg_text_scale, cfg_img_scale, cfg_renorm_min, mode)
            if mode == 0:   # apply the global scale without the Euler update: x = 0 - bf16(v*1) trick is not exact; do it directly
                zero = torch.zeros(v.shape, dtype=torch.float32, device=self.device)
                ops.cfg_stage2_euler(zero, st["tmp"], st["partials"], nparts, cfg_renorm_min, -1.0, use_global_scale=True)
                return zero.to(BF16)
            return st["tmp"].clone()
        return v.clone()

    # ------------------------------------------------------------------------------------------------
    # autoregressive text
    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    @_bf16_weights
    def quantize_language_model(self, kind="nf4", release_bf16=True):
        """The reference's quantised LOAD MODES (app.py:114-131: bitsandbytes NF4 resp. INT8 over every nn.Linear of the language model) for the whole
        forward path -- prefill, the denoise loop and text decode.  The decoder layers' projections are kept as NF4 codes + fp32 block absmax ("nf4":
        blocks of 64, no double quantisation) or row-wise absmax INT8 ("int8_rowwise"); a layer's bf16 matrices are materialised right before its GEMMs
        (``w = bf16(code_book[code] * absmax)``: bitsandbytes' matmul_4bit in front of F.linear) and the Lq = 1 decode streams the codes themselves.
        Embeddings, lm_head, norms, biases, ViT / connector / VAE stay bf16 like the library leaves non-Linear and skipped modules.
        ``release_bf16=True`` frees the bf16 projection weights afterwards (28.3 -> 8.6 GB at 7B with "nf4"); reload the checkpoint to go back.
        An option that changes results, like the reference's modes; parity with bitsandbytes' binaries is unpinned (oracle/nf4.py).
        "int8" -- the reference's mode 3, LLM.int8 with fp16 outlier columns (app.py:126-131) -- is NOT built and is refused by name.
        With ``release_bf16=True`` the model can no longer produce a state dict (``state_dict()`` raises: the projection weights are gone) and any later
        invalidation of the packed weights (``.to()``, a second call) needs a NEW model build + checkpoint load."""
        from .qwen2_navit import LLM_INT8_NOT_BUILT
        if kind == "int8":
            raise NotImplementedError(LLM_INT8_NOT_BUILT)
        if kind not in ("nf4", "int8_rowwise"):
            raise NotImplementedError(f"quantize_language_model(kind={kind!r}): 'nf4' and 'int8_rowwise' are built")
        lm = self.language_model
        self._ensure_bf16()
        if hasattr(lm.model, "release_train_buffers"):
            lm.model.release_train_buffers()          # (tape pool of a previous training step: tens of GB at 7B, train_step.py)
        lm.weight_store = kind
        lm.invalidate_packed()
        eng = lm.engine()
        if release_bf16:
            for L in lm.model.layers:
                mods = [getattr(L.self_attn, n + s) for n in ("q_proj", "k_proj", "v_proj", "o_proj") for s in (("", "_moe_gen") if eng.mot else ("",))]
                for s in (("", "_moe_gen") if eng.moe_mlp else ("",)):
                    m = getattr(L, "mlp" + s)
                    mods += [m.gate_proj, m.up_proj, m.down_proj]
                for m in mods:
                    m.weight.data = torch.empty(0, dtype=m.weight.dtype, device=m.weight.device)
            lm._packed_fresh()
            self._released_bf16 = kind
        return eng.layers.resident_bytes()

    @torch.no_grad()
    @_bf16_weights
    def generate_text(self, past_key_values, packed_key_value_indexes, key_values_lens, packed_start_tokens,
                      packed_query_position_ids, max_length, do_sample=False, temperature=1.0, end_token_id=None,
                      use_graph=None, weight_quant=None):
        """bagel.py:930-1000.  Returns the INPUT token of every step, shape (steps, B) int64 (first row = bos).

        Execution: a ``DecodeSession`` (decode.py) -- paged KV cache adopted from ``past_key_values``, loop state on the
        device, step 0 launched eagerly, the remaining steps replayed from one captured hipGraph (``use_graph=False``
        or BAGEL_DECODE_GRAPH=0 keeps every step eager).  ``past_key_values`` receives the new K/V rows at the end, as
        the reference's in-place cache update does.  ``weight_quant="int8_rowwise"`` (or ``model.decode_weight_quant``) streams row-wise
        INT8 copies of the layer weights instead of bf16 -- an option that changes results, like the reference's quantised
        load modes; default off."""
        import os
        from .decode import DecodeSession
        lm = self.language_model
        kv_lens = [int(x) for x in key_values_lens.tolist()]
        if packed_key_value_indexes is not None and int(packed_key_value_indexes.numel()) != sum(kv_lens):
            raise ValueError("packed_key_value_indexes does not cover key_values_lens")
        if max_length <= 0:
            return torch.empty((0, len(kv_lens)), dtype=torch.long, device=self.device)
        if weight_quant is None:
            weight_quant = getattr(self, "decode_weight_quant", None)    # model-level switch, like the reference's load-time modes (app.py:114-131)
        self._last_decode_session = None      # release the previous call's page pools BEFORE this call allocates its own (16 requests x 5 k tokens: 9 GB;
        #                                       holding both made every call pay a fresh hipMalloc of that size, ~100 ms)
        # do_sample: the draw happens ON THE DEVICE inside the step (Gumbel-max over Philox numbers: the same categorical distribution as the reference's
        # multinomial(softmax(logits / temperature)), bagel.py:980-983), seeded once per call from torch's generator -- torch.manual_seed makes a run
        # reproducible, the decode replays from the hipGraph.  BAGEL_DECODE_SAMPLER=torch keeps torch.multinomial on the logits of every (eager) step.
        device_sampler = bool(do_sample) and os.environ.get("BAGEL_DECODE_SAMPLER", "device") != "torch"
        gumbel = ("gumbel", float(temperature), int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())) if device_sampler else None
        sess = DecodeSession(lm.engine(check=True), lm.model.embed_tokens.weight.data, lm.lm_head.weight.data, past_key_values, kv_lens,
                             packed_start_tokens, packed_query_position_ids, max_length, weight_quant=weight_quant, sampler=gumbel)
        self._last_decode_session = sess
        if use_graph is None:
            use_graph = os.environ.get("BAGEL_DECODE_GRAPH", "1") != "0"
        sampler = None
        if do_sample and not device_sampler:
            # sampling draws from torch's generator (an RNG stream cannot be matched across devices anyway, bagel.py:980-983)
            sampler = lambda logits: torch.multinomial(torch.softmax(logits.float() / temperature, dim=-1), num_samples=1).squeeze(1)  # noqa: E731
        step = 0
        while step < max_length:
            if step == 1 and use_graph and max_length > 2:
                sess.capture(include_advance=sampler is None)
            sess.step(sampler)
            step += 1
            if end_token_id is not None and sess.last_token(0) == end_token_id:   # only support batch=1 (bagel.py:996)
                break
        sess.write_back(past_key_values)
        sess.check_engine_status()          # (the persistent decode engine reports a spin that gave up through a device word; one sync here)
        return sess.tokens_so_far()

    @torch.no_grad()
    def chat(self, tokenizer, new_token_ids, image_transform, images, prompt, max_length, do_sample=False, temperature=1.0):
        """Evaluation entry point (bagel.py:1004-1074): ViT prefill per image, text prefill, greedy/sampled decode."""
        cache = NaiveCache(self.config.llm_config.num_hidden_layers)
        newlens, new_rope = [0], [0]
        for image in images:
            gi, newlens, new_rope = self.prepare_vit_images(newlens, new_rope, [image], image_transform, new_token_ids)
            cache = self.forward_cache_update_vit(cache, **gi)
        gi, newlens, new_rope = self.prepare_prompts(newlens, new_rope, [prompt], tokenizer, new_token_ids)
        cache = self.forward_cache_update_text(cache, **gi)
        gi = self.prepare_start_tokens(newlens, new_rope, new_token_ids)
        toks = self.generate_text(past_key_values=cache, max_length=max_length, do_sample=do_sample, temperature=temperature,
                                  end_token_id=new_token_ids["eos_token_id"], **gi)
        output = tokenizer.decode(toks[:, 0])
        return output.split("<|im_end|>")[0].split("<|im_start|>")[1]

    # ------------------------------------------------------------------------------------------------
    # training step: Bagel.forward (per-token losses) and its backward (train_step.PackedTrainStep)
    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _rows(index_or_mask):
        t = torch.as_tensor(index_or_mask)
        return torch.nonzero(t, as_tuple=False).flatten() if t.dtype == torch.bool else t.to(torch.long)

    _TRAINABLE_PREFIXES = ("language_model.", "llm2vae.", "vae2llm.", "time_embedder.", "connector.", "vit_model.")

    def forward(self, sequence_length, packed_text_ids, packed_text_indexes, sample_lens, packed_position_ids,
                nested_attention_masks=None, split_lens=None, attn_modes=None, ce_loss_indexes=None, packed_label_ids=None,
                packed_vit_tokens=None, packed_vit_token_indexes=None, packed_vit_position_ids=None, vit_token_seqlens=None,
                padded_latent=None, patchified_vae_latent_shapes=None, packed_latent_position_ids=None,
                packed_vae_token_indexes=None, packed_timesteps=None, mse_loss_indexes=None, noise=None):
        """Bagel.forward (bagel.py:101-229): per-token losses ``dict(mse=[n_mse, 64] fp32, ce=[n_ce] fp32)`` of a packed
        training batch.

        Same arguments as the reference plus ``noise`` (the ``randn_like`` draw of :184; drawn on the GPU when omitted).
        The block mask comes as the per-sample additive masks of the non-flex path (decoded back into splits and checked)
        or as flat ``split_lens`` + ``attn_modes``; it runs as per-split sequences of the varlen attention kernel.

        With grad mode on and at least one parameter that requires grad the losses are attached to ONE autograd node
        (train_step.PackedTrainStep), so the reference's ``loss.backward()`` (train/pretrain_unified_navit.py:683-735) fills
        ``param.grad`` from the hand-written reverse kernels; otherwise (``torch.no_grad()``, or a frozen model) no tape is kept."""
        kw = dict(sequence_length=sequence_length, packed_text_ids=packed_text_ids, packed_text_indexes=packed_text_indexes,
                  sample_lens=sample_lens, packed_position_ids=packed_position_ids, nested_attention_masks=nested_attention_masks,
                  split_lens=split_lens, attn_modes=attn_modes, ce_loss_indexes=ce_loss_indexes, packed_label_ids=packed_label_ids,
                  packed_vit_tokens=packed_vit_tokens, packed_vit_token_indexes=packed_vit_token_indexes,
                  packed_vit_position_ids=packed_vit_position_ids, vit_token_seqlens=vit_token_seqlens, padded_latent=padded_latent,
                  patchified_vae_latent_shapes=patchified_vae_latent_shapes, packed_latent_position_ids=packed_latent_position_ids,
                  packed_vae_token_indexes=packed_vae_token_indexes, packed_timesteps=packed_timesteps, mse_loss_indexes=mse_loss_indexes,
                  noise=noise)
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad] if torch.is_grad_enabled() else []
        fp32 = [n for n, p in named if p.dtype == torch.float32]
        if fp32:
            # The inference entry points cast fp32 weights to bf16 in place; doing that to TRAINABLE parameters would destroy the caller's
            # master copy and leave the optimizer updating bf16 values (updates below bf16 resolution are rounded away) -- the reference
            # trains fp32 masters with bf16 compute (FSDP MixedPrecision, train/fsdp_utils.py).  Refuse instead of changing the recipe.
            raise TypeError(
                f"Bagel.forward with grad: {len(fp32)} trainable parameter(s) are fp32 (first: {fp32[0]}).  The MI355X engines compute on bf16 "
                "parameters; keep the fp32 MASTER copies in the optimizer: model.to(torch.bfloat16), then wrap the optimizer with "
                "bagel_amd.train_utils.MasterWeightOptimizer(model, lambda ps: torch.optim.AdamW(ps, ...)) -- fp32 masters and optimizer "
                "state, bf16 compute copies refreshed in place after every step.")
        self._ensure_bf16()
        if not named:
            with torch.no_grad():
                return self._forward_losses(tape=None, **kw)
        bad = [n for n, _ in named if not n.startswith(self._TRAINABLE_PREFIXES)]
        if bad:
            raise NotImplementedError(
                "the backward is built for the language model, llm2vae / vae2llm, the time embedder, the connector and the SigLIP tower; "
                f"freeze the rest (the sin-cos position tables are frozen in the reference too): {bad[:4]}")
        from .train_step import PackedTrainStep
        mse, ce = PackedTrainStep.apply(self, kw, *[p for _, p in named])
        has_mse = bool(self.config.visual_gen) and padded_latent is not None and mse_loss_indexes is not None
        return dict(mse=mse if has_mse else None, ce=ce if ce_loss_indexes is not None else None)

    def _forward_losses(self, tape, sequence_length, packed_text_ids, packed_text_indexes, sample_lens, packed_position_ids,
                        nested_attention_masks=None, split_lens=None, attn_modes=None, ce_loss_indexes=None, packed_label_ids=None,
                        packed_vit_tokens=None, packed_vit_token_indexes=None, packed_vit_position_ids=None, vit_token_seqlens=None,
                        padded_latent=None, patchified_vae_latent_shapes=None, packed_latent_position_ids=None,
                        packed_vae_token_indexes=None, packed_timesteps=None, mse_loss_indexes=None, noise=None):
        """The forward itself; ``tape`` (train_step.TrainTape or None) collects what the backward needs."""
        F_ = tape.front if tape is not None else None
        dev = self.device
        H = self.hidden_size
        total = int(sequence_length)
        if total != int(sum(int(x) for x in sample_lens)):
            raise NotImplementedError("sequence_length must equal sum(sample_lens) (the packer's pad split is part of the last sample)")
        seq = torch.zeros((total, H), dtype=BF16, device=dev)
        text_rows = self._rows(packed_text_indexes)
        self._embed_into(seq, packed_text_ids, self._dev(text_rows, torch.int32))
        if F_ is not None:
            F_.update(M=total, text_rows=text_rows.cpu(), text_ids=torch.as_tensor(packed_text_ids).cpu().to(torch.long))
        und_rows = text_rows
        if self.config.visual_und and packed_vit_tokens is not None:
            lens = [int(x) for x in torch.as_tensor(vit_token_seqlens).tolist()]
            cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
            vit_tape = {} if (F_ is not None and any(p.requires_grad for p in self.vit_model.parameters())) else None
            feats = self.vit_model(packed_pixel_values=packed_vit_tokens, packed_flattened_position_ids=packed_vit_position_ids,
                                   cu_seqlens=cu, max_seqlen=max(lens), **({"tape": vit_tape} if vit_tape is not None else {}))
            c = self.connector
            n = feats.shape[0]
            hmid = torch.empty((n, H), dtype=BF16, device=dev)
            ops.gemm(feats, c.fc1.weight.data, hmid, bias0=c.fc1.bias.data, epilogue=ops.EPI_GELU_TANH)
            emb = torch.empty_like(hmid)
            ops.gemm(hmid, c.fc2.weight.data, emb, bias0=c.fc2.bias.data)
            ops.add_table_rows(emb, self.vit_pos_embed.pos_embed.data, self._dev(packed_vit_position_ids, torch.long))
            vit_rows = self._rows(packed_vit_token_indexes)
            ops.copy_rows(emb, seq, n, H, dst_rows=self._dev(vit_rows, torch.int32))
            und_rows = torch.cat([text_rows, vit_rows], dim=0)
            if F_ is not None:
                F_.update(vit_feats=feats, vit_hmid=hmid, vit_rows=self._dev(vit_rows, torch.int32), vit_tape=vit_tape)
        gen_rows = None
        if self.config.visual_gen and padded_latent is not None:
            p, C = self.latent_patch_size, self.latent_channel
            pieces = []
            for latent, (h, w) in zip(padded_latent, patchified_vae_latent_shapes):
                lat = latent[:, : h * p, : w * p].reshape(C, h, p, w, p)
                pieces.append(lat.permute(1, 3, 2, 4, 0).reshape(h * w, p * p * C))      # "chpwq->hwpqc" (:180)
            clean = torch.cat(pieces, dim=0).to(device=dev, dtype=torch.float32).contiguous()
            if noise is None:
                noise = torch.randn(clean.shape, dtype=torch.float32, device=dev)
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            # timestep warp on the host (a few floats; the same fp32 ops the reference runs): sigmoid, then the shift
            t = torch.sigmoid(torch.as_tensor(packed_timesteps, dtype=torch.float32).cpu())
            t = self.timestep_shift * t / (1 + (self.timestep_shift - 1) * t)
            t_dev = t.to(dev).contiguous()
            gen_rows = self._rows(packed_vae_token_indexes)
            vae_rows = self._dev(gen_rows, torch.int32)
            x16 = ops.flow_mix(clean, noise, t_dev)                                       # (1 - t) x0 + t eps, cast for vae2llm
            ops.gemm(x16, self.vae2llm.weight.data, seq, bias0=self.vae2llm.bias.data, c_rows0=vae_rows, M0=x16.shape[0])
            uniq, inv = torch.unique(t, return_inverse=True)
            keep = [] if F_ is not None else None
            temb = torch.cat([self._timestep_embedding(float(u), keep) for u in uniq], dim=0)   # one time-MLP pass per distinct t
            ops.flow_add_rows(seq, vae_rows, temb, inv.to(device=dev, dtype=torch.int32), self.latent_pos_embed.pos_embed.data,
                              self._dev(packed_latent_position_ids, torch.long))
            if F_ is not None:
                F_.update(x16=x16, vae_rows=vae_rows, temb_inv=inv.cpu(), time_sinus=torch.cat([a for a, _ in keep], 0),
                          time_h=torch.cat([b for _, b in keep], 0))
        last = self.language_model.forward_train(
            packed_sequence=seq, sample_lens=sample_lens, attention_mask=nested_attention_masks,
            packed_position_ids=packed_position_ids, packed_und_token_indexes=und_rows, packed_gen_token_indexes=gen_rows,
            split_lens=split_lens, attn_modes=attn_modes, tape=tape)
        if F_ is not None:
            F_["last"] = last
        mse = None
        if gen_rows is not None and mse_loss_indexes is not None:
            mrows = self._rows(mse_loss_indexes)
            mrows_d = self._dev(mrows, torch.int32)
            preds = torch.empty((mrows.numel(), self.patch_latent_dim), dtype=BF16, device=dev)
            ops.gemm(last, self.llm2vae.weight.data, preds, bias0=self.llm2vae.bias.data, a_rows0=mrows_d, M0=mrows.numel())
            src = torch.nonzero(t > 0, as_tuple=False).flatten()                         # has_mse (:215)
            if src.numel() != mrows.numel():
                raise ValueError("mse_loss_indexes must address exactly the latent tokens with timestep > 0")
            src_d = src.to(device=dev, dtype=torch.int32)
            mse = ops.mse_rows(preds, noise, clean, src_d)
            if F_ is not None:
                F_.update(mse_rows=mrows_d, preds=preds, noise=noise, clean=clean, mse_src=src_d)
        ce = None
        if ce_loss_indexes is not None:
            crows = self._rows(ce_loss_indexes)
            crows_d = self._dev(crows, torch.int32)
            head = self.language_model.lm_head.weight.data
            logits = torch.empty((crows.numel(), head.shape[0]), dtype=BF16, device=dev)
            ops.gemm(last, head, logits, a_rows0=crows_d, M0=crows.numel())
            labels = self._dev(packed_label_ids, torch.long)
            ce = ops.cross_entropy(logits, labels)
            if F_ is not None:
                F_.update(ce_rows=crows_d, logits=logits, labels=labels)
        return dict(mse=mse, ce=ce)

    def _backward_losses(self, tape, d_mse, d_ce):
        """Reverse of ``_forward_losses``: the upstream gradients of the per-token losses -> every parameter gradient
        (train_step._Grads).  Runs inside PackedTrainStep.backward, i.e. without grad mode."""
        from . import train_step as TS
        F_ = tape.front
        dev, H, M = self.device, self.hidden_size, F_["M"]
        lm = self.language_model
        eng = lm.engine(check=False)
        grads = TS._Grads()
        last = F_["last"]
        d_last = torch.zeros((M, H), dtype=BF16, device=dev)
        if d_ce is not None and "logits" in F_:
            head = lm.lm_head.weight
            dlogits = ops.cross_entropy_bwd(F_["logits"], F_["labels"], d_ce.to(device=dev, dtype=torch.float32).contiguous())
            n = dlogits.shape[0]
            ops.gemm(dlogits, eng.wt_of("lm_head", head), d_last, c_rows0=F_["ce_rows"], M0=n, residual=d_last)
            if head.requires_grad:
                Xt = ops.transpose(last, rows=F_["ce_rows"], n=n)
                dW = torch.empty(tuple(head.shape), dtype=BF16, device=dev)
                grads.add(head, ops.gemm(ops.transpose(dlogits), Xt, dW))
        if d_mse is not None and "preds" in F_:
            lin = self.llm2vae
            dpred = ops.mse_rows_bwd(F_["preds"], F_["noise"], F_["clean"], F_["mse_src"], d_mse.to(device=dev, dtype=torch.float32).contiguous())
            n = dpred.shape[0]
            ops.gemm(dpred, eng.wt_of("llm2vae", lin.weight), d_last, c_rows0=F_["mse_rows"], M0=n, residual=d_last)
            if lin.weight.requires_grad or lin.bias.requires_grad:
                Xt = ops.transpose(last, rows=F_["mse_rows"], n=n)
                dW = torch.empty(tuple(lin.weight.shape), dtype=BF16, device=dev)
                grads.add(lin.weight, ops.gemm(ops.transpose(dpred), Xt, dW))
                grads.add(lin.bias, ops.colsum(dpred))
        g = TS.engine_backward_train(eng, tape, d_last, grads)            # d loss / d packed input sequence
        # ---- text tokens: embedding rows shared by several tokens sum their gradients (bagel.py:148)
        emb = lm.model.embed_tokens.weight
        if emb.requires_grad:
            ids, rows = F_["text_ids"], F_["text_rows"]
            order = torch.argsort(ids, stable=True)
            uniq, counts = torch.unique_consecutive(ids[order], return_counts=True)
            seg = torch.zeros(uniq.numel() + 1, dtype=torch.int32)
            seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
            dE = torch.zeros(tuple(emb.shape), dtype=BF16, device=dev)
            i32 = lambda x: x.to(device=dev, dtype=torch.int32)  # noqa: E731
            ops.rows_segment_sum(g, i32(rows[order]), i32(seg), i32(uniq), dE)
            grads.add(emb, dE)
        # ---- ViT tokens: connector (fc1 - gelu_tanh - fc2), then the SigLIP tower when it is trainable
        if "vit_feats" in F_:
            c = self.connector
            vit_tape = F_.get("vit_tape")
            if vit_tape is not None or any(p.requires_grad for p in c.parameters()):
                feats, hmid, n = F_["vit_feats"], F_["vit_hmid"], F_["vit_feats"].shape[0]
                d_emb = torch.empty((n, H), dtype=BF16, device=dev)
                ops.copy_rows(g, d_emb, n, H, src_rows=F_["vit_rows"])
                grads.add(c.fc2.weight, TS._wgrad(d_emb, hmid))
                grads.add(c.fc2.bias, ops.colsum(d_emb))
                d_h = torch.empty_like(hmid)
                ops.gemm(d_emb, TS._wt(c.fc2.weight.data), d_h)
                pre = torch.empty_like(hmid)
                ops.gemm(feats, c.fc1.weight.data, pre, bias0=c.fc1.bias.data)        # the un-activated fc1 output, recomputed
                ops.act_bwd(pre, d_h, ops.EPI_GELU_TANH)
                grads.add(c.fc1.weight, TS._wgrad(pre, feats))
                grads.add(c.fc1.bias, ops.colsum(pre))
                if vit_tape is not None:
                    d_feats = torch.empty_like(feats)
                    ops.gemm(pre, TS._wt(c.fc1.weight.data), d_feats)
                    TS.siglip_backward(self.vit_model, vit_tape, d_feats, grads)
        # ---- latent tokens: vae2llm + per-image timestep embedding (bagel.py:186-191); the position table is frozen
        if "x16" in F_:
            x16, n = F_["x16"], F_["x16"].shape[0]
            d_lat = torch.empty((n, H), dtype=BF16, device=dev)
            ops.copy_rows(g, d_lat, n, H, src_rows=F_["vae_rows"])
            v2l = self.vae2llm
            if v2l.weight.requires_grad or v2l.bias.requires_grad:
                grads.add(v2l.weight, TS._wgrad(d_lat, x16))
                grads.add(v2l.bias, ops.colsum(d_lat))
            te = self.time_embedder
            if any(p.requires_grad for p in te.parameters()):
                inv = F_["temb_inv"]
                nu = int(inv.max()) + 1
                # one row per distinct timestep = the column sum over that image's latent tokens (thousands of rows: the two-stage
                # column reduction, not the per-id gather reverse that sums a segment serially)
                d_temb = torch.stack([ops.colsum(d_lat, torch.nonzero(inv == u, as_tuple=False).flatten().to(device=dev, dtype=torch.int32))
                                      for u in range(nu)], dim=0)
                sinus, th = F_["time_sinus"], F_["time_h"]
                grads.add(te.mlp[2].weight, TS._wgrad(d_temb, th))
                grads.add(te.mlp[2].bias, ops.colsum(d_temb))
                d_th = torch.empty_like(th)
                ops.gemm(d_temb, TS._wt(te.mlp[2].weight.data), d_th)
                pre = torch.empty_like(th)
                ops.gemm(sinus, te.mlp[0].weight.data, pre, bias0=te.mlp[0].bias.data)
                ops.act_bwd(pre, d_th, ops.EPI_SILU)
                grads.add(te.mlp[0].weight, TS._wgrad(pre, sinus))
                grads.add(te.mlp[0].bias, ops.colsum(pre))
        return grads
