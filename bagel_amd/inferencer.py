"""InterleaveInferencer: stateful interleaved text/image context builder (API of the reference's inferencer.py:22-313).

Contexts are plain dicts {'kv_lens': [int], 'ropes': [int], 'past_key_values': NaiveCache}; the three CFG contexts
are kept in step exactly as the reference does (inferencer.py:229-256).  Differences are execution-only: tensors
live on the model's GPU, the VAE runs through the HIP kernels, and no torch.autocast region is needed (the kernels
own their precision; what the reference's region does to the VAE -- bf16 convolutions -- is ``vae_precision_in_autocast``)."""
from copy import deepcopy

import torch

from .modeling.bagel.qwen2_navit import NaiveCache

VLM_THINK_SYSTEM_PROMPT = '''You should first think about the reasoning process in the mind and then provide the user with the answer. 
The reasoning process is enclosed within <think> </think> tags, i.e. <think> reasoning process here </think> answer here'''

GEN_THINK_SYSTEM_PROMPT = '''You should first think about the planning process in the mind and then generate the image. 
The planning process is enclosed within <think> </think> tags, i.e. <think> planning process here </think> image here'''


def _is_pil(x):
    try:
        from PIL import Image
        return isinstance(x, Image.Image)
    except ImportError:
        return False


class _VaeAt:
    """The VAE pinned to one precision for the calls made through it (AutoEncoder.encode / decode ``precision=``)."""

    def __init__(self, vae, precision):
        self.vae, self.precision = vae, precision

    def encode(self, x, *a, **k):
        return self.vae.encode(x, *a, precision=self.precision, **k)

    def decode(self, z):
        return self.vae.decode(z, precision=self.precision)


class InterleaveInferencer:
    # How the VAE runs inside ``interleave_inference`` / ``__call__``.  The reference opens ``torch.autocast("cuda", bfloat16)`` around that
    # whole method (inferencer.py:233), so its VAE encode (edit requests) and decode run with bf16 convolutions there, while scripts that
    # call the VAE outside any autocast region (eval/gen/gen_images_mp.py:93) get fp32.  "bf16" mirrors the former; "fp32" keeps the
    # decoder in fp32 there too.  Direct calls of ``decode_image`` / ``update_context_image`` follow the caller's own autocast region.
    vae_precision_in_autocast = "bf16"

    def __init__(self, model, vae_model, tokenizer, vae_transform, vit_transform, new_token_ids):
        self.model = model
        self.vae_model = vae_model
        self.tokenizer = tokenizer
        self.vae_transform = vae_transform
        self.vit_transform = vit_transform
        self.new_token_ids = new_token_ids
        self._vae_precision = None

    def _vae(self):
        from .modeling.autoencoder import AutoEncoder
        if self._vae_precision is not None and isinstance(self.vae_model, AutoEncoder):
            return _VaeAt(self.vae_model, self._vae_precision)
        return self.vae_model

    # ---- context bookkeeping --------------------------------------------------------------------------
    def init_gen_context(self):
        return {"kv_lens": [0], "ropes": [0],
                "past_key_values": NaiveCache(self.model.config.llm_config.num_hidden_layers)}

    def _advance(self, ctx, prepare, update, **prepare_kw):
        gi, ctx["kv_lens"], ctx["ropes"] = prepare(curr_kvlens=ctx["kv_lens"], curr_rope=ctx["ropes"], **prepare_kw)
        ctx["past_key_values"] = update(ctx["past_key_values"], gi)
        return ctx

    @torch.no_grad()
    def update_context_text(self, text, gen_context):
        return self._advance(gen_context, self.model.prepare_prompts,
                             lambda kv, gi: self.model.forward_cache_update_text(kv, **gi),
                             prompts=[text], tokenizer=self.tokenizer, new_token_ids=self.new_token_ids)

    @torch.no_grad()
    def update_context_image(self, image, gen_context, vae=True, vit=True):
        assert vae or vit
        if vae:
            self._advance(gen_context, self.model.prepare_vae_images,
                          lambda kv, gi: self.model.forward_cache_update_vae(self._vae(), kv, **gi),
                          images=[image], transforms=self.vae_transform, new_token_ids=self.new_token_ids)
        if vit:
            self._advance(gen_context, self.model.prepare_vit_images,
                          lambda kv, gi: self.model.forward_cache_update_vit(kv, **gi),
                          images=[image], transforms=self.vit_transform, new_token_ids=self.new_token_ids)
        return gen_context

    # ---- generation -----------------------------------------------------------------------------------
    @torch.no_grad()
    def gen_image(self, image_shape, gen_context, cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_text_precontext=None,
                  cfg_img_precontext=None, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0, cfg_renorm_type="global",
                  num_timesteps=50, timestep_shift=3.0, enable_taylorseer=False):
        m = self.model
        gi = m.prepare_vae_latent(curr_kvlens=gen_context["kv_lens"], curr_rope=gen_context["ropes"],
                                  image_sizes=[image_shape], new_token_ids=self.new_token_ids)
        extra = {}
        for tag, pre in (("cfg_text", cfg_text_precontext), ("cfg_img", cfg_img_precontext)):
            c = m.prepare_vae_latent_cfg(curr_kvlens=pre["kv_lens"], curr_rope=pre["ropes"], image_sizes=[image_shape])
            extra[f"{tag}_past_key_values"] = pre["past_key_values"]
            extra[f"{tag}_packed_position_ids"] = c["cfg_packed_position_ids"]
            extra[f"{tag}_packed_query_indexes"] = c["cfg_packed_query_indexes"]
            extra[f"{tag}_key_values_lens"] = c["cfg_key_values_lens"]
            extra[f"{tag}_packed_key_value_indexes"] = c["cfg_packed_key_value_indexes"]
        latents = m.generate_image(past_key_values=gen_context["past_key_values"], num_timesteps=num_timesteps,
                                   cfg_text_scale=cfg_text_scale, cfg_img_scale=cfg_img_scale, cfg_interval=cfg_interval,
                                   cfg_renorm_min=cfg_renorm_min, cfg_renorm_type=cfg_renorm_type,
                                   timestep_shift=timestep_shift, enable_taylorseer=enable_taylorseer, **gi, **extra)
        return self.decode_image(latents[0], image_shape)

    def latent_to_chw(self, latent, image_shape):
        """(h*w, p*p*C) packed latent -> (1, C, h*p, w*p)   [einsum 'nhwpqc->nchpwq', inferencer.py:178-180]."""
        m = self.model
        H, W = image_shape
        h, w = H // m.latent_downsample, W // m.latent_downsample
        p, C = m.latent_patch_size, m.latent_channel
        return latent.reshape(1, h, w, p, p, C).permute(0, 5, 1, 3, 2, 4).reshape(1, C, h * p, w * p)

    def decode_image(self, latent, image_shape):
        from PIL import Image
        image = self._vae().decode(self.latent_to_chw(latent, image_shape))
        return Image.fromarray(self.image_to_u8(image).cpu().numpy())

    @staticmethod
    def image_to_u8(image):
        """(1, 3, H, W) decoder output -> (H, W, 3) uint8 on the GPU: ((x * 0.5 + 0.5).clamp(0, 1) * 255) with the reference's truncating
        cast (inferencer.py:182-183), one kernel.  A bf16 image (the decoder inside the autocast region) goes through the eager-bf16
        rounding points those elementwise ops have on a bf16 tensor."""
        from . import ops
        if image.dtype == torch.bfloat16:
            return ops.chw_bf16_to_u8(image[0])
        return ops.chw_f32_to_u8(image[0].float())

    @torch.no_grad()
    def gen_text(self, gen_context, max_length: int = 500, do_sample: bool = True, temperature: float = 1.0):
        ctx = deepcopy(gen_context)
        gi = self.model.prepare_start_tokens(ctx["kv_lens"], ctx["ropes"], self.new_token_ids)
        toks = self.model.generate_text(past_key_values=ctx["past_key_values"], max_length=max_length, do_sample=do_sample,
                                        temperature=temperature, end_token_id=self.new_token_ids["eos_token_id"], **gi)
        out = self.tokenizer.decode(toks[:, 0])
        return out.split("<|im_end|>")[0].split("<|im_start|>")[1]

    @torch.no_grad()
    def interleave_inference(self, input_lists, think=False, understanding_output=False, max_think_token_n=1000,
                             do_sample=False, text_temperature=0.3, cfg_text_scale=3.0, cfg_img_scale=1.5,
                             cfg_interval=[0.4, 1.0], timestep_shift=3.0, num_timesteps=50, cfg_renorm_min=0.0,
                             cfg_renorm_type="global", image_shapes=(1024, 1024), enable_taylorseer=False):
        # the region the reference wraps in torch.autocast(bfloat16) (inferencer.py:233): the VAE runs at vae_precision_in_autocast in here
        prev, self._vae_precision = self._vae_precision, self.vae_precision_in_autocast
        try:
            return self._interleave_inference(input_lists, think, understanding_output, max_think_token_n, do_sample, text_temperature,
                                              cfg_text_scale, cfg_img_scale, cfg_interval, timestep_shift, num_timesteps, cfg_renorm_min,
                                              cfg_renorm_type, image_shapes, enable_taylorseer)
        finally:
            self._vae_precision = prev

    def _interleave_inference(self, input_lists, think, understanding_output, max_think_token_n, do_sample, text_temperature, cfg_text_scale,
                              cfg_img_scale, cfg_interval, timestep_shift, num_timesteps, cfg_renorm_min, cfg_renorm_type, image_shapes,
                              enable_taylorseer):
        outputs = []
        gen_context = self.init_gen_context()
        cfg_text_context = deepcopy(gen_context)
        cfg_img_context = deepcopy(gen_context)
        if think:
            system_prompt = VLM_THINK_SYSTEM_PROMPT if understanding_output else GEN_THINK_SYSTEM_PROMPT
            gen_context = self.update_context_text(system_prompt, gen_context)
            cfg_img_context = self.update_context_text(system_prompt, cfg_img_context)
        for term in input_lists:
            if isinstance(term, str):
                cfg_text_context = deepcopy(gen_context)
                gen_context = self.update_context_text(term, gen_context)
                cfg_img_context = self.update_context_text(term, cfg_img_context)
            elif _is_pil(term):
                from .data.data_utils import pil_img2rgb
                term = self.vae_transform.resize_transform(pil_img2rgb(term))
                gen_context = self.update_context_image(term, gen_context, vae=not understanding_output)
                image_shapes = term.size[::-1]
                cfg_text_context = deepcopy(gen_context)
            else:
                raise ValueError(f"Unsupported input type: {type(term)}")
        if understanding_output:
            outputs.append(self.gen_text(gen_context, do_sample=do_sample, temperature=text_temperature,
                                         max_length=max_think_token_n))
            return outputs
        if think:
            thought = self.gen_text(gen_context, do_sample=do_sample, temperature=text_temperature, max_length=max_think_token_n)
            gen_context = self.update_context_text(thought, gen_context)
            outputs.append(thought)
        outputs.append(self.gen_image(image_shapes, gen_context, cfg_text_precontext=cfg_text_context,
                                      cfg_img_precontext=cfg_img_context, cfg_text_scale=cfg_text_scale,
                                      cfg_img_scale=cfg_img_scale, cfg_interval=cfg_interval, timestep_shift=timestep_shift,
                                      num_timesteps=num_timesteps, cfg_renorm_min=cfg_renorm_min,
                                      cfg_renorm_type=cfg_renorm_type, enable_taylorseer=enable_taylorseer))
        return outputs

    def __call__(self, image=None, text=None, **kargs):
        result = {"image": None, "text": None}
        if image is None and text is None:
            print("Please provide at least one input: either an image or text.")
            return result
        inputs = [x for x in (image, text) if x is not None]
        for item in self.interleave_inference(inputs, **kargs):
            if _is_pil(item):
                result["image"] = item
            elif isinstance(item, str):
                result["text"] = item
        return result
